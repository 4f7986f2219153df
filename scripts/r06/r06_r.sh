#!/bin/bash
# round 6, call r: operand prefetch (two register sets, requests of K chunk kc + 1 in front of the MFMA group of kc) in the loader-wave forms
mkdir -p gpurun_out/r06_r
timeout 900 python -m pytest tests/test_planes_gpu.py -x -q -m gpu -k "deferred_epilogue or linear_tiles or transposed_column" 2>&1 | tail -3
CBX_PL_TILES=0,32,35,41,42,21,33 timeout 600 python scripts/df_micro.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_r/df_micro.log
CBX_DIAGS=0,16,17,18 CBX_DIAG_TILES=32,35 CBX_LIB_PATH=$PWD/chatterbox_amd/build/libcbx_hip_diag.so timeout 900 python scripts/diag_loader_forms.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_r/diag.log
