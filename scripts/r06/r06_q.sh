#!/bin/bash
# round 6, call q: where the loader-wave forms of the plane GEMM spend their time (diag switches, interleaved), loader priority
mkdir -p gpurun_out/r06_q
CBX_DIAGS=0,7,23,16,18,17,19 CBX_DIAG_TILES=32,42 CBX_LIB_PATH=$PWD/chatterbox_amd/build/libcbx_hip_diag.so timeout 900 python scripts/diag_loader_forms.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_q/diag_loader_forms_2.log
