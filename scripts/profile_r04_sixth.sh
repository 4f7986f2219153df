#!/bin/bash
# Sixth GPU call of round 4: the two tests whose expectations predated the new defaults; CU-partitioned pipelined schedule A/B.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/sixth
mkdir -p $O
cd $R
timeout 200 python -m pytest tests/test_models_gpu.py tests/test_rccl_world1_gpu.py tests/test_zz_abi_v9_gpu.py -q -m gpu -rfE -p no:cacheprovider -k "bf16_weight or torchrun or default_geometry" > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
timeout 400 python scripts/cu_mask_ab.py $O/cu_mask_ab.json > $O/cu_mask_ab.log 2>&1; grep -v amdgpu.ids $O/cu_mask_ab.log | tail -20
