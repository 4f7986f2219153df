#!/bin/bash
# round 6, call z: plane attention on a part-filled chip (batch 1: 2 rows x 8 heads x 4 query tiles of 256 = 64 workgroups): the 128-query forms (versions 1, 5) against the default
mkdir -p gpurun_out/r06_z
for spec in "2 1000" "2 500" "4 1000" "2 3500" "8 1000"; do set -- $spec
CBX_ROWS=$1 CBX_T=$2 CBX_ATTN_VERSIONS=4,5,1 timeout 400 python scripts/attn_micro.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_z/attn_micro_small_grids.log
done
