#!/bin/bash
# round 6, call ab: the 64 x 64 tile of small grids (batch 1) with a deeper ring (43: 3 stages) or 64-wide K tiles (44: 2 stages, 45: 3 stages) against the 2-stage BK = 32 form (9 = automatic there)
mkdir -p gpurun_out/r06_ab
timeout 600 python -m pytest tests/test_planes_gpu.py -x -q -m gpu -k "linear_tiles and (43 or 44 or 45)" 2>&1 | tail -3
for spec in "2 1000" "2 500" "4 1000"; do set -- $spec
CBX_ROWS=$1 CBX_T=$2 CBX_PL_TILES=0,9,43,44,45 timeout 400 python scripts/df_micro.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_ab/df_micro_small.log
done
