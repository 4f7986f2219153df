#!/bin/bash
# round 6, call w: the automatic plane-GEMM choice at the other sizes it serves (batch 1: 2 rows x T 1000; 60 s voice conversion: 2 / 8 rows x T 3500; streaming round 0: 16 rows x T 530), interleaved
mkdir -p gpurun_out/r06_w
for spec in "2 1000" "2 3500" "8 3500" "16 530"; do set -- $spec
CBX_ROWS=$1 CBX_T=$2 CBX_PL_TILES=0,32,35,42,41 CBX_REPS=30 timeout 400 python scripts/df_micro.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_w/df_micro_sizes.log
done
