#!/bin/bash
# round 6, call ac: where the 64 x 64 tile stops paying (N = 256 shapes at 5 / 6 / 8 rows x T 1000: 80 / 94 / 126 tiles of 128 x 128)
mkdir -p gpurun_out/r06_ac
for spec in "5 1000" "6 1000" "8 1000" "3 1000"; do set -- $spec
CBX_ROWS=$1 CBX_T=$2 CBX_PL_TILES=0,9,44,32 timeout 400 python scripts/df_micro.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_ac/df_micro_mid.log
done
