#!/bin/bash
# round 6, call p: the plane-GEMM forms again, INTERLEAVED rounds (call n / o: the position in a sequential sweep decided more than the form)
mkdir -p gpurun_out/r06_p
CBX_PL_TILES=0,32,35,41,42,8,17 timeout 600 python scripts/df_micro.py 2>&1 | tee gpurun_out/r06_p/df_micro.log
CBX_PL_TILES=42,41,35,32,0 timeout 600 python scripts/df_micro.py 2>&1 | tee gpurun_out/r06_p/df_micro_reversed.log
CBX_ROWS=64 CBX_PL_TILES=0,32,35,41,42 CBX_REPS=20 timeout 600 python scripts/df_micro.py 2>&1 | tee gpurun_out/r06_p/df_micro_rows64.log
