#!/bin/bash
# round 6, call n: deferred-epilogue forms of the plane GEMM -- parity (bit for bit against the plain twins) and per-shape timing
mkdir -p gpurun_out/r06_n
timeout 900 python -m pytest tests/test_planes_gpu.py -x -q -m gpu -k "deferred_epilogue or linear_tiles" 2>&1 | tail -5
for i in 1 2; do timeout 300 python scripts/df_micro.py 2>&1 | tee -a gpurun_out/r06_n/df_micro.log; done
CBX_ROWS=64 timeout 300 python scripts/df_micro.py 2>&1 | tee -a gpurun_out/r06_n/df_micro_rows64.log
