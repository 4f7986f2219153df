#!/bin/bash
# A/B of GEMM tile variants inside ONE GPU call (same box, warm clocks): full bench, serial, no CPU baseline.
mkdir -p gpurun_out
for t in 0 32 0 32; do
  echo "== CBX_GEMM_TILE=$t" >> gpurun_out/ab.log
  CBX_GEMM_TILE=$t CBX_BENCH_VERBOSE=1 timeout 240 python bench.py --steps 2 --warmup 1 --no-cpu-baseline >> gpurun_out/ab.log 2>&1
done
grep -E "==|\"metric\"|flow_s|t3_s" gpurun_out/ab.log | cut -c1-400
