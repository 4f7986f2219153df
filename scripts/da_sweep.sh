#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/da.log
timeout 150 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "decode_attn" 2>&1 | tail -3 >> gpurun_out/da.log
for u in 4 8 16; do
  echo "== DA_U $u" >> gpurun_out/da.log
  CBX_DA_U=$u timeout 120 python scripts/prof_t3.py 30 8 2>&1 | grep -v amdgpu.ids | tail -3 >> gpurun_out/da.log
done
CBX_DA_U=16 timeout 150 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "t3_vs_reference or t3_batched" 2>&1 | tail -3 >> gpurun_out/da.log
cat gpurun_out/da.log
