#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/shapes.log
export CBX_GEMM_SHAPES=qkv,attn_out,ff1+gelu,ff2,conv3_256,conv3_320,res1x1,enc_ff1
for t in 128 12864 64; do
  echo "== split tile $t" >> gpurun_out/shapes.log
  CBX_SPLIT_TILE=$t CBX_PRECS=1,3,6 timeout 100 python scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/shapes.log
done
cat gpurun_out/shapes.log
