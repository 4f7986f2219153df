#!/bin/bash
# round 6, call y: plane attention with the DMAs of tile t + 2 issued between the softmax and the PV product (version 6) against the default (4) and the staggered form (2); parity first
mkdir -p gpurun_out/r06_y
timeout 900 python -m pytest tests/test_planes_gpu.py -x -q -m gpu -k "flash_attn_planes" 2>&1 | tail -3
timeout 400 python scripts/attn_micro.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_y/attn_micro.log
CBX_ROWS=2 CBX_T=3500 CBX_REPS=10 timeout 400 python scripts/attn_micro.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_y/attn_micro.log
CBX_ROWS=2 CBX_T=1000 timeout 400 python scripts/attn_micro.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_y/attn_micro.log
