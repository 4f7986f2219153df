#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "split" 2>&1 | tail -15 > gpurun_out/split_tests.log
cat gpurun_out/split_tests.log
timeout 200 python scripts/split_eval.py golden 2>&1 | grep -v amdgpu.ids > gpurun_out/split_speed.log
timeout 200 python scripts/split_eval.py speed 2>&1 | grep -v amdgpu.ids >> gpurun_out/split_speed.log
cat gpurun_out/split_speed.log
