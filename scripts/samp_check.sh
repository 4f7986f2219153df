#!/bin/bash
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "sampler" 2>&1 | tail -5
timeout 200 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "t3 or end_to_end" 2>&1 | tail -5
CBX_BENCH_VERBOSE=1 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | cut -c1-420 | tail -5
