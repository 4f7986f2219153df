#!/bin/bash
timeout 200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "layernorm or decode_attn" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "flow or meanflow or end_to_end" 2>&1 | tail -3
CBX_BENCH_VERBOSE=1 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tail -4
