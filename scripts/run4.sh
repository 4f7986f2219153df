timeout 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -x 2>&1 | tail -5
timeout 120 python scripts/bench_gemm.py 2>&1 | tail -16
