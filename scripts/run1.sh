timeout 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "gemv" 2>&1 | tail -15
timeout 300 python -m pytest tests/test_models_gpu.py -q -m gpu -x -k "t3" 2>&1 | tail -15
timeout 200 python scripts/prof_t3.py 30 8 2>&1 | tail -6
