timeout 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -8
