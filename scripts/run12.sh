export CBX_BENCH_VERBOSE=1
timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | tail -3
