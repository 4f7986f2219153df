export CBX_BENCH_VERBOSE=1
timeout 200 python bench.py --workload turbo --batch 1 --steps 3 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -2
timeout 200 python bench.py --workload nano --batch 1 --steps 3 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -1
timeout 200 python bench.py --workload turbo --batch 8 --steps 3 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -1
