timeout 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "conv or linear or bmm or swiglu" 2>&1 | tail -2
for t in 0 1288 12864; do
  echo "== CBX_GEMM_TILE=$t"
  CBX_GEMM_TILE=$t timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['config']['stage_ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us'])
"
done
