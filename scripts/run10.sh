timeout 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "flash or conv or linear or bmm" 2>&1 | tail -2
for t in 1 0; do
  echo "== CBX_FLASH_PREFETCH=$t"
  CBX_FLASH_PREFETCH=$t timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['config']['stage_ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us'])
"
done
