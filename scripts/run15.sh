timeout 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "conv or linear or bmm or swiglu or flash" 2>&1 | tail -2
for rk in gemm_f32 flash_attn_f32; do
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --serial --roofline-kernel $rk 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['config']['stage_ms_per_step'], d['roofline']['kernel'][:20], d['roofline']['achieved'], d['roofline']['avg_launch_us'])
"
done
