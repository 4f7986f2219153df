timeout 300 python -m pytest tests/test_models_gpu.py -q -m gpu -x -k "pipelined or end_to_end" 2>&1 | tail -3
export CBX_BENCH_VERBOSE=1
timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('PIPE', d['value'], d['ms_per_step'], d['p50_first_audio_latency_ms'], d['roofline']['achieved'])
"
timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --serial 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('SERIAL', d['value'], d['ms_per_step'], d['p50_first_audio_latency_ms'], d['roofline']['achieved'])
"
