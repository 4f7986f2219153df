#!/bin/bash
# Round 5, call J: synthesize_pipelined with two T3 states (the worker thread runs ahead) -- equality test + bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/j
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_models_gpu.py -q -m gpu -p no:cacheprovider -rfE -k "pipelined" > $O/pytest_pipelined.log 2>&1; tail -3 $O/pytest_pipelined.log
timeout 400 python bench.py --steps 10 --warmup 3 --no-alt-precisions --no-streaming --no-cpu-baseline > $O/bench_pipelined_steps10.json 2> $O/bench.err
tail -2 $O/bench.err | cut -c1-300
python -c "
import json; d=json.load(open('$O/bench_pipelined_steps10.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'], 'schedule', d['schedule'], 'p50 lat', d['p50_first_audio_latency_ms'])
print('other', d.get('other_schedule')); print('stage_ms', d['stage_ms']); print('decode', d['decode_step']['ms_per_step'], d['decode_step']['frac'], d.get('decode_step_in_throughput_schedule'))
"
