#!/bin/bash
# Round 5, call M: one / two / three T3 batches in flight (each on its own high-priority stream) beside the flow stream, K = 16
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/m
mkdir -p $O
cd $R
for n in 3 2; do
CBX_PIPE_T3_STREAMS=$n timeout 300 python bench.py --steps 16 --warmup 4 --no-alt-precisions --no-streaming --no-cpu-baseline --no-autotune > $O/bench_t3_in_flight_$n.json 2> $O/bench_$n.err
tail -1 $O/bench_$n.err | cut -c1-200
python -c "
import json; d=json.load(open('$O/bench_t3_in_flight_$n.json'))
print('T3 in flight $n: value', d['value'], 'ms/step', d['ms_per_step'], 'p50 lat', d['p50_first_audio_latency_ms'], '| serial', d['other_schedule']['value'], '| decode in schedule', d.get('decode_step_in_throughput_schedule',{}).get('ms_per_step'))
"
done
