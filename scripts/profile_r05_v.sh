#!/bin/bash
# Round 5, call V: stream priorities in the throughput schedule (the flow stream is the critical one with two decode chains in flight): T3 high / flow normal (shipped),
# both normal, T3 normal / flow high
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/v
mkdir -p $O
cd $R
for pr in "0,0" "0,-1" "-1,0"; do
n=$(echo $pr | tr ',-' '_m')
CBX_PIPE_PRIO=$pr timeout 300 python bench.py --steps 16 --warmup 4 --no-alt-precisions --no-streaming --no-cpu-baseline --no-autotune > $O/bench_prio_$n.json 2> $O/bench_$n.err
python -c "
import json; d=json.load(open('$O/bench_prio_$n.json'))
print('priorities (T3, flow) = $pr: value', d['value'], 'ms/step', d['ms_per_step'], 'p50 lat', d['p50_first_audio_latency_ms'], '| serial', d['other_schedule']['value'], '| decode in schedule', d.get('decode_step_in_throughput_schedule',{}).get('ms_per_step'))
"
done
