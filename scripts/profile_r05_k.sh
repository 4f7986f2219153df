#!/bin/bash
# Round 5, call K: two T3 streams (consecutive batches' decodes beside each other) against one
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/k
mkdir -p $O
cd $R
for n in 2 1; do
CBX_PIPE_T3_STREAMS=$n timeout 300 python bench.py --steps 8 --warmup 3 --no-alt-precisions --no-streaming --no-cpu-baseline --no-autotune > $O/bench_t3streams_$n.json 2> $O/bench_$n.err
tail -1 $O/bench_$n.err | cut -c1-300
python -c "
import json; d=json.load(open('$O/bench_t3streams_$n.json'))
print('T3 streams $n: value', d['value'], 'ms/step', d['ms_per_step'], 'p50 lat', d['p50_first_audio_latency_ms'], '| serial', d['other_schedule']['value'], '| decode in schedule', d.get('decode_step_in_throughput_schedule',{}).get('ms_per_step'))
"
done
