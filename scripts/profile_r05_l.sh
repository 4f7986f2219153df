#!/bin/bash
# Round 5, call L: two T3 streams: co-resident forms on / off; the driver's K = 20
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/l
mkdir -p $O
cd $R
run() { # name env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps ${STEPS:-8} --warmup 3 --no-alt-precisions --no-streaming --no-cpu-baseline --no-autotune > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "
import json; d=json.load(open('$O/bench_$name.json'))
print('$name: value', d['value'], 'ms/step', d['ms_per_step'], 'p50 lat', d['p50_first_audio_latency_ms'], '| serial', d['other_schedule']['value'], '| decode in schedule', d.get('decode_step_in_throughput_schedule',{}).get('ms_per_step'))
"
}
run s2_cores0 CBX_PIPE_T3_STREAMS=2 CBX_PIPE_CORES=0
run s1_cores0 CBX_PIPE_T3_STREAMS=1 CBX_PIPE_CORES=0
STEPS=20 run s2_cores1_k20 CBX_PIPE_T3_STREAMS=2 CBX_PIPE_CORES=1
