#!/bin/bash
# Round 5, call P: with two T3 batches in flight the flow stream is the critical one: its LayerNorm / split-GEMM launches capped (stream attribute) or not
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/p
mkdir -p $O
cd $R
for a in 0 1; do
CBX_PIPE_STREAM_ATTR=$a timeout 300 python bench.py --steps 16 --warmup 4 --no-alt-precisions --no-streaming --no-cpu-baseline --no-autotune > $O/bench_stream_attr_$a.json 2> $O/bench_$a.err
tail -1 $O/bench_$a.err | cut -c1-200
python -c "
import json; d=json.load(open('$O/bench_stream_attr_$a.json'))
print('co-resident stream attribute $a: value', d['value'], 'ms/step', d['ms_per_step'], 'p50 lat', d['p50_first_audio_latency_ms'], '| serial', d['other_schedule']['value'], '| decode in schedule', d.get('decode_step_in_throughput_schedule',{}).get('ms_per_step'))
"
done
