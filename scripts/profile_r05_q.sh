#!/bin/bash
# Round 5, call Q: two T3 batches in flight: which flow kernels need their co-resident form? attention 4 / 5 x GEMM tiles default / co-resident, K = 16
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/q
mkdir -p $O
cd $R
run() { name=$1; shift
env "$@" timeout 300 python bench.py --steps 16 --warmup 4 --no-alt-precisions --no-streaming --no-cpu-baseline --no-autotune > $O/bench_$name.json 2> $O/bench_$name.err
python -c "
import json; d=json.load(open('$O/bench_$name.json'))
print('$name: value', d['value'], 'ms/step', d['ms_per_step'], '| serial', d['other_schedule']['value'], '| decode in schedule', d.get('decode_step_in_throughput_schedule',{}).get('ms_per_step'))
"; }
run attn4_tile_cores CBX_PIPE_ATTN=4
run attn5_tile_default CBX_PIPE_TILE=0
run attn5_tile_cores CBX_FOO=1
