#!/bin/bash
# Round 5, call X: the bench line with the throughput-schedule geometry key (short run), configs[3] on one GPU (256 utterances in device batches of 32)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/x
mkdir -p $O
cd $R
timeout 300 python bench.py --steps 6 --warmup 2 --no-alt-precisions --no-streaming --no-cpu-baseline --no-autotune --config3 > $O/bench_config3.json 2> $O/bench.err
tail -1 $O/bench.err | cut -c1-200
python -c "
import json; d=json.load(open('$O/bench_config3.json'))
print('value', d['value'], d['ms_per_step'], d['schedule'], '| serial', d['other_schedule']['value'])
print('t3_geometry_throughput_schedule', d['t3_geometry_throughput_schedule'])
print('configs3', d.get('configs3'))
"
