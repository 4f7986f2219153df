#!/bin/bash
# Round 5, call Y: with the flow stream critical, does the decode step still want its co-residency geometry?  (T3 default geometry beside the co-resident flow forms)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/y
mkdir -p $O
cd $R
for c in 0 1; do
CBX_PIPE_T3_CORES=$c timeout 300 python bench.py --steps 16 --warmup 4 --no-alt-precisions --no-streaming --no-cpu-baseline --no-autotune > $O/bench_t3_cores_$c.json 2> $O/bench_$c.err
python -c "
import json; d=json.load(open('$O/bench_t3_cores_$c.json'))
print('T3 co-residency geometry $c: value', d['value'], 'ms/step', d['ms_per_step'], '| serial', d['other_schedule']['value'], '| decode in schedule', d.get('decode_step_in_throughput_schedule',{}).get('ms_per_step'))
"
done
