#!/bin/bash
# Round 5, call O: the co-resident plane GEMM on tile 8 (128 x 128 x 64, 2 stages, 128 KiB) against tile 17 (x 32, 3 stages, 96 KiB), K = 16
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/o
mkdir -p $O
cd $R
for t in 8 17; do
CBX_PL_CORES_K64=$t timeout 300 python bench.py --steps 16 --warmup 4 --no-alt-precisions --no-streaming --no-cpu-baseline --no-autotune > $O/bench_cores_tile_$t.json 2> $O/bench_$t.err
tail -1 $O/bench_$t.err | cut -c1-200
python -c "
import json; d=json.load(open('$O/bench_cores_tile_$t.json'))
print('co-resident GEMM tile $t: value', d['value'], 'ms/step', d['ms_per_step'], 'p50 lat', d['p50_first_audio_latency_ms'], '| serial', d['other_schedule']['value'], '| decode in schedule', d.get('decode_step_in_throughput_schedule',{}).get('ms_per_step'))
"
done
