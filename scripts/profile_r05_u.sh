#!/bin/bash
# Round 5, call U: the vocoder of batch k on its own stream beside encoder + CFM(k + 1) (alternating range-flag words): equality test, bench pair
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/u
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_models_gpu.py tests/test_ops_gpu.py tests/test_planes_gpu.py -q -m gpu -p no:cacheprovider -rfE -k "pipelined or range or fp16_range or voice" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for h in 1 0; do
CBX_PIPE_HIFT_STREAM=$h timeout 300 python bench.py --steps 16 --warmup 4 --no-alt-precisions --no-streaming --no-cpu-baseline --no-autotune > $O/bench_hift_stream_$h.json 2> $O/bench_$h.err
tail -1 $O/bench_$h.err | cut -c1-200
python -c "
import json; d=json.load(open('$O/bench_hift_stream_$h.json'))
print('vocoder on its own stream $h: value', d['value'], 'ms/step', d['ms_per_step'], 'p50 lat', d['p50_first_audio_latency_ms'], '| serial', d['other_schedule']['value'], d['stage_ms'])
"
done
