#!/bin/bash
# Round 5, call I: ABI v13 (per-call geometry, co-resident forms, threaded T3 enqueue) -- parity tests of everything touched, then the bench with the
# throughput schedule as the timed region and the serial schedule measured beside it
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/i
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_planes_gpu.py tests/test_zzz_stage_seams_gpu.py -q -m gpu -p no:cacheprovider -rfE > $O/pytest_planes_seams.log 2>&1; tail -3 $O/pytest_planes_seams.log
timeout 600 python -m pytest tests/test_models_gpu.py tests/test_baseline_shapes_gpu.py tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -rfE -k "pipelined or end_to_end or flow_t1000 or rms_fused or e2e or flow or voice" > $O/pytest_models.log 2>&1; tail -3 $O/pytest_models.log
timeout 400 python bench.py --steps 10 --warmup 3 --no-alt-precisions --no-streaming --no-cpu-baseline > $O/bench_pipelined_steps10.json 2> $O/bench.err
tail -2 $O/bench.err | cut -c1-300
python -c "
import json; d=json.load(open('$O/bench_pipelined_steps10.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'], 'schedule', d['schedule'], 'p50 lat', d['p50_first_audio_latency_ms'])
print('other', d.get('other_schedule')); print('stage_ms', d['stage_ms']); print('decode', d['decode_step']['ms_per_step'], d['decode_step']['frac'], d.get('decode_step_in_throughput_schedule'))
print('t3_geometry', d['t3_geometry']); print('roofline', d['roofline']['kernel'][:40], d['roofline']['frac'], d['roofline']['avg_launch_us'])
for r in d['roofline_secondary']: print('   ', r['kernel'][:40], r['frac'], r['avg_launch_us'], r['share_of_step'])
"
