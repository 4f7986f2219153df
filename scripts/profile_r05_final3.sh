#!/bin/bash
# Round 5: the driver's bench command on the round's last commit (the line under profiles/r05_bench_steps20_warmup5.json), then smoke()
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/final3
mkdir -p $O
cd $R
CBX_BENCH_VERBOSE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench_steps20_warmup5.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'], d['schedule'], 'p50 lat', d['p50_first_audio_latency_ms'], '| other', d['other_schedule']['value'], d['other_schedule']['ms_per_step'])
print(d['stage_ms'], d['decode_step']['ms_per_step'], d['decode_step']['frac'], d['roofline']['frac'], d['t3_geometry_throughput_schedule'], d['parity']['tokens_equal'], d['parity']['mel_l1'], d['parity']['wav_rmse'])
"
timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -1
