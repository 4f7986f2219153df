#!/bin/bash
# round 6, call h: the token loop in C (cbx_t3_loop_*) -- whole hardware suite (every T3 golden now runs through it), bench line, Turbo / Nano lines
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_h
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
timeout 700 python bench.py --steps 8 --warmup 2 > $O/bench_mtl.json 2> $O/bench_mtl.err
python - <<PY
import json
d=json.loads(open("$O/bench_mtl.json").read().strip().splitlines()[-1])
print("mtl", d["value"], d["stage_ms"], d["decode_step"]["ms_per_step"], "serial", d.get("value_serial"), "bf16x6", d.get("value_bf16x6"), "lat", d.get("p50_first_audio_latency_ms_pipelined"), d.get("p50_first_audio_latency_ms_serial"), d.get("p50_first_audio_latency_ms_streaming"), "trips", d.get("f16x3_range_trips"), d.get("bf16x6_repeat_cost_ms"))
print(d.get("streaming"))
print(d.get("parity"))
PY
tail -3 $O/bench_mtl.err
