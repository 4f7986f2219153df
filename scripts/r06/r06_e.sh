#!/bin/bash
# round 6, call e: overlapped streaming synthesis (tests + bench line), the range-flag trip-rate test, the whole bench line with the new keys,
# rocprofv3 kernel stats of Turbo at batch 1 on the DPP build
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_e
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_stream_gpu.py tests/test_models_gpu.py tests/test_examples_gpu.py -q -m gpu -s -k "stream or range_flag or pipelined or example" > $O/pytest.txt 2>&1
grep -E "passed|failed|error|scale|e\+0|/4" $O/pytest.txt | tail -20
timeout 600 python bench.py --steps 8 --warmup 2 > $O/bench_mtl.json 2> $O/bench_mtl.err
python - <<PY
import json
d=json.loads(open("$O/bench_mtl.json").read().strip().splitlines()[-1])
print("mtl", d["value"], d["stage_ms"], d["decode_step"]["ms_per_step"], "serial", d.get("value_serial"), "bf16x6", d.get("value_bf16x6"), "lat", d.get("p50_first_audio_latency_ms_pipelined"), d.get("p50_first_audio_latency_ms_serial"), d.get("p50_first_audio_latency_ms_streaming"), "trips", d.get("f16x3_range_trips"), d.get("bf16x6_repeat_cost_ms"))
print(d.get("streaming"))
PY
tail -3 $O/bench_mtl.err
cd /tmp && export TMPDIR=/tmp
CBX_TURBO_TUNE="row_splits=16,row_chunks=2" timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_turbo -o turbo -- python $R/bench.py --workload turbo --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-streaming --no-parity > $O/bench_turbo_under_rocprof.json 2> /tmp/rocprof_turbo.err
cp $(find /tmp/prof_turbo -name "*kernel_stats.csv" | head -1) $O/turbo_b1_kernel_stats.csv
head -9 $O/turbo_b1_kernel_stats.csv | cut -c1-160
