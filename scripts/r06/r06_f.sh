#!/bin/bash
# round 6, call f: bisection sampler on ping-pong reductions, streaming with a fast first round (3 rounds), the fixed pipelined test
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_f
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_stream_gpu.py tests/test_models_gpu.py tests/test_ops_gpu.py -q -m gpu -k "stream or pipelined or sampler or turbo" > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_mtl.json 2> $O/bench_mtl.err
python - <<PY
import json
d=json.loads(open("$O/bench_mtl.json").read().strip().splitlines()[-1])
print("mtl", d["value"], d["stage_ms"], d["decode_step"]["ms_per_step"], "serial", d.get("value_serial"), "bf16x6", d.get("value_bf16x6"))
print(d.get("streaming"))
PY
tail -3 $O/bench_mtl.err
cd /tmp && export TMPDIR=/tmp
CBX_TURBO_TUNE="row_splits=16,row_chunks=2" timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_turbo -o turbo -- python $R/bench.py --workload turbo --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-streaming --no-parity > $O/bench_turbo_under_rocprof.json 2> /tmp/rocprof_turbo.err
cp $(find /tmp/prof_turbo -name "*kernel_stats.csv" | head -1) $O/turbo_b1_kernel_stats.csv
head -8 $O/turbo_b1_kernel_stats.csv | cut -c1-160
cd $R
CBX_TURBO_TUNE="row_splits=16,row_chunks=2" timeout 300 python bench.py --workload turbo --batch 1 --steps 6 --warmup 2 --no-cpu-baseline --no-streaming > $O/bench_turbo_b1.json 2> $O/bench_turbo_b1.err
python - <<PY
import json
d=json.loads(open("$O/bench_turbo_b1.json").read().strip().splitlines()[-1])
print("turbo", d["value"], d["stage_ms"], d["decode_step"]["ms_per_step"], d["decode_step"]["frac"])
PY
