#!/bin/bash
# round 6, call c: Turbo / Nano at batch 1 on the row path with the 4-way sampler search: bench lines + rocprofv3 kernel stats of the Turbo run
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_c
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py -q -m gpu -x -k "sampler or turbo" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
for wl in turbo nano; do
  CBX_TURBO_TUNE="row_splits=16,row_chunks=2" timeout 300 python bench.py --workload $wl --batch 1 --steps 6 --warmup 2 --no-cpu-baseline --no-streaming > $O/bench_${wl}_b1.json 2> $O/bench_${wl}_b1.err
  python - <<PY
import json
d=json.loads(open("$O/bench_${wl}_b1.json").read().strip().splitlines()[-1])
print("$wl", d["value"], d["stage_ms"], d["decode_step"]["ms_per_step"], d["decode_step"]["frac"])
PY
done
cd /tmp && export TMPDIR=/tmp
CBX_TURBO_TUNE="row_splits=16,row_chunks=2" timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_turbo -o turbo -- python $R/bench.py --workload turbo --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-streaming --no-parity > $O/bench_turbo_under_rocprof.json 2> /tmp/rocprof_turbo.err
cp $(find /tmp/prof_turbo -name "*kernel_stats.csv" | head -1) $O/turbo_b1_kernel_stats.csv
head -14 $O/turbo_b1_kernel_stats.csv | cut -c1-200
