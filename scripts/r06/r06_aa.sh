#!/bin/bash
# round 6, call aa: the automatic small-grid form of the plane attention in situ (batch-1 flow, A = this build, B = forced version 4 through the test hook is not available per process:
# so the batch-1 flow is timed and profiled; compare with call z's per-launch numbers), parity of every version, and the batch-1 bench lines
mkdir -p gpurun_out/r06_aa
timeout 900 python -m pytest tests/test_planes_gpu.py tests/test_baseline_shapes_gpu.py -x -q -m gpu -k "flash_attn_planes or estimator_planes or s3gen_t1000 or e2e_8s" 2>&1 | tail -3
CBX_B=1 CBX_LABEL=b1 timeout 300 python scripts/flow_ab.py 2>&1 | grep "flow ms" | tee gpurun_out/r06_aa/flow_b1.log
cd /tmp && export TMPDIR=/tmp
CBX_B=1 CBX_REPS=3 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b1 -o f -- python $GRAFT_REPO_ROOT/scripts/flow_ab.py > /tmp/prof_b1.log 2>&1
cp $(find /tmp/prof_b1 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r06_aa/flow_b1_kernel_stats.csv
cd $GRAFT_REPO_ROOT
head -16 gpurun_out/r06_aa/flow_b1_kernel_stats.csv | cut -c1-200
for spec in "mtl_b1:--batch 1" "turbo_b1:--workload turbo --batch 1" "nano_b1:--workload nano --batch 1"; do
  name=${spec%%:*}; flags=${spec#*:}
  timeout 500 python bench.py --steps 6 --warmup 2 --no-cpu-baseline $flags > gpurun_out/r06_aa/bench_$name.json 2> gpurun_out/r06_aa/bench_$name.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_aa/bench_$name.json").read().strip().splitlines()[-1])
print("$name", d["value"], d.get("stage_ms"), (d.get("decode_step") or {}).get("ms_per_step"))
PY
done
