#!/bin/bash
# round 6, call m: the lean single-pass form of the few-row kernel -- Turbo / Nano lines (3 runs each: box noise), row tests, measured-vs-floor tolerances
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_m
mkdir -p $O
cd $R
rm -f gpurun_out/measured_vs_floor.jsonl
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py tests/test_baseline_shapes_gpu.py -q -m gpu -k "gemv_row or attn_parts or turbo or hift_full or f0_max or bench_shape or e2e or 60" > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
cat gpurun_out/measured_vs_floor.jsonl
for i in 1 2 3; do
for spec in "turbo_b1:--workload turbo --batch 1" "nano_b1:--workload nano --batch 1"; do
  name=${spec%%:*}; flags=${spec#*:}
  timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-streaming --no-parity $flags > $O/bench_${name}_$i.json 2> $O/bench_${name}_$i.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_${name}_$i.json").read().strip().splitlines()[-1])
    print("$name", d["value"], (d.get("decode_step") or {}).get("ms_per_step"), (d.get("decode_step") or {}).get("frac"))
except Exception as e:
    print("$name FAILED", e)
PY
done
done
