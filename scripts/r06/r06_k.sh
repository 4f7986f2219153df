#!/bin/bash
# round 6, call k: the few-row path generalised to 1 .. 4 rows (Turbo / Nano B <= 4, Llama B <= 2): hardware tests, then same-box A/B of Multilingual at B = 1 / 2 and Turbo at B = 2 / 4
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_k
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
for spec in "mtl_b1_row:CBX_T3_ROW_PATH=1:--batch 1" "mtl_b1_mfma:CBX_T3_ROW_PATH=0:--batch 1" "mtl_b2_row:CBX_T3_ROW_PATH=1:--batch 2" "mtl_b2_mfma:CBX_T3_ROW_PATH=0:--batch 2" \
            "turbo_b2_row:CBX_TURBO_TUNE=row_path=1:--workload turbo --batch 2" "turbo_b2_mfma:CBX_TURBO_TUNE=row_path=0:--workload turbo --batch 2" \
            "turbo_b4_row:CBX_TURBO_TUNE=row_path=1:--workload turbo --batch 4" "turbo_b4_mfma:CBX_TURBO_TUNE=row_path=0:--workload turbo --batch 4"; do
  name=${spec%%:*}; rest=${spec#*:}; envv=${rest%%:*}; flags=${rest#*:}
  env $envv timeout 400 python bench.py --steps 4 --warmup 1 --schedule serial --no-cpu-baseline --no-streaming --no-alt-precisions --no-autotune $flags > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d.get("stage_ms"), (d.get("decode_step") or {}).get("ms_per_step"), (d.get("decode_step") or {}).get("frac"), d.get("p50_first_audio_latency_ms"))
except Exception as e:
    print("$name FAILED", e)
PY
done
