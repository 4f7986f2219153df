#!/bin/bash
# round 6, call l: whole hardware suite on the few-row kernels (M <= 4; engines: Turbo / Nano B <= 2), Turbo at B = 1 / 2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_l
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
for spec in "turbo_b1:--workload turbo --batch 1" "turbo_b2:--workload turbo --batch 2" "nano_b1:--workload nano --batch 1"; do
  name=${spec%%:*}; flags=${spec#*:}
  timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-streaming $flags > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d.get("stage_ms"), (d.get("decode_step") or {}).get("ms_per_step"), (d.get("decode_step") or {}).get("frac"), (d.get("roofline") or {}).get("kernel","")[:30], (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("traffic"))
except Exception as e:
    print("$name FAILED", e)
PY
done
