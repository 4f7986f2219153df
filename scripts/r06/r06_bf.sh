#!/bin/bash
# round 6, call bf: the other Llama-T3 workloads on the write-through build (Multilingual at B = 1 / 32)
O=gpurun_out/r06_bf
mkdir -p $O
for spec in "mtl_b1:--batch 1" "mtl_b32:--batch 32 --steps 3 --warmup 1"; do
  name=${spec%%:*}; flags=${spec#*:}
  timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline $flags > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d.get("stage_ms"), (d.get("decode_step") or {}).get("ms_per_step"), (d.get("decode_step") or {}).get("frac"))
except Exception as e:
    print("$name FAILED", e)
PY
done
