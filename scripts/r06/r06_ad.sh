#!/bin/bash
# round 6, call ad: full GPU suite on the build with the small-grid choices (plane attention: 128-query form, plane GEMM: 64 x 64 x 64 tiles up to 128 tiles of 128 x 128), then the batch-1 / batch-2 lines
mkdir -p gpurun_out/r06_ad
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 | tee gpurun_out/r06_ad/gpu_tests.txt
for b in 1 2 4; do CBX_B=$b CBX_LABEL=B$b timeout 300 python scripts/flow_ab.py 2>&1 | grep "flow ms" | tee -a gpurun_out/r06_ad/flow_small_batches.log; done
for spec in "mtl_b1:--batch 1" "turbo_b1:--workload turbo --batch 1" "nano_b1:--workload nano --batch 1" "mtl_b2:--batch 2"; do
  name=${spec%%:*}; flags=${spec#*:}
  timeout 500 python bench.py --steps 6 --warmup 2 --no-cpu-baseline $flags > gpurun_out/r06_ad/bench_$name.json 2> gpurun_out/r06_ad/bench_$name.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_ad/bench_$name.json").read().strip().splitlines()[-1])
print("$name", d["value"], d.get("value_serial"), d.get("stage_ms"), (d.get("decode_step") or {}).get("ms_per_step"), d.get("p50_first_audio_latency_ms_serial"))
PY
done
