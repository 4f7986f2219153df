#!/bin/bash
# round 6, call ap: the prefill of the GPT-2 backbones through cbx_gpt2_prefill (ABI v16) behind the cached conditioning prefix: parity, prefill wall time, Turbo / Nano at batch 1
# with both (a) and with neither (b: CBX_T3_CSTEP... no: CBX_T3_SHARE_PREFIX=0 and the Python sequence are only reachable per engine; b = the round-end numbers of call an / final)
mkdir -p gpurun_out/r06_ap
timeout 1500 python -m pytest tests/test_models_gpu.py tests/test_baseline_shapes_gpu.py tests/test_examples_gpu.py tests/test_zz_abi_v9_gpu.py -x -q -m gpu -k "turbo or nano or prefix or example or Turbo" 2>&1 | tail -4
timeout 600 python scripts/turbo_prefill_time.py 2>&1 | grep max_gen | tee gpurun_out/r06_ap/turbo_prefill_time.log
for i in 1 2; do for w in turbo nano; do
  timeout 500 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --workload $w --batch 1 > gpurun_out/r06_ap/${w}_$i.json 2> gpurun_out/r06_ap/${w}_$i.err
  python - "${w}_$i" <<'P'
import json, sys
d = json.loads(open(f"gpurun_out/r06_ap/{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d.get("stage_ms"), (d.get("decode_step") or {}).get("ms_per_step"), "first audio serial", d.get("p50_first_audio_latency_ms_serial"), flush=True)
P
done; done
