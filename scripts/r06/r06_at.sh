#!/bin/bash
# round 6, call at: cbx_gemm_f32 on 64-wide K tiles where a handful of workgroups walk a long K (the GPT-2 backbones' prefill at batch 1), on top of cbx_gpt2_prefill + the cached
# conditioning prefix: parity, prefill wall / GPU time, Turbo / Nano at batch 1
mkdir -p gpurun_out/r06_at
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py tests/test_baseline_shapes_gpu.py tests/test_examples_gpu.py tests/test_zz_abi_v9_gpu.py -x -q -m gpu -k "linear or turbo or nano or prefix or prefill or example or t3 or T3" 2>&1 | tail -4
timeout 300 python scripts/turbo_prefill_profile.py 2>&1 | grep -v amdgpu | tail -3 | cut -c1-200 | tee gpurun_out/r06_at/turbo_prefill_profile.log
timeout 600 python scripts/turbo_prefill_time.py 2>&1 | grep max_gen | tee gpurun_out/r06_at/turbo_prefill_time.log
for i in 1 2; do for w in turbo nano; do
  timeout 500 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --workload $w --batch 1 > gpurun_out/r06_at/${w}_$i.json 2> gpurun_out/r06_at/${w}_$i.err
  python - "${w}_$i" <<'P'
import json, sys
d = json.loads(open(f"gpurun_out/r06_at/{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d.get("stage_ms"), (d.get("decode_step") or {}).get("ms_per_step"), "first audio serial", d.get("p50_first_audio_latency_ms_serial"), flush=True)
P
done; done
