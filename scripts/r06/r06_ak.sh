#!/bin/bash
# round 6, call ak: the cached conditioning prefix of a voice in the T3 prefill (ABI v15: cbx_flash_attn_kv_f32): parity, then the bench line (serial + throughput schedule)
# with it (a) and without (b: CBX_T3_SHARE_PREFIX=0), A / B / A / B on one box
mkdir -p gpurun_out/r06_ak
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_zz_abi_v9_gpu.py tests/test_models_gpu.py tests/test_baseline_shapes_gpu.py tests/test_stream_gpu.py -x -q -m gpu -k "flash_attn or prefix or prefill or t3 or T3 or pipelined or stream or e2e" 2>&1 | tail -4
run() {
  env $2 timeout 900 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r06_ak/$1.json 2> gpurun_out/r06_ak/$1.err
  python - "$1" <<'P'
import json, sys
d = json.load(open(f"gpurun_out/r06_ak/{sys.argv[1]}.json"))
print(sys.argv[1], "pipelined", d.get("value_pipelined"), "serial", d.get("value_serial"), "stage_ms", d.get("stage_ms"), "first audio serial", d.get("p50_first_audio_latency_ms_serial"), "parity", (d.get("parity") or {}).get("tokens_equal"), flush=True)
P
}
for i in 1 2; do
run a$i "CBX_NONE=1"
run b$i "CBX_T3_SHARE_PREFIX=0"
done
