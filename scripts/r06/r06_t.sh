#!/bin/bash
# round 6, call t: side-library experiment -- q | k | v on a loader-wave form (35 plain / 42 deferred) INSIDE the throughput schedule instead of the co-resident form 8; A / B / C / A / B / C on one box
mkdir -p gpurun_out/r06_t
run() { # name, extra env
  env $2 timeout 900 python bench.py --steps 8 --warmup 3 > gpurun_out/r06_t/$1.json 2> gpurun_out/r06_t/$1.err
  python - "$1" <<'P'
import json, sys
d = json.load(open(f"gpurun_out/r06_t/{sys.argv[1]}.json"))
print(sys.argv[1], "pipelined", d.get("value_pipelined"), "serial", d.get("value_serial"), "stage_ms", d.get("stage_ms"), flush=True)
P
}
for i in 1 2; do
run a$i "CBX_NONE=1"
run b$i "CBX_LIB_PATH=$PWD/chatterbox_amd/build/libcbx_hip_exp35.so"
run c$i "CBX_LIB_PATH=$PWD/chatterbox_amd/build/libcbx_hip_exp42.so"
done
