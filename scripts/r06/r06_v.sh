#!/bin/bash
# round 6, call v: side-library experiment -- EVERY plane GEMM of the throughput schedule on its serial-schedule form (d) against the co-resident choice (a); A / D / A / D on one box
mkdir -p gpurun_out/r06_v
run() { # name, extra env
  env $2 timeout 900 python bench.py --steps 8 --warmup 3 > gpurun_out/r06_v/$1.json 2> gpurun_out/r06_v/$1.err
  python - "$1" <<'P'
import json, sys
d = json.load(open(f"gpurun_out/r06_v/{sys.argv[1]}.json"))
print(sys.argv[1], "pipelined", d.get("value_pipelined"), "serial", d.get("value_serial"), "stage_ms", d.get("stage_ms"), flush=True)
P
}
for i in 1 2; do
run a$i "CBX_NONE=1"
run d$i "CBX_LIB_PATH=$PWD/chatterbox_amd/build/libcbx_hip_exp.so"
done
