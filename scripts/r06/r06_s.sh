#!/bin/bash
# round 6, call s: (1) the automatic plane-GEMM choice of this build in situ, both schedules; (2) side-library experiment: the deferred GELU form (12 waves, 168 VGPRs) for ff1
# INSIDE the throughput schedule, where the flow otherwise runs on co-resident forms -- same box, A / B / A
mkdir -p gpurun_out/r06_s
run() { # name, extra env
  env $2 timeout 900 python bench.py --steps 8 --warmup 3 > gpurun_out/r06_s/$1.json 2> gpurun_out/r06_s/$1.err
  python - "$1" <<'P'
import json, sys
d = json.load(open(f"gpurun_out/r06_s/{sys.argv[1]}.json"))
g = [r for r in d.get("roofline_secondary", []) if "gemm_pl" in r["kernel"]]
print(sys.argv[1], "pipelined", d.get("value_pipelined"), "serial", d.get("value_serial"), "stage_ms", d.get("stage_ms"), "gemm_pl", g and (g[0]["frac"], g[0]["avg_launch_us"]), flush=True)
P
}
run a1 "CBX_NONE=1"
run b1 "CBX_LIB_PATH=$PWD/chatterbox_amd/build/libcbx_hip_exp.so"
run a2 "CBX_NONE=1"
run b2 "CBX_LIB_PATH=$PWD/chatterbox_amd/build/libcbx_hip_exp.so"
