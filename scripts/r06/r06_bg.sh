#!/bin/bash
# round 6, call bg: scripts/micro/gemv_anatomy.hip -- the bare stream + hand-off with the decode GEMV's parts added one at a time
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/micro/gemv_anatomy.hip -o /tmp/gemv_anatomy || exit 1
timeout 200 /tmp/gemv_anatomy 20 2>&1 | tee gpurun_out/r06_bg_gemv_anatomy_micro.log
