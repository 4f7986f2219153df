#!/bin/bash
# round 6, call ax: scripts/micro/mall_prefetch.hip in its per-op mode -- what a PLAIN non-temporal stream of each decode op's bytes costs inside a dependent hipGraph chain
# (the floor the real kernels are measured against), per launch geometry
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/micro/mall_prefetch.hip -o /tmp/mall_prefetch || exit 1
timeout 240 /tmp/mall_prefetch 20 ops 2>&1 | tee gpurun_out/r06_ax_plain_stream_floors.log
