#!/bin/bash
# round 6, call aw: scripts/micro/mall_prefetch.hip -- does warming the Infinity Cache from a parallel hipGraph branch shorten a decode-shaped chain of dependent streaming launches?
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/micro/mall_prefetch.hip -o /tmp/mall_prefetch || exit 1
timeout 240 /tmp/mall_prefetch 20 2>&1 | tee gpurun_out/r06_aw_mall_prefetch.log
