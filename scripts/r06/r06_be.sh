#!/bin/bash
# round 6, call be: scripts/micro/store_policy.hip in its hand-off mode -- a bare weight stream + a 64 KB image written by every launch and read whole by every workgroup of the next
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/micro/store_policy.hip -o /tmp/store_policy || exit 1
timeout 200 /tmp/store_policy 20 handoff 2>&1 | tee gpurun_out/r06_be_handoff_micro.log
