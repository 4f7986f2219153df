#!/bin/bash
# round 6, call bd: scripts/micro/store_policy.hip -- what the bytes a launch writes cost the next launch of a dependent chain, per store policy (plain / sc1 / sc0 sc1 / nt)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/micro/store_policy.hip -o /tmp/store_policy || exit 1
timeout 200 /tmp/store_policy 20 2>&1 | tee gpurun_out/r06_bd_store_policy_micro.log
