#!/bin/bash
# round 6, call az: decode-store policy A/B -- plain stores (the shipped library) against write-through sc1 / sc0 sc1 / nt stores of the decode kernels' outputs
# (scripts/wt_build.sh side libraries), interleaved processes on one box
for lib in "" wt1 wt2 wt3 "" wt1 wt2 wt3; do
  CBX_LIB_PATH=${lib:+$PWD/chatterbox_amd/build/libcbx_hip_$lib.so} timeout 200 python scripts/wt_ab.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/r06_az_decode_store_policy_ab.log
done
