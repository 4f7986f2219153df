#!/bin/bash
# round 6, call ba: write-through (sc1) stores in the S3Gen flow's kernels (plane GEMM epilogues, plane attention, narrow LayerNorm): ms per serial flow pass and the mel's digest,
# side library wtf1 against wt1 (same decode kernels, plain flow stores), interleaved processes on one box
for lib in wt1 wtf1 wt1 wtf1; do
  CBX_LABEL=$lib CBX_REPS=7 CBX_LIB_PATH=$PWD/chatterbox_amd/build/libcbx_hip_$lib.so timeout 200 python scripts/flow_ab.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/r06_ba_flow_store_policy_ab.log
done
