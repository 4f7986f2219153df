#!/bin/bash
# Tenth GPU call: launch timeline of the GPT-2 decode step (Turbo, Nano at batch 1) and of the final Llama geometry (o / down separated).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/tenth
mkdir -p $O
cd $R
export CBX_LIB_PATH=$R/chatterbox_amd/build/libcbx_hip_trace.so
timeout 150 python scripts/trace_decode_turbo.py $O > $O/turbo.log 2>&1; grep -v amdgpu.ids $O/turbo.log | tail -30
CBX_TRACE_NANO=1 timeout 150 python scripts/trace_decode_turbo.py $O/nano > $O/nano.log 2>&1; grep -v amdgpu.ids $O/nano.log | tail -14
CBX_TRACE_VARIANTS=0 timeout 200 python scripts/trace_decode.py $O/llama > $O/llama.log 2>&1; grep -v amdgpu.ids $O/llama.log | tail -14
