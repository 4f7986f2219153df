#!/bin/bash
# Launch timeline of the T3 decode step (cbx_common.h CBX_TRACE): a SIDE build of the decode kernels whose workgroups stamp the chip-wide 100 MHz
# counter at their phase boundaries.  Run HERE to build (hipcc, no GPU):  scripts/trace_decode.sh
# then on the GPU box:  scripts/trace_decode.sh run [out_dir]   (python scripts/trace_decode.py does the run and the analysis)
set -e
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -ffp-contract=on"
if [ "$1" != "run" ]; then
  python -m chatterbox_amd.build > /dev/null
  for f in gemv_decode attention sampler elementwise t3_step; do
    /opt/rocm/bin/hipcc $FLAGS -DCBX_TRACE -c chatterbox_amd/csrc/$f.hip -o chatterbox_amd/build/${f}_trace.o &
  done
  wait
  objs=$(ls chatterbox_amd/build/*.hip.o | grep -v -e gemv_decode.hip.o -e attention.hip.o -e sampler.hip.o -e elementwise.hip.o -e t3_step.hip.o -e gemv_pair.hip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs chatterbox_amd/build/gemv_decode_trace.o chatterbox_amd/build/attention_trace.o \
      chatterbox_amd/build/sampler_trace.o chatterbox_amd/build/elementwise_trace.o chatterbox_amd/build/t3_step_trace.o -o chatterbox_amd/build/libcbx_hip_trace.so
  echo built chatterbox_amd/build/libcbx_hip_trace.so
  exit 0
fi
export CBX_LIB_PATH=$PWD/chatterbox_amd/build/libcbx_hip_trace.so
python scripts/trace_decode.py "${2:-gpurun_out/r04/trace}" 2>&1 | grep -v amdgpu.ids
