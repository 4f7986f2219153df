#!/bin/bash
# Where does the time of the plane-format GEMM / attention go?  Builds gemm_planes.hip and attention_planes.hip with -DCBX_DIAG (runtime switches:
# GEMM cbx_gemm_pl_t.reserved0 = 1 no DMA after the prologue | 2 no ds_read / MFMA | 4 no epilogue stores; attention CBX_ATTN_DIAG = 1 no DMA in
# the loop | 2 no S MFMAs | 4 no softmax | 8 no PV MFMAs | 16 no P split) into a side library and times the CFM shapes with parts switched off.
# Run HERE to build (hipcc, no GPU), then on the GPU box: scripts/diag_planes.sh run
set -e
cd "$(dirname "$0")/.."
if [ "$1" != "run" ]; then
  for f in gemm_planes attention_planes; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCBX_DIAG -c chatterbox_amd/csrc/$f.hip -o chatterbox_amd/build/${f}_diag.o
  done
  objs=$(ls chatterbox_amd/build/*.hip.o | grep -v gemm_planes.hip.o | grep -v attention_planes.hip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs chatterbox_amd/build/gemm_planes_diag.o chatterbox_amd/build/attention_planes_diag.o -o chatterbox_amd/build/libcbx_hip_diag.so
  echo built chatterbox_amd/build/libcbx_hip_diag.so
  exit 0
fi
export CBX_LIB_PATH=$PWD/chatterbox_amd/build/libcbx_hip_diag.so
python scripts/diag_planes.py 2>&1 | grep -v amdgpu.ids
