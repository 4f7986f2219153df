#!/bin/bash
# Where does the split GEMM's time go?  Builds gemm_split.hip with -DCBX_DIAG (runtime switches in cbx_gemm_t.reserved0: 1 = no global
# loads after the prologue, 2 = no conversion arithmetic, 4 = no LDS reads / MFMA, 8 = no LDS stores) into a side library and times the
# CFM shapes with parts of the K loop switched off.  Run HERE to build (hipcc, no GPU), then on the GPU box: scripts/diag_gemm.sh run
set -e
cd "$(dirname "$0")/.."
if [ "$1" != "run" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCBX_DIAG -c chatterbox_amd/csrc/gemm_split.hip -o chatterbox_amd/build/gemm_split_diag.o
  objs=$(ls chatterbox_amd/build/*.hip.o | grep -v gemm_split.hip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs chatterbox_amd/build/gemm_split_diag.o -o chatterbox_amd/build/libcbx_hip_diag.so
  echo built chatterbox_amd/build/libcbx_hip_diag.so
  exit 0
fi
export CBX_LIB_PATH=$PWD/chatterbox_amd/build/libcbx_hip_diag.so
for d in 0 1 2 3 4 8 10 11 15; do
  echo "== diag $d (1 no loads, 2 no convert, 4 no mfma, 8 no lds stores)"
  CBX_DIAG=$d CBX_PRECS="${CBX_PRECS:-16,6}" CBX_GEMM_SHAPES=qkv,ff2,conv3_320,enc_ff2,big python scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids
done
