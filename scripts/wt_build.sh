#!/bin/bash
# Side libraries for the decode-store policy A/B (cbx_common.h CBX_WT): gemv_decode.hip and attention.hip compiled with -DCBX_WT=0 (plain stores), 2 (sc0 sc1), 3 (nt) -- the shipped library is 1 (sc1) --
# linked with the other objects of the in-tree build -> chatterbox_amd/build/libcbx_hip_wt{0,2,3}.so.  Run HERE (hipcc, no GPU); select one with CBX_LIB_PATH on the GPU box.
set -e
cd "$(dirname "$0")/.."
python -c "from chatterbox_amd import build; build.build(verbose=False)"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -ffp-contract=on"
for n in ${WT_MODES:-0 2 3}; do
  for f in gemv_decode attention; do
    /opt/rocm/bin/hipcc $FLAGS -DCBX_WT=$n -c chatterbox_amd/csrc/$f.hip -o chatterbox_amd/build/${f}_wt$n.o &
  done
done
wait
for n in ${WT_MODES:-0 2 3}; do
  objs=$(ls chatterbox_amd/build/*.hip.o | grep -v "/gemv_decode.hip.o" | grep -v "/attention.hip.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs chatterbox_amd/build/gemv_decode_wt$n.o chatterbox_amd/build/attention_wt$n.o -o chatterbox_amd/build/libcbx_hip_wt$n.so
  echo built chatterbox_amd/build/libcbx_hip_wt$n.so
done
