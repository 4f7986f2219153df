#!/bin/bash
# round 6, call u: the serial flow with this round's automatic plane-GEMM forms (a) against the round-5 choice (b: side library), A / B x 3 on one box
mkdir -p gpurun_out/r06_u
for i in 1 2 3; do
CBX_LABEL=a$i timeout 300 python scripts/flow_ab.py 2>&1 | grep "flow ms" | tee -a gpurun_out/r06_u/flow_ab.log
CBX_LABEL=b$i CBX_LIB_PATH=$PWD/chatterbox_amd/build/libcbx_hip_nodf.so timeout 300 python scripts/flow_ab.py 2>&1 | grep "flow ms" | tee -a gpurun_out/r06_u/flow_ab.log
done
