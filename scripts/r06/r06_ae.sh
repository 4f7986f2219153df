#!/bin/bash
# round 6, call ae: LayerNorm from the GEMM epilogue (fused_ln 1 / 2: loses at batch 8) at batch 1 / 2, where the two LayerNorm launches per block are 16 % of a block's time
mkdir -p gpurun_out/r06_ae
for b in 1 2 8; do for f in 0 1 2 0 2; do CBX_FUSED_LN=$f CBX_B=$b CBX_LABEL="B$b fused_ln=$f" CBX_REPS=7 timeout 300 python scripts/flow_ab.py 2>&1 | grep "flow ms" | tee -a gpurun_out/r06_ae/flow_fused_ln.log; done; done
