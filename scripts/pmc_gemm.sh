#!/bin/bash
# SQ counters of the split-bf16 GEMM on two CFM shapes (where does a K tile's time go: MFMA, LDS, VALU, waiting?).
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_gemm
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  tag=$(echo $set | cut -d' ' -f1)
  CBX_REPS=5 CBX_PRECS="${PMC_PRECS:-6:0}" CBX_GEMM_SHAPES=ff2,qkv timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_gemm_$tag -o p -- python $R/scripts/bench_gemm.py > /tmp/pmc_gemm_$tag.log 2>&1
  f=$(find /tmp/pmc_gemm_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py $f $R/gpurun_out/pmc_gemm/$tag.csv
  tail -2 /tmp/pmc_gemm_$tag.log
done
grep -h "gemm_split" $R/gpurun_out/pmc_gemm/*.csv | cut -c1-60,150-400
