#!/bin/bash
# tile sweep of the split-bf16 GEMM on the bench-shape flow pass (one GPU call, same box)
mkdir -p gpurun_out; : > gpurun_out/sweep.log
for t in 128 12864 64; do
  CBX_SPLIT_TILE=$t timeout 120 python scripts/split_eval.py speed 2>&1 | grep -v "^prec 1" >> gpurun_out/sweep.log
done
cd /tmp && export TMPDIR=/tmp
CBX_SPLIT_TILE=12864 CBX_S3GEN_PRECISION=3 CBX_REPS=2 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_split -o split -- python $GRAFT_REPO_ROOT/scripts/flow_only.py > $GRAFT_REPO_ROOT/gpurun_out/prof_split.log 2>&1
cp $(find /tmp/prof_split -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/split_kernel_stats.csv
cp $(find /tmp/prof_split -name "*kernel_trace.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/split_kernel_trace.csv
cat $GRAFT_REPO_ROOT/gpurun_out/sweep.log; head -8 $GRAFT_REPO_ROOT/gpurun_out/split_kernel_stats.csv | cut -c1-180
