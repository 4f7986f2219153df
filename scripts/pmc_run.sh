#!/bin/bash
# HBM traffic counters (separate FETCH_SIZE / WRITE_SIZE passes, kernel-trace only) for the S3Gen pass and an eager T3 decode.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_flow_$c -o p -- python $R/scripts/flow_only.py > /tmp/log_flow_$c.txt 2>&1
  f=$(find /tmp/pmc_flow_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py $f $R/gpurun_out/final/flow_only_pmc_$c.csv
  CBX_STEPS=6 timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_t3_$c -o p -- python $R/scripts/prof_t3_eager.py > /tmp/log_t3_$c.txt 2>&1
  f=$(find /tmp/pmc_t3_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py $f $R/gpurun_out/final/t3_eager_pmc_$c.csv
done
ls -la $R/gpurun_out/final
