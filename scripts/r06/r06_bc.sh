#!/bin/bash
# round 6, call bc: the PMC passes over the eager Llama decode on the write-through build (roofline.traffic of the bench line reads profiles/r06_t3_eager_pmc_*.csv)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_bc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  CBX_STEPS=6 timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_t3_$c -o p -- python $R/scripts/prof_t3_eager.py > /tmp/log_t3_$c.txt 2>&1
  f=$(find /tmp/pmc_t3_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python $R/scripts/pmc_summary.py $f $O/t3_eager_pmc_$c.csv
  head -8 $O/t3_eager_pmc_$c.csv | cut -c1-200
done
