cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_hbm
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/scripts/flow_only.py > /tmp/log_$c.txt 2>&1
  grep -v "simple_timer" /tmp/log_$c.txt | tail -2 | cut -c1-200
  f=$(ls /tmp/pmc_$c/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py $f $R/gpurun_out/pmc_hbm/$c.csv
done
head -4 $R/gpurun_out/pmc_hbm/FETCH_SIZE.csv; head -4 $R/gpurun_out/pmc_hbm/WRITE_SIZE.csv
