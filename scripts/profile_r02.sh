#!/bin/bash
# Round-2 measurement set (one GPU call, ~6 minutes): the driver's bench command, rocprofv3 kernel stats of the SAME command, and the
# HBM-traffic PMC passes (separate FETCH_SIZE / WRITE_SIZE runs, kernel-trace only) of the S3Gen pass and an eager T3 decode.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
CBX_BENCH_VERBOSE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench.err
tail -1 $O/bench_steps20_warmup5.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> /tmp/rocprof_bench.err
cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) $O/bench_steps20_warmup5_kernel_stats.csv
head -12 $O/bench_steps20_warmup5_kernel_stats.csv | cut -c1-170
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_flow_$c -o p -- python $R/scripts/flow_only.py > /tmp/log_flow_$c.txt 2>&1
  f=$(find /tmp/pmc_flow_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py $f $O/flow_only_pmc_$c.csv
  CBX_STEPS=6 timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_t3_$c -o p -- python $R/scripts/prof_t3_eager.py > /tmp/log_t3_$c.txt 2>&1
  f=$(find /tmp/pmc_t3_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py $f $O/t3_eager_pmc_$c.csv
done
ls -la $O
