#!/bin/bash
# Round-end measurement set (one GPU call): the default bench line + rocprofv3 kernel stats of the same command.
# (PMC passes: scripts/pmc_run.sh -- separate runs, rocprofv3 --pmc does not survive hipGraph replays.)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final
cd $R
CBX_BENCH_VERBOSE=1 timeout 400 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
tail -1 gpurun_out/final/bench.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $R/bench.py > $R/gpurun_out/final/bench_under_rocprof.json 2> /tmp/rocprof_bench.err
cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) $R/gpurun_out/final/bench_kernel_stats.csv
head -14 $R/gpurun_out/final/bench_kernel_stats.csv | cut -c1-150
