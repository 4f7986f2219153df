#!/bin/bash
# Round 5, call N: rocprofv3 kernel stats of the bench in its throughput schedule (the profiler serialises the two streams' kernels: durations are
# "alone on the chip", but of the CO-RESIDENT forms) -- what the flow + vocoder stream costs per kernel class in that mode
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/n
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $R/bench.py --steps 4 --warmup 2 --no-alt-precisions --no-streaming --no-cpu-baseline --no-autotune > $O/bench_under_rocprof.json 2> /tmp/rocprof_bench.err
tail -2 /tmp/rocprof_bench.err | cut -c1-200
cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) $O/bench_pipelined_kernel_stats.csv
head -30 $O/bench_pipelined_kernel_stats.csv | cut -c1-220 | sed 's/(anonymous namespace):://g'
