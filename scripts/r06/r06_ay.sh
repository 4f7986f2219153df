#!/bin/bash
# round 6, call ay: rocprofv3 --kernel-trace --stats of the DRIVER'S command shape (the throughput schedule: two T3 chains beside flow + vocoder on co-resident kernel forms),
# beside final.sh's serial-schedule statistics: per-kernel averages as they are INSIDE the headline's timed region
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_ay
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pipe -o b -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-streaming --no-alt-precisions > $O/bench_pipelined_under_rocprof.json 2> /tmp/rocprof_pipe
tail -3 /tmp/rocprof_pipe | cut -c1-300
cp $(find /tmp/prof_pipe -name "*kernel_stats.csv" | head -1) $O/bench_pipelined_steps10_kernel_stats.csv
head -14 $O/bench_pipelined_steps10_kernel_stats.csv | cut -c1-200
tail -1 $O/bench_pipelined_under_rocprof.json | cut -c1-400
