#!/bin/bash
# round 6, call bb: the library with write-through (sc1) stores in the decode kernels -- full GPU suite, smoke, the driver's bench command, rocprofv3 kernel statistics of the serial schedule
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_bb
mkdir -p $O
cd $R
timeout 1000 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r06_gpu_tests_final.txt; cat gpurun_out/r06_gpu_tests_final.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench_steps20_warmup5.err
tail -1 $O/bench_steps20_warmup5.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_serial -o b -- python $R/bench.py --schedule serial --steps 5 --warmup 2 --no-cpu-baseline --no-streaming --no-alt-precisions > $O/bench_serial_under_rocprof.json 2> /tmp/rocprof_ser
cp $(find /tmp/prof_serial -name "*kernel_stats.csv" | head -1) $O/bench_serial_steps5_kernel_stats.csv
head -8 $O/bench_serial_steps5_kernel_stats.csv | cut -c1-170
