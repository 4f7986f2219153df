#!/bin/bash
# Round 5, call S1 (on the round's final kernels / defaults): PMC passes (FETCH_SIZE, WRITE_SIZE) over the flow kernels and the eager decode step, rocprofv3 kernel stats of the serial schedule
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/s1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_flow_$c -o p -- python $R/scripts/flow_only.py > /tmp/log_flow_$c.txt 2>&1
  f=$(find /tmp/pmc_flow_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py $f $O/flow_only_pmc_$c.csv || tail -3 /tmp/log_flow_$c.txt
  CBX_STEPS=6 timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_t3_$c -o p -- python $R/scripts/prof_t3_eager.py > /tmp/log_t3_$c.txt 2>&1
  f=$(find /tmp/pmc_t3_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py $f $O/t3_eager_pmc_$c.csv || tail -3 /tmp/log_t3_$c.txt
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $R/bench.py --schedule serial --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precisions --no-streaming > $O/bench_serial_under_rocprof.json 2> /tmp/rocprof_bench.err
cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) $O/bench_serial_steps5_kernel_stats.csv
head -14 $O/bench_serial_steps5_kernel_stats.csv | cut -c1-200 | sed 's/(anonymous namespace):://g'
python -c "import json; d=json.load(open('$O/bench_serial_under_rocprof.json')); print('serial under rocprof', d['value'], d['ms_per_step'], d['stage_ms'])"
