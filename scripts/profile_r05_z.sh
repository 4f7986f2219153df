#!/bin/bash
# Round 5, call Z: clocks and power while the bench runs its serial and its throughput schedule (VERDICT r04 item 6 asked: "rocm-smi clocks / power during the overlap")
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/z
mkdir -p $O
cd $R
sample() { # $1 = output file; samples until the file $1.stop exists
  while [ ! -e $1.stop ]; do
    echo "t $(date +%s.%N) $(rocm-smi --showpower --showclocks --showuse --csv 2>/dev/null | tail -n +2 | tr '\n' ' ')" >> $1
    sleep 0.25
  done
}
for sch in serial pipelined; do
  rm -f $O/smi_$sch.log $O/smi_$sch.log.stop
  sample $O/smi_$sch.log &
  SP=$!
  timeout 300 python bench.py --schedule $sch --steps 12 --warmup 3 --no-alt-precisions --no-streaming --no-cpu-baseline --no-autotune > $O/bench_$sch.json 2> $O/bench_$sch.err
  touch $O/smi_$sch.log.stop
  wait $SP
  python -c "import json; d=json.load(open('$O/bench_$sch.json')); print('$sch', d['value'], d['ms_per_step'])"
  tail -2 $O/smi_$sch.log | cut -c1-300
done
rocm-smi --showpower --showclocks --showuse --csv 2>/dev/null | head -3 | cut -c1-300
