#!/bin/bash
# Round-end measurement set of round 6 (one GPU call; results copied to profiles/r06_*):
#  1. (after 3, so that its roofline.traffic reads this build's PMC passes) the driver's command `bench.py --steps 20 --warmup 5`
#  2. rocprofv3 --kernel-trace --stats of the serial schedule (per-kernel averages with the GPU to one stage at a time)
#  3. PMC passes (separate FETCH_SIZE / WRITE_SIZE runs, kernel-trace only) over the flow pass, the eager Llama decode and the eager Turbo decode
#  4. the other workloads: Turbo / Nano at batch 1, Multilingual at B = 1 / 32, configs[3] on one GPU, configs[4] (60 s voice conversion)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_final
mkdir -p $O
cd $R
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_serial -o b -- python $R/bench.py --schedule serial --steps 5 --warmup 2 --no-cpu-baseline --no-streaming --no-alt-precisions > $O/bench_serial_under_rocprof.json 2> /tmp/rocprof_serial.err
cp $(find /tmp/prof_serial -name "*kernel_stats.csv" | head -1) $O/bench_serial_steps5_kernel_stats.csv
head -12 $O/bench_serial_steps5_kernel_stats.csv | cut -c1-170
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_turbo -o t -- python $R/bench.py --workload turbo --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-streaming --no-parity > $O/bench_turbo_b1_under_rocprof.json 2> /tmp/rocprof_turbo.err
cp $(find /tmp/prof_turbo -name "*kernel_stats.csv" | head -1) $O/turbo_b1_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_flow_$c -o p -- python $R/scripts/flow_only.py > /tmp/log_flow_$c.txt 2>&1
  f=$(find /tmp/pmc_flow_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python $R/scripts/pmc_summary.py $f $O/flow_only_pmc_$c.csv
  CBX_STEPS=6 timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_t3_$c -o p -- python $R/scripts/prof_t3_eager.py > /tmp/log_t3_$c.txt 2>&1
  f=$(find /tmp/pmc_t3_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python $R/scripts/pmc_summary.py $f $O/t3_eager_pmc_$c.csv
  CBX_STEPS=8 timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_tb_$c -o p -- python $R/scripts/prof_turbo_eager.py > /tmp/log_tb_$c.txt 2>&1
  f=$(find /tmp/pmc_tb_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python $R/scripts/pmc_summary.py $f $O/turbo_eager_pmc_$c.csv
done
cd $R
# the bench line reads roofline.traffic from profiles/r06_*_pmc_*.csv: the passes above, on this build
for f in $O/*_pmc_*.csv; do cp $f profiles/r06_$(basename $f); done
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench_steps20_warmup5.err
tail -1 $O/bench_steps20_warmup5.json | cut -c1-300
ls $O
for spec in "turbo_b1:--workload turbo --batch 1" "nano_b1:--workload nano --batch 1" "mtl_b1:--batch 1" "mtl_b32:--batch 32 --steps 3 --warmup 1" "config3:--config3 --steps 4 --warmup 1" "vc60_b1:--workload vc60 --batch 1" "vc60_b4:--workload vc60 --batch 4 --steps 3 --warmup 1"; do
  name=${spec%%:*}; flags=${spec#*:}
  timeout 500 python bench.py --steps 6 --warmup 2 --no-cpu-baseline $flags > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d.get("stage_ms"), (d.get("decode_step") or {}).get("ms_per_step"), (d.get("decode_step") or {}).get("frac"))
except Exception as e:
    print("$name FAILED", e)
PY
done
