#!/bin/bash
# Round 5, call R: measurement set on the round's kernels: (1) PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) over the flow kernels and the eager decode step
# -> roofline.traffic of the bench line; (2) rocprofv3 kernel stats of the SERIAL schedule; (3) the other workloads (Turbo / Nano / Multilingual at batch 1, B = 32, configs[3])
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/r
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
ls -la $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $R/bench.py --schedule serial --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precisions --no-streaming > $O/bench_serial_under_rocprof.json 2> /tmp/rocprof_bench.err
cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) $O/bench_serial_steps5_kernel_stats.csv
head -12 $O/bench_serial_steps5_kernel_stats.csv | cut -c1-200 | sed 's/(anonymous namespace):://g'
cd $R
for f in 1 0; do  # LayerNorm of norm3 from the out-projection's epilogue (the default) against LayerNorm launches, serial schedule, same box
  CBX_FUSED_LN=$f timeout 200 python bench.py --schedule serial --steps 6 --warmup 2 --no-cpu-baseline --no-alt-precisions --no-streaming --no-autotune > $O/bench_serial_fused_ln_$f.json 2> $O/bench_ln_$f.err
  python -c "import json; d=json.load(open('$O/bench_serial_fused_ln_$f.json')); print('serial, fused_ln $f:', d['value'], d['ms_per_step'], d['stage_ms'])"
done
for w in turbo nano; do
  timeout 150 python bench.py --workload $w --batch 1 --steps 8 --warmup 2 --no-cpu-baseline --no-alt-precisions --no-streaming > $O/bench_${w}_b1.json 2> $O/bench_${w}.err
  python -c "import json; d=json.load(open('$O/bench_${w}_b1.json')); print('$w b1', d['value'], d['config'].get('stage_ms_per_step'), d.get('decode_step', {}).get('ms_per_step'), d.get('decode_step', {}).get('frac'))"
done
timeout 150 python bench.py --batch 1 --steps 6 --warmup 2 --no-cpu-baseline --no-alt-precisions --no-streaming > $O/bench_mtl_b1.json 2> $O/bench_mtl_b1.err
python -c "import json; d=json.load(open('$O/bench_mtl_b1.json')); print('mtl b1', d['value'], d['schedule'], d.get('other_schedule',{}).get('value'), d['stage_ms'])"
timeout 200 python bench.py --batch 32 --steps 4 --warmup 2 --no-cpu-baseline --no-alt-precisions --no-streaming --no-autotune > $O/bench_mtl_b32.json 2> $O/bench_mtl_b32.err
python -c "import json; d=json.load(open('$O/bench_mtl_b32.json')); print('mtl b32', d['value'], d['schedule'], d.get('other_schedule',{}).get('value'), d['stage_ms'])"
