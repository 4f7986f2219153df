#!/bin/bash
# Fifth GPU call of round 4: the NEW default decode geometry (split-K column-tile q/k/v, head on 2-tile workgroups, pipelined attention, epilogue
# prefetch).  Order: allow-list from this hardware -> whole -m gpu suite (no -x) -> bench line -> PMC traffic of the decode kernels -> rocprofv3
# kernel stats of the bench command -> Turbo / Nano batch-1 lines with the old and the new launch knobs.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/fifth
mkdir -p $O
cd $R
CBX_GREEN_BUDGET_S=200 timeout 300 python scripts/green_variants.py $O > $O/green.log 2>&1; tail -2 $O/green.log
python scripts/write_green.py $O/green_variants.json "round 4, fifth GPU call (scripts/profile_r04_fifth.sh)"
cp chatterbox_amd/decode_green.json $O/decode_green.json
timeout 900 python -m pytest tests -q -m gpu -rfE --durations=8 -p no:cacheprovider --junitxml=$O/pytest_gpu.xml > $O/pytest_gpu.log 2>&1
tail -15 $O/pytest_gpu.log
CBX_BENCH_VERBOSE=1 timeout 400 python bench.py --gpus 1 --steps 10 --warmup 3 > $O/bench_steps10_warmup3.json 2> $O/bench.err; tail -c 600 $O/bench_steps10_warmup3.json
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  CBX_STEPS=6 timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_t3_$c -o p -- python $R/scripts/prof_t3_eager.py > /tmp/log_t3_$c.txt 2>&1
  f=$(find /tmp/pmc_t3_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py $f $O/t3_eager_pmc_$c.csv
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precisions --no-streaming > $O/bench_under_rocprof.json 2> /tmp/rocprof_bench.err
cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) $O/bench_steps5_warmup2_kernel_stats.csv
head -12 $O/bench_steps5_warmup2_kernel_stats.csv | cut -c1-170
cd $R
for w in "turbo" "nano"; do
  for e in "CBX_DA_PIPE=0 CBX_GEMV_PRE_EPI=0" "X=1"; do
    env $e timeout 300 python bench.py --workload $w --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precisions --no-streaming 2> /dev/null | tail -1 > $O/bench_${w}_b1_$(echo $e | tr ' =' '__').json
    python -c "import json,sys; d=json.load(open('$O/bench_${w}_b1_$(echo $e | tr ' =' '__').json')); print('$w b1 [$e]', d['value'], d['config'].get('stage_ms_per_step'), d.get('decode_step', {}).get('ms_per_step'), d.get('decode_step', {}).get('frac'))"
  done
done
