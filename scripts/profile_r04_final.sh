#!/bin/bash
# Round-4 final set (one GPU call, on the round's last kernel commit): whole -m gpu suite (no -x), the driver's bench command, rocprofv3 kernel stats
# of a bench run, the other workloads' lines.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/final
mkdir -p $O
cd $R
timeout 1000 python -m pytest tests -q -m gpu -rfE --durations=6 -p no:cacheprovider --junitxml=$O/pytest_gpu.xml > $O/pytest_gpu.log 2>&1
tail -12 $O/pytest_gpu.log
CBX_BENCH_VERBOSE=1 timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench.err
python -c "import json; d=json.load(open('$O/bench_steps20_warmup5.json')); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['t3_geometry'], d['decode_step']['ms_per_step'], d['decode_step']['frac'], d['roofline']['frac'], d.get('pipelined_schedule'), d['parity'])"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precisions --no-streaming > $O/bench_under_rocprof.json 2> /tmp/rocprof_bench.err
cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) $O/bench_steps5_warmup2_kernel_stats.csv
head -10 $O/bench_steps5_warmup2_kernel_stats.csv | cut -c1-150
cd $R
for w in "--batch 1:mtl_b1" "--batch 32 --steps 3 --warmup 1:mtl_b32" "--config3 --steps 2 --warmup 1:config3"; do
  flags=${w%%:*}; tag=${w##*:}
  timeout 400 python bench.py $flags --no-cpu-baseline --no-alt-precisions --no-streaming > $O/bench_$tag.json 2> $O/bench_$tag.err
  python -c "import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['config'].get('stage_ms_per_step'), d.get('configs3'))"
done
