#!/bin/bash
# Round-4 final set, second edition (after the x-lane clamp and the LayerNorm form of the column-tile GEMV): whole -m gpu suite, the driver's bench command,
# batch-1 lines.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/final2
mkdir -p $O
cd $R
timeout 1000 python -m pytest tests -q -m gpu -rfE -p no:cacheprovider --junitxml=$O/pytest_gpu.xml > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
CBX_BENCH_VERBOSE=1 timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench.err
python -c "import json; d=json.load(open('$O/bench_steps20_warmup5.json')); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['t3_geometry'], d['decode_step']['ms_per_step'], d['decode_step']['frac'], d['roofline']['frac'], d.get('pipelined_schedule'), d['parity'])"
for w in "--workload turbo --batch 1:turbo_b1" "--workload nano --batch 1:nano_b1"; do
  flags=${w%%:*}; tag=${w##*:}
  timeout 300 python bench.py $flags --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precisions --no-streaming > $O/bench_$tag.json 2> $O/bench_$tag.err
  python -c "import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['config'].get('stage_ms_per_step'), d.get('decode_step', {}).get('ms_per_step'), d.get('decode_step', {}).get('frac'))"
done
