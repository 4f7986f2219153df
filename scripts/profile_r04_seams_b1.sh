#!/bin/bash
# Round 4, the last GPU seconds: the stage seams at batch 1 (Turbo), where the flow is host-bound -- seams off / on on one box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/seams_b1
mkdir -p $O
cd $R
for tag in off on; do
  s=0; [ $tag = on ] && s=1
  CBX_FLOW_CSEAM=$s CBX_HIFT_CSEAM=$s timeout 70 python bench.py --workload turbo --batch 1 --steps 8 --warmup 2 --no-cpu-baseline --no-alt-precisions --no-streaming --no-autotune > $O/bench_turbo_b1_$tag.json 2> $O/bench_$tag.err
  python -c "import json; d=json.load(open('$O/bench_turbo_b1_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['config']['stage_seams'])"
done
