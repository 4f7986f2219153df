#!/bin/bash
# Round 4, the last GPU seconds: the fused feed-forward launch (cbx_mlp_planes, opt-in) at batch 1 (Turbo, Multilingual), where the flow's GEMMs are small.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/b1_fused_mlp
mkdir -p $O
cd $R
for w in "--workload turbo --batch 1:turbo_b1" "--batch 1:mtl_b1"; do
  flags=${w%%:*}; tag=${w##*:}
  for m in 0 1; do
    CBX_FUSED_MLP=$m timeout 50 python bench.py $flags --steps 6 --warmup 2 --no-cpu-baseline --no-alt-precisions --no-streaming --no-autotune > $O/bench_${tag}_mlp$m.json 2> $O/bench_${tag}_$m.err
    python -c "import json; d=json.load(open('$O/bench_${tag}_mlp$m.json')); print('$tag fused_mlp=$m', d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'])"
  done
done
