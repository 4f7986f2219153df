#!/bin/bash
# Fourteenth GPU call: lower split thresholds of the decode attention (Turbo and Llama at batch 1, 250 tokens).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/fourteenth
mkdir -p $O
cd $R
for sm in 512 384 256 128 512; do
  CBX_DA_SPLIT_MIN=$sm timeout 200 python bench.py --workload turbo --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-alt-precisions --no-streaming --no-autotune 2> /dev/null | tail -1 \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('turbo b1 split_min=$sm', d['value'], d['config'].get('stage_ms_per_step',{}).get('t3_s'), d.get('decode_step', {}).get('ms_per_step'))" | tee -a $O/split_min_low.log
done
for sm in 512 256 128 512; do
  CBX_DA_SPLIT_MIN=$sm timeout 200 python bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-alt-precisions --no-streaming --no-autotune 2> /dev/null | tail -1 \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mtl b1 split_min=$sm', d['value'], d['config'].get('stage_ms_per_step',{}).get('t3_s'), d.get('decode_step', {}).get('ms_per_step'))" | tee -a $O/split_min_low.log
done
