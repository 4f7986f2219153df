#!/bin/bash
# Thirteenth GPU call: where does the split-context decode attention start to pay now that the one-workgroup walk is pipelined?  Turbo batch 1, 250 and 1000 tokens.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/thirteenth
mkdir -p $O
cd $R
for sm in 512 768 1024 1536 100000 512; do
  for tk in 250 1000; do
    CBX_DA_SPLIT_MIN=$sm timeout 200 python bench.py --workload turbo --batch 1 --tokens $tk --steps 3 --warmup 1 --no-cpu-baseline --no-alt-precisions --no-streaming --no-autotune 2> /dev/null | tail -1 \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('turbo b1 split_min=$sm tokens=$tk', d['value'], d['config'].get('stage_ms_per_step',{}).get('t3_s'), d.get('decode_step', {}).get('ms_per_step'))" | tee -a $O/turbo_split_min.log
  done
done
