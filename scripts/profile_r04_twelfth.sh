#!/bin/bash
# Twelfth GPU call: batch-1 decode geometries now that rows past M no longer cost bytes (Turbo / Nano via CBX_TURBO_TUNE, Llama B = 1 via decode_ab).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/twelfth
mkdir -p $O
cd $R
for tune in "" "d_ks=1,d_nw=16" "od_tc=4,d_ks=1,d_nw=8" "od_tc=4,d_ks=1,d_nw=16" "qkv_tc=12,od_tc=4,d_ks=1,d_nw=16" "d_ks=4,d_nw=8" "half_tiles=0" ""; do
  CBX_TURBO_TUNE="$tune" timeout 200 python bench.py --workload turbo --batch 1 --steps 4 --warmup 2 --no-cpu-baseline --no-alt-precisions --no-streaming --no-autotune 2> /dev/null | tail -1 \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('turbo b1 tune=[$tune]', d['value'], d['config'].get('stage_ms_per_step',{}).get('t3_s'), d.get('decode_step', {}).get('ms_per_step'))" | tee -a $O/turbo_b1_tune.log
done
CBX_AB_BATCH=1 CBX_AB_B1=1 timeout 200 python scripts/decode_ab.py $O/decode_ab_b1.json 2>&1 | grep -v amdgpu.ids | tee $O/decode_ab_b1.log
