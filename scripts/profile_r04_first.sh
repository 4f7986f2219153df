#!/bin/bash
# FIRST GPU call of round 4.  Round 3 ended with the GPU budget spent; the ABI v9 decode variants (12- / 4-column GEMV tiles, down projection
# without partial images, software-pipelined decode attention) were written afterwards, verified on the SIMT emulator (tests/simt/) and never
# timed.  This call (a) runs their GPU tests, (b) times every variant on the T3 stage at the bench shape inside ONE box, (c) Turbo at batch 1.
#   gpurun --timeout 1500 -- 'bash scripts/profile_r04_first.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/first
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_zz_abi_v9_gpu.py tests/test_ops_gpu.py tests/test_models_gpu.py tests/test_planes_gpu.py -q -m gpu \
    -k "abi_v9 or c_level_decode_step or half_tile or decode_attn or transposed_column_range or stream" > $O/pytest_v9.log 2>&1
tail -3 $O/pytest_v9.log
# T3 stage time, B = 8, 250 tokens, 30 layers: index 3 = the shipped default, 6.. = the round-3 variants (scripts/t3_decode_time.py VARIANTS)
T3_VARIANTS=3,6,7,8,9,10,11,12,13,14,15,16,17,18 timeout 700 python scripts/t3_decode_time.py > $O/t3_decode_variants.log 2>&1
cat $O/t3_decode_variants.log | tail -24
for tune in "" "qkv_tc=12" "od_tc=4,d_ks=1,d_nw=8" "qkv_tc=12,od_tc=4,d_ks=1,d_nw=16"; do
  for pipe in 0 1; do
    CBX_TURBO_TUNE="$tune" CBX_DA_PIPE=$pipe timeout 300 python bench.py --workload turbo --batch 1 --steps 5 --warmup 2 --no-cpu-baseline \
        --no-alt-precisions --no-streaming 2> /dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('turbo b1 tune=[$tune] pipe=$pipe', d['value'], d.get('stage_ms_per_step'), d.get('decode_step', {}).get('ms_per_token'))" | tee -a $O/turbo_b1_variants.log
  done
done
