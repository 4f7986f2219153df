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
    -k "abi_v9 or c_level_decode_step or half_tile or decode_attn or transposed_column_range or stream or flow_batched_ragged" > $O/pytest_v9.log 2>&1
tail -3 $O/pytest_v9.log
# T3 stage time, B = 8, 250 tokens, 30 layers: index 3 = the shipped default, 6.. = the round-3 variants (scripts/t3_decode_time.py VARIANTS)
T3_VARIANTS=3,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33 timeout 1200 python scripts/t3_decode_time.py > $O/t3_decode_variants.log 2>&1
cat $O/t3_decode_variants.log | tail -24
for tune in "" "qkv_tc=12" "od_tc=4,d_ks=1,d_nw=8" "qkv_tc=12,od_tc=4,d_ks=1,d_nw=16" "chain=1,od_tc=4,d_ks=1,d_nw=8" "chain=1,qkv_tc=12,od_tc=4,d_ks=1,d_nw=8"; do
  for pipe in 0 1; do
    CBX_TURBO_TUNE="$tune" CBX_DA_PIPE=$pipe timeout 300 python bench.py --workload turbo --batch 1 --steps 5 --warmup 2 --no-cpu-baseline \
        --no-alt-precisions --no-streaming 2> /dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('turbo b1 tune=[$tune] pipe=$pipe', d['value'], d.get('stage_ms_per_step'), d.get('decode_step', {}).get('ms_per_token'))" | tee -a $O/turbo_b1_variants.log
  done
done
# GPT-2 decode GEMVs all carry a bias / LayerNorm-fold constants in their epilogue: the epilogue prefetch and the speculative attention step on Turbo, batch 1
for env in "CBX_GEMV_PRE_EPI=1" "CBX_DA_PIPE=4" "CBX_GEMV_PRE_EPI=1 CBX_DA_PIPE=5"; do
  env $env timeout 300 python bench.py --workload turbo --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precisions --no-streaming 2> /dev/null \
      | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('turbo b1 [$env]', d['value'], d.get('stage_ms_per_step'), d.get('decode_step', {}).get('ms_per_token'))" | tee -a $O/turbo_b1_variants.log
done
# the flash form of the encoder's rel-pos attention (cbx_flash_relpos_f32, emulator-verified): encoder time at the bench shape and at 60 s
for fl in 0 1; do
  CBX_ENC_FLASH=$fl timeout 300 python - <<'PY' 2>&1 | tail -2 | tee -a $O/enc_flash_ab.log
import os, time, torch
from chatterbox_amd import synth
from chatterbox_amd.s3gen import FlowEngine
eng = FlowEngine(synth.s3gen_state_dict(0), "cuda")
for B, N in ((8, 500), (1, 1750)):
    tok = torch.randint(0, 6561, (B, N), device="cuda"); lens = torch.full((B,), N, dtype=torch.int32, device="cuda")
    eng.encode(tok, lens); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): eng.encode(tok, lens)
    torch.cuda.synchronize(); print(f"CBX_ENC_FLASH={os.environ['CBX_ENC_FLASH']} encoder B={B} N={N}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms")
PY
done
