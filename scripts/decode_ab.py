"""A/B of T3 decode-step geometries on ONE engine at the bench shape (B = 8, 30 layers, context ~225): T3Engine.apply_variant + measure_decode
(hipGraph replays of the whole token step, HIP events; best of `reps`), logits after one step against the built-in geometry's.  Run on the GPU box:
    python scripts/decode_ab.py [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chatterbox_amd import synth
from chatterbox_amd.autotune import LIB_KNOBS, split_variant
from chatterbox_amd.t3 import T3Engine

dev = torch.device("cuda:0")
L, B = int(os.environ.get("CBX_AB_LAYERS", "30")), int(os.environ.get("CBX_AB_BATCH", "8"))
eng = T3Engine(synth.t3_state_dict(L, 0), dev, n_layers=L)
P7 = dict(da_pipe=7, pre_epi=1)
QK = dict(qkv_ks=4, qkv_ct=3)
VARIANTS = [dict(), dict(P7), dict(QK), dict(QK, head_ct=2), dict(QK, head_ct=3), dict(QK, head_ct=4), dict(QK, head_ct=2, **P7),
            dict(QK, head_ct=2, half_tiles=0, d_ks2=4, d_nw2=16), dict(QK, head_ct=2, half_tiles=0, d_ks2=4, d_nw2=8), dict(QK, head_ct=2, half_tiles=0, d_ks2=4, d_nw2=8, **P7),
            dict(QK, head_ct=2, od_tc=4), dict(QK, head_ct=2, od_tc=4, **P7), dict(QK, head_ct=2, od_tc=4, d_ks2=1, d_nw2=8), dict(QK, head_ct=2, d_ks2=4, d_nw2=8),
            dict(qkv_ks=4, qkv_ct=2, head_ct=2), dict(qkv_ks=4, qkv_ct=4, head_ct=2), dict(qkv_ks=2, qkv_ct=2, head_ct=2), dict(head_ct=2),
            dict(QK, head_ct=2, da_pipe=5, pre_epi=1), dict(QK, head_ct=2, da_pipe=3, pre_epi=1), dict()]
if os.environ.get('CBX_AB_SHORT'):
    VARIANTS = [dict(), dict(o_nw2=16), dict(d_nw2=8), dict(o_nw2=16, d_nw2=8), dict(gu_nw=4), dict(da_pipe=3), dict(da_pipe=1, da_u=8), dict(pre_epi=0), dict()]
if os.environ.get('CBX_AB_B1'):  # Llama at batch 1 (2 rows): geometries without partial images / with narrow tiles
    VARIANTS = [dict(), dict(od_tc=4, d_ks2=1, d_nw2=8), dict(od_tc=4, d_ks2=1, d_nw2=16), dict(d_ks2=1, d_nw2=16), dict(qkv_ks=0), dict(qkv_ks=0, od_tc=4, d_ks2=1, d_nw2=8), dict(qkv_ks=0, qkv_tc=12, od_tc=4, d_ks2=1, d_nw2=8), dict(half_tiles=0, d_ks2=4, d_nw2=8), dict(da_pipe=3), dict()]
rows, ref = [], None
for v in VARIANTS:
    t, k = split_variant(v)
    try:
        eng.apply_variant(dict(T3Engine._TUNE, **t), dict(LIB_KNOBS, **k))
        ms, lg = eng.measure_decode(B=B, ctx=225, steps=48, reps=3)
    except Exception as e:
        rows.append(dict(variant=v, error=f"{type(e).__name__}: {e}"[:200]))
        print(rows[-1], flush=True)
        continue
    if ref is None:
        ref = lg
    d = float((lg - ref).abs().max())
    rows.append(dict(variant=v, ms_per_token=round(ms, 4), identical=bool(torch.equal(lg, ref)), max_abs_diff=d))
    print(f"{ms:.4f} ms/token  identical={rows[-1]['identical']} (max |d logits| {d:.2e})  {v}", flush=True)
if os.environ.get('CBX_AB_SHORT'):
    pass
if len(sys.argv) > 1:
    json.dump(dict(B=B, layers=L, ctx=225, rows=rows, device=torch.cuda.get_device_name(0)), open(sys.argv[1], "w"), indent=1)
