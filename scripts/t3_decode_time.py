"""T3 stage time at the bench shape (B = 8, 64 text tokens, 250 speech tokens) for the decode variants (run on the GPU box)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chatterbox_amd import ops, synth
from chatterbox_amd.autotune import LIB_KNOBS, split_variant
from chatterbox_amd.t3 import T3Engine

dev = torch.device("cuda:0")
sd = synth.t3_state_dict(30, 0)
B, N = 8, 250
texts = [synth.text_tokens(64, seed=b) for b in range(B)]
u = synth.rand((B, N), seed=1)
res = {}
VARIANTS = [("v1", {}), ("v2", dict(half_tiles=0)), ("v2", dict(half_tiles=1, d_ks2=4)), ("v2", dict(half_tiles=1, d_ks2=2)),
            ("v2", dict(half_tiles=1, d_ks2=2, d_nw2=16)), ("v2", dict(half_tiles=0, d_ks2=2)),
            # round-3 tile variants (written without GPU access, verified on the SIMT emulator; first thing to time in round 4):
            # q/k/v on 256 workgroups; o / down on 256 workgroups with the residual added by the down projection (no partial images)
            ("v2", dict(qkv_tc=12)), ("v2", dict(od_tc=4, d_ks2=1, d_nw2=8)), ("v2", dict(od_tc=4, d_ks2=1, d_nw2=16)),
            ("v2", dict(qkv_tc=12, od_tc=4, d_ks2=1, d_nw2=16)), ("v2", dict(od_tc=4, d_ks2=1, d_nw2=16, o_nw2=16)),
            # software-pipelined decode attention (cbx_set_decode_attn_pipeline): alone, with 8 rows per step, and with the tile variants
            ("v2", dict(od_tc=4, d_ks2=1, d_nw2=8, deep=1)), ("v2", dict(da_pipe=1, qkv_tc=12, od_tc=4, d_ks2=1, d_nw2=8, deep=1)),
            ("v2", dict(da_pipe=1)), ("v2", dict(da_pipe=1, da_u=8)), ("v2", dict(prefill_prec=6)), ("v2", dict(da_pipe=2)), ("v2", dict(da_pipe=3)), ("v2", dict(da_pipe=1, qkv_tc=12, od_tc=4, d_ks2=1, d_nw2=16)),
            # last session of round 3 (no GPU): speculative first K / V step of the decode attention (da_pipe bit 2), epilogue operands of every
            # GEMV requested with its first weight batch (pre_epi), and both with the pipelined stream / on the reordered geometry
            ("v2", dict(da_pipe=4)), ("v2", dict(da_pipe=5)), ("v2", dict(da_pipe=7)), ("v2", dict(pre_epi=1)), ("v2", dict(da_pipe=5, pre_epi=1)),
            ("v2", dict(da_pipe=5, pre_epi=1, qkv_tc=12, od_tc=4, d_ks2=1, d_nw2=8))]
if os.environ.get("T3_VARIANTS"):  # e.g. T3_VARIANTS=1 profiles only the default v2 configuration
    VARIANTS = [VARIANTS[int(i)] for i in os.environ["T3_VARIANTS"].split(",")]
for mode, tune in VARIANTS:
    os.environ["CBX_T3_DECODE"] = mode
    eng = T3Engine(sd, dev)
    tune = dict(tune)
    da_u, da_pipe, pre_epi = tune.get("da_u", 4), tune.get("da_pipe", 0), tune.get("pre_epi", 0)
    t, k = split_variant(tune)  # per-engine geometry (ABI v10): tile keys + launch knobs, nothing process-wide
    eng.apply_variant(dict(eng.tune, **t), dict(LIB_KNOBS, **k))
    tune = t
    kw = dict(max_new_tokens=N, uniforms=u, ban_eos=True, ban_from=6561)
    toks = eng.generate(synth.t3_cond(), texts, **kw)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        toks = eng.generate(synth.t3_cond(), texts, **kw)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    key = f"{mode} {tune} da_u={da_u} da_pipe={da_pipe} pre_epi={pre_epi}"
    res[key] = [t.tolist() for t in toks]
    print(f"{key:50s} T3 stage {min(ts) * 1e3:7.1f} ms  ({min(ts) / (N - 1) * 1e3:.3f} ms/token incl. prefill)", flush=True)
    del eng
    torch.cuda.empty_cache()
ks = list(res)
for k in ks[1:]:
    same = res[k] == res[ks[0]]
    print(f"tokens {k} == {ks[0]}: {same}")
