"""Decode-step time for a few launch geometries (run on the GPU box)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chatterbox_amd import synth
from chatterbox_amd.t3 import T3Engine
dev = torch.device("cuda:0")
eng = T3Engine(synth.t3_state_dict(30, 0), dev)
B, N = 8, 40
texts = [synth.text_tokens(64, seed=b) for b in range(B)]
u = synth.rand((B, N), seed=1)
variants = [dict(qkv_nw=8, o_ks=4, gu_nw=8, d_ks=8, head_nw=4), dict(qkv_nw=4, o_ks=4, gu_nw=8, d_ks=8, head_nw=4),
            dict(qkv_nw=8, o_ks=2, gu_nw=8, d_ks=4, head_nw=4), dict(qkv_nw=8, o_ks=8, gu_nw=8, d_ks=8, head_nw=8),
            dict(qkv_nw=8, o_ks=4, gu_nw=4, d_ks=8, head_nw=8), dict(qkv_nw=8, o_ks=1, gu_nw=8, d_ks=2, head_nw=4)]
for v in variants:
    eng.tune = v
    eng._state.clear()
    eng.generate(synth.t3_cond(), texts, max_new_tokens=N, uniforms=u, ban_eos=True, ban_from=6561)
    st = list(eng._state.values())[0]
    g = st["graph"]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): g.replay()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(v, f"-> {(t1-t0)/100*1e3:.3f} ms/step", flush=True)
