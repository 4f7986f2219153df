"""Two never-timed opt-ins at the bench shape (one box): (a) the T3 prefill on the bf16x6 split kernels (tune prefill_prec = 6) against the exact fp32 MFMA,
(b) the conformer encoder with the flash rel-pos attention (CBX_ENC_FLASH=1) against the materialised scores."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chatterbox_amd import synth
from chatterbox_amd.s3gen import FlowEngine
from chatterbox_amd.t3 import T3Engine

dev = torch.device("cuda:0")
B, N = 8, 32
eng = T3Engine(synth.t3_state_dict(30, 0), dev, n_layers=30)
texts = [synth.text_tokens(64, seed=b) for b in range(B)]
u = synth.rand((B, N), seed=1).to(dev)
kw = dict(max_new_tokens=N, uniforms=u, ban_eos=True, ban_from=6561)
toks = {}
for prec in (0, 6, 0, 6):
    eng.tune["prefill_prec"] = prec
    eng.generate(synth.t3_cond(), texts, **kw)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        toks[prec] = eng.generate(synth.t3_cond(), texts, **kw)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"T3 prefill_prec={prec}: generate({N} tokens) {min(ts) * 1e3:.1f} ms  (prefill = this minus {N - 1} decode steps)", flush=True)
print("tokens equal:", [a.tolist() for a in toks[0]] == [a.tolist() for a in toks[6]], flush=True)
del eng
torch.cuda.empty_cache()
fe = FlowEngine(synth.s3gen_state_dict(0), dev)
for fl in ("0", "1", "0", "1"):
    fe.ENC_FLASH = fl
    for Bz, Nt in ((8, 500), (1, 1750)):
        tok = torch.randint(0, 6561, (Bz, Nt), device=dev)
        lens = torch.full((Bz,), Nt, dtype=torch.int32, device=dev)
        fe.encode(tok, lens)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fe.encode(tok, lens)
        torch.cuda.synchronize()
        print(f"encoder ENC_FLASH={fl} B={Bz} N={Nt}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms", flush=True)
