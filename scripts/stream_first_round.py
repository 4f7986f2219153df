"""What the first round of engine.synthesize_stream (bench workload: B = 8, 12 + 3 tokens, 10 s prompt) is made of: T3 alone (prefill + 15 steps), flow + vocoder alone over
those tokens, and the generator's first yield -- medians of 5 after 2 warm runs."""
import sys
import time

import torch

sys.path.insert(0, ".")
from chatterbox_amd import synth  # noqa: E402
from chatterbox_amd.engine import ChatterboxEngine  # noqa: E402

dev = torch.device("cuda", 0)
B, N, L = 8, 250, 30
eng = ChatterboxEngine(synth.t3_state_dict(L, 0), synth.s3gen_state_dict(0), dev, n_t3_layers=L)
t3c, gen = synth.t3_cond(prompt_len=150), synth.s3gen_ref()
texts = [synth.text_tokens(64, seed=b) for b in range(B)]
kw = dict(first_chunk=12, chunk=100, chunk_growth=1.35)


def med(f, n=5, warm=2):
    ts = []
    for i in range(warm + n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        if i >= warm:
            ts.append(1e3 * (time.perf_counter() - t0))
    return sorted(ts)[len(ts) // 2]


us = torch.rand(B, N, device=dev)
t3kw = dict(max_new_tokens=N, uniforms=us, ban_eos=True, ban_from=6561, temperature=0.8, top_p=1.0, min_p=0.05, repetition_penalty=1.2, cfg_weight=0.5)
hold = {}


def t3_15():
    hold["h"] = eng.t3.generate(t3c, texts, run_steps=15, async_mode=True, **t3kw)


print(f"T3: prefill + 15 steps               {med(t3_15):7.1f} ms", flush=True)
toks, done = eng.t3.peek(hold["h"])
tok = torch.stack([t[:15] for t in toks]).to(dev)
ns = torch.full((B,), 15, dtype=torch.int32, device=dev)
P = gen["prompt_token"].shape[-1]
z = torch.randn(B, 2 * (P + N), 80, device=dev)


def flow15():
    hold["mel"] = eng.flow.inference(tok, ns, gen, z=z[:, : 2 * (P + 15)], n_steps=10, hold_back=[6] * B)


print(f"flow (encoder + 10 CFM steps), 15 tok {med(flow15):7.1f} ms   (mel frames: {hold['mel'].shape[1]})", flush=True)
print(f"vocoder over them                     {med(lambda: eng.hift.inference(hold['mel'])):7.1f} ms", flush=True)


def first_yield():
    g = eng.synthesize_stream(texts, t3c, gen, max_new_tokens=N, uniforms=us, ban_eos=True, ban_from=6561, **kw)
    next(g)
    hold["t_first"] = time.perf_counter()
    for _ in g:
        pass


ts = []
for i in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    first_yield()
    torch.cuda.synchronize()
    if i >= 2:
        ts.append(1e3 * (hold["t_first"] - t0))
print(f"synthesize_stream: first yield         {sorted(ts)[len(ts) // 2]:7.1f} ms", flush=True)
