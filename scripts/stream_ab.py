"""Same-box A/B of the chunked-synthesis schedules on the bench workload (configs[2] shape: B = 8, 250 tokens, 10 s prompt):
first-audio latency and total wall time of engine.synthesize_stream for a list of (label, kwargs) variants, 3 runs each (median)."""
import json
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
VARIANTS = [
    ("serial rounds 28/78/128/178/228/250 (rounds 3-5)", dict(first_chunk=25, chunk=50, overlap=False)),
    ("overlap, T3 beside round 0, 15/115/250", dict(first_chunk=12, chunk=100, chunk_growth=1.35, first_alone=False)),
    ("overlap, round 0 alone, 15/115/250", dict(first_chunk=12, chunk=100, chunk_growth=1.35)),
    ("overlap, round 0 alone, 15/65/165/250", dict(first_chunk=12, chunk=50, chunk_growth=2.0)),
    ("overlap, round 0 alone, 28/128/250", dict(first_chunk=25, chunk=100, chunk_growth=1.22)),
    ("overlap, round 0 alone, 9/109/250", dict(first_chunk=6, chunk=100, chunk_growth=1.41)),
    ("serial rounds, 15/115/250", dict(first_chunk=12, chunk=100, chunk_growth=1.35, overlap=False)),
]
for label, kw in VARIANTS:
    fl, tl, nr = [], [], 0
    for rep in range(4):
        g = torch.Generator(device=dev).manual_seed(99 + rep)
        us = torch.rand(B, N, generator=g, device=dev)
        torch.cuda.synchronize()
        ts, first, nr = time.perf_counter(), None, 0
        for r in eng.synthesize_stream(texts, t3c, gen, max_new_tokens=N, uniforms=us, ban_eos=True, ban_from=6561, **kw):
            nr += 1
            if first is None:
                first = time.perf_counter() - ts
        torch.cuda.synchronize()
        if rep:  # the first run of a variant captures graphs / warms forms
            fl.append(first), tl.append(time.perf_counter() - ts)
    fl.sort(), tl.sort()
    print(json.dumps(dict(variant=label, rounds=nr, p50_first_audio_ms=round(1e3 * fl[1], 1), p50_total_ms=round(1e3 * tl[1], 1),
                          audio_s_per_wall_s=round(B * (N - 1) / 25.0 / tl[1], 1))), flush=True)
