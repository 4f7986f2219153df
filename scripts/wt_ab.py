"""Decode-store policy A/B (cbx_common.h CBX_WT; side libraries from scripts/wt_build.sh): ms per token step of the default T3 decode geometry at the bench shape
(B = 8, 30 layers, context 225; hipGraph replays, HIP events, best of 3) for the library CBX_LIB_PATH names, and the identity of its logits with the first run's.
    for lib in "" wt1 wt2 wt3 "" wt1; do CBX_LIB_PATH=${lib:+$PWD/chatterbox_amd/build/libcbx_hip_$lib.so} python scripts/wt_ab.py; done"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chatterbox_amd import synth
from chatterbox_amd.t3 import T3Engine

dev = torch.device("cuda:0")
name = os.path.basename(os.environ.get("CBX_LIB_PATH") or "libcbx_hip.so")
eng = T3Engine(synth.t3_state_dict(30, 0), dev)
out = []
for ctx in (225, 100, 350):
    ms, lg = eng.measure_decode(B=8, ctx=ctx, steps=48, reps=3)
    ref_path = f"/tmp/wt_ref_{ctx}.pt"
    if not os.path.exists(ref_path):
        torch.save(lg.cpu(), ref_path)
    same = bool(torch.equal(lg.cpu(), torch.load(ref_path)))
    out.append(f"ctx {ctx}: {ms:.4f} ms/token (logits identical to the first run: {same})")
print(f"{name:22s} " + " | ".join(out), flush=True)
