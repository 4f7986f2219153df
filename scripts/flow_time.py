"""Wall time of the S3Gen flow + HiFT pass at the bench shape (default precision), 5 passes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chatterbox_amd import synth
from chatterbox_amd.hift import HiFTEngine
from chatterbox_amd.s3gen import FlowEngine
dev = torch.device("cuda:0")
sd = synth.s3gen_state_dict(0)
flow, hift = FlowEngine(sd, dev), HiFTEngine(sd, dev)
B, N = int(os.environ.get("CBX_B", "8")), int(os.environ.get("CBX_N", "250"))
toks = torch.stack([synth.speech_tokens(N, seed=b) for b in range(B)]).to(dev)
lens = torch.full((B,), N, dtype=torch.int32, device=dev)
ref = synth.s3gen_ref()
z = synth.randn((B, 2 * (250 + N), 80), seed=9).to(dev)
mels = {}
modes = [False, True, False, True] if flow.use_planes and flow.precision == 16 and not os.environ.get("CBX_FLOW_NO_AB") else [flow.use_planes]
for planes in modes:  # A/B inside one process
    flow.use_planes = planes
    ts = []
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mel = flow.inference(toks, lens, ref, z=z)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        wav, _ = hift.inference(mel)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((1e3 * (t1 - t0), 1e3 * (t2 - t1)))
    mels[planes] = mel
    print(os.environ.get("TAG", ""), f"planes={int(planes)} flow/hift ms:", " ".join(f"{a:.1f}/{b:.1f}" for a, b in ts), flush=True)
if len(mels) == 2:
    d = (mels[True] - mels[False]).abs()
    print(f"mel planes vs fp32-operand path: mean |d| {d.mean():.3e} max {d.max():.3e} (mel max {mels[False].abs().max():.2f})", flush=True)
