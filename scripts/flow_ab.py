"""ms per S3Gen flow pass (encoder + 10 CFM steps) at the bench shape (B = 8, 250 tokens, 10 s prompt), serial, eager: median of CBX_REPS passes after two warm ones.
For same-box A / B runs of side libraries (CBX_LIB_PATH)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from chatterbox_amd import synth  # noqa: E402
from chatterbox_amd.s3gen import FlowEngine  # noqa: E402

dev = torch.device("cuda:0")
flow = FlowEngine(synth.s3gen_state_dict(0), dev)
B, N = int(os.environ.get("CBX_B", "8")), 250
toks = torch.stack([synth.speech_tokens(N, seed=b) for b in range(B)]).to(dev)
lens = torch.full((B,), N, dtype=torch.int32, device=dev)
ref = synth.s3gen_ref()
z = synth.randn((B, 2 * (250 + N), 80), seed=9).to(dev)  # fixed noise: the mel's digest identifies a bit-identical library
ts = []
for i in range(2 + int(os.environ.get("CBX_REPS", "9"))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mel = flow.inference(toks, lens, ref, z=z)
    torch.cuda.synchronize()
    if i >= 2:
        ts.append(1e3 * (time.perf_counter() - t0))
ts.sort()
import hashlib  # noqa: E402
dig = hashlib.sha256(mel.cpu().numpy().tobytes()).hexdigest()[:12]
print(f"{os.environ.get('CBX_LABEL', '')} flow ms: median {ts[len(ts) // 2]:.2f} min {ts[0]:.2f} max {ts[-1]:.2f}  mel sha256 {dig}", flush=True)
