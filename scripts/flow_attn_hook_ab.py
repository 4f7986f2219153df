"""Same-process A / B of the plane-attention choice inside the serial flow at a given number of speech tokens (CBX_NTOK; 10 s prompt, B = 8): automatic (hook 0) against
version 4 forced through the test hook, interleaved."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from chatterbox_amd import ops, synth  # noqa: E402
from chatterbox_amd.s3gen import FlowEngine  # noqa: E402

dev = torch.device("cuda:0")
flow = FlowEngine(synth.s3gen_state_dict(0), dev)
B, N = 8, int(os.environ.get("CBX_NTOK", "15"))
toks = torch.stack([synth.speech_tokens(N, seed=b) for b in range(B)]).to(dev)
lens = torch.full((B,), N, dtype=torch.int32, device=dev)
ref = synth.s3gen_ref()
res = {0: [], 4: []}
for i in range(12):
    v = (0, 4)[i % 2]
    ops.lib.cbx_set_attn_planes_version(v)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    flow.inference(toks, lens, ref)
    torch.cuda.synchronize()
    if i >= 2:
        res[v].append(1e3 * (time.perf_counter() - t0))
ops.lib.cbx_set_attn_planes_version(0)
print(f"{N} tokens: automatic {sorted(res[0])[len(res[0]) // 2]:.2f} ms | version 4 forced {sorted(res[4])[len(res[4]) // 2]:.2f} ms", flush=True)
