"""LayerNorm (C = 256) timing on the CFM shapes: plain and with the fused Mish (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chatterbox_amd import ops
dev = torch.device("cuda:0")
rows = 16000
x = torch.randn(rows, 256, device=dev); y = torch.empty_like(x)
w = torch.randn(256, device=dev); b = torch.randn(256, device=dev)
xs = torch.randn(rows, 320, device=dev)[:, :256]
for name, xin, act in (("plain", x, ops.NONE), ("mish", x, ops.MISH), ("plain ld320", xs, ops.NONE), ("gelu", x, ops.GELU_ERF)):
    for _ in range(5): ops.layernorm(xin, w, b, y, act=act)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): ops.layernorm(xin, w, b, y, act=act)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    print(f"CBX_LN_NARROW={os.environ.get('CBX_LN_NARROW','1')} {name:12s} {us:7.2f} us  {2*rows*256*4/us/1e6:6.2f} TB/s", flush=True)
