"""Per-shape timing of the implicit-GEMM kernel on the CFM / prefill shapes (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chatterbox_amd import ops
dev = torch.device("cuda:0")
ops.GEMM_DIAG = int(os.environ.get("CBX_DIAG", "0"))
M = 16000
shapes = [("qkv", M, 1536, 256, 1), ("attn_out", M, 256, 512, 1), ("ff1+gelu", M, 1024, 256, 1), ("ff2", M, 256, 1024, 1),
          ("conv3_256", M, 256, 768, 3), ("conv3_320", M, 256, 960, 3), ("conv3_512", M, 256, 1536, 3), ("res1x1", M, 256, 256, 1),
          ("final_proj", M, 80, 256, 1), ("enc_ff1", 8000, 2048, 512, 1), ("enc_ff2", 8000, 512, 2048, 1),
          ("t3_prefill_qkv", 1648, 3072, 1024, 1), ("t3_prefill_o", 1648, 1024, 1024, 1), ("t3_prefill_down", 1648, 1024, 4096, 1),
          ("big", 8192, 8192, 1024, 1)]
# ramp the clocks: ~1 s of dense work before any timing
wa, wb, wc = torch.randn(8192, 1024, device=dev), torch.randn(8192, 1024, device=dev), torch.empty(8192, 8192, device=dev)
t0 = time.time()
while time.time() - t0 < 1.0:
    ops.linear(wa, wb, wc); torch.cuda.synchronize()
sel = os.environ.get('CBX_GEMM_SHAPES')
for name, m, n, k, taps in shapes:
    if sel and name not in sel.split(','): continue
    cin = k // taps
    PAD = int(os.environ.get("CBX_PAD", "0"))
    x = torch.randn(16, m // 16 + 4, cin + PAD, device=dev)[..., :cin]
    w = torch.randn(n, k, device=dev) * 0.05
    b = torch.randn(n, device=dev)
    out = torch.empty(16, m // 16, n + PAD, device=dev)[..., :n]
    r = torch.randn(16, m // 16, n + PAD, device=dev)[..., :n]
    ACT = getattr(ops, os.environ.get("CBX_ACT", "NONE"))
    def run():
        if taps == 1:
            ops.conv1d(x[:, : m // 16], w, out, taps=1, cin=cin, bias=b, residual=None if ACT else r, act=ACT)
        else:
            ops.conv1d(x[:, : m // 16], w, out, taps=taps, cin=cin, bias=b, pad_left=taps - 1)
    line = f"{name:16s} M={m:6d} N={n:5d} K={k:5d} "
    for prec, tile in [(int(v.split(":")[0]), int(v.split(":")[1]) if ":" in v else 0) for v in os.environ.get("CBX_PRECS", "1").split(",")]:
        ops.lib.cbx_set_split_tile(tile)
        with ops.gemm_precision(prec):
            for _ in range(3): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = int(os.environ.get('CBX_REPS', '100'))
            e0.record()
            for _ in range(reps): run()
            e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        fl = 2.0 * m * n * k
        line += f" | p{prec}/t{tile}: {us:7.1f} us {fl/us/1e6:6.1f} TF"
    print(line, flush=True)
