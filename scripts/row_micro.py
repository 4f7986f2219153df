"""Per-launch time of the batch-1 row kernels (cbx_gemv_row_f32) inside a dependent hipGraph chain over DISTINCT weights (24 layers' worth: what a Turbo
decode step pays per launch), per shape and rows-per-wave: `python scripts/row_micro.py [turbo|nano]`."""
import sys

import torch

sys.path.insert(0, ".")
from chatterbox_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
D = 768 if (len(sys.argv) > 1 and sys.argv[1] == "nano") else 1024
L = 24
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=dev) * 0.02
shapes = {"c_attn (LN)": (3 * D, D, "ln"), "c_proj (merge)": (D, D, "attn"), "c_fc (LN, gelu)": (4 * D, D, "ln"), "mlp c_proj": (D, 4 * D, "plain"),
          "head (LN)": (6563, D, "ln")}
x, lnw, lnb = rn(4 * D), rn(4 * D) + 1, rn(4 * D)
H = D // 64
parts = torch.zeros(H, 16, ops.ATTN_PART_REC, device=dev)
parts[:, :, 1] = 1.0
parts[:, :, 4:] = rn(H, 16, 64)
for name, (N, K, pro) in shapes.items():
    Ws = [rn(N, K) for _ in range(L)]
    bias, out = rn(N), torch.zeros(N, device=dev)
    line = f"{name:18s} N={N:5d} K={K:5d} {4 * N * K / 1e6:6.1f} MB:"
    for R in ((0, 1, 2) if K > 1024 else (0, 1, 2, 3, 4, 8)):
        kw = dict(bias=bias, rows_per_wave=R)
        if pro == "ln":
            kw.update(ln=(lnw[:K], lnb[:K]))
        if pro == "attn":
            kw.update(parts=parts)

        def chain():
            for W in Ws:
                ops.gemv_row(None if pro == "attn" else x[:K], W, out, res=out if pro != "ln" else None, **kw)
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            chain()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            chain()
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / (10 * L)
        line += f"  R={R}: {us:5.2f} us ({4 * N * K / us / 1e6:4.2f} TB/s)"
    print(line, flush=True)
