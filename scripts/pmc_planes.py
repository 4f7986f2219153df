"""PMC target: a few launches of the plane-format attention and GEMM kernels at the bench shape (scripts/pmc_planes.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from chatterbox_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
ROWS, T = 16, 1000
M, Tp = ROWS * T, 1000
qkP, attP, vtP = ops.Planes(M, 1024, dev), ops.Planes(M, 512, dev), ops.Planes(ROWS * 512, Tp, dev, zero=True)
ops.split_planes(torch.randn(M, 1024, device=dev), qkP)
ops.split_planes(torch.randn(ROWS * 512, Tp, device=dev), vtP)
lens = torch.full((ROWS,), T, dtype=torch.int32, device=dev)
for _ in range(6):
    ops.flash_attn_planes(qkP.cols(0, 512), qkP.cols(512, 512), vtP, attP, Z=ROWS, H=8, T=T, vt_sb=512 * vtP.ld, scale=0.125, key_lens=lens)
for name, N, K, tile in [("qk", 1024, 256, 4), ("ff2", 256, 1024, 8)]:
    x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05
    xP, wP, outP, out = ops.split_planes(x), ops.split_planes(w), ops.Planes(M, N, dev), torch.empty(M, N, device=dev)
    ops.lib.cbx_set_planes_tile(tile)
    for _ in range(6):
        if name == "qk":
            ops.linear_planes(xP, wP, outp=outP)
        else:
            ops.linear_planes(xP, wP, out=out, residual=out)
torch.cuda.synchronize()
