"""Timing of the plane-format GEMM / attention with parts switched off (-DCBX_DIAG side library, scripts/diag_planes.sh run)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if os.environ.get("CBX_DIAG_CHILD") != "1":  # the attention switch is read from the environment at launch: one child per setting
    env = dict(os.environ, CBX_DIAG_CHILD="1")
    subprocess.run([sys.executable, __file__, "gemm"], env=env)
    for d in (0, 1, 2, 4, 8, 16, 2 | 8, 4 | 16, 2 | 4 | 8 | 16, 1 | 2 | 4 | 8 | 16):
        subprocess.run([sys.executable, __file__, "attn"], env=dict(env, CBX_ATTN_DIAG=str(d)))
    sys.exit(0)

import torch  # noqa: E402

from chatterbox_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
ROWS, T = 16, 1000
M = ROWS * T


def timeit(fn, reps=30):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


if sys.argv[1] == "gemm":
    print("# GEMM: us with diag = 0 | 1 no DMA | 2 no MFMA | 4 no stores | 3 | 5 | 6 | 7", flush=True)
    for name, N, K, taps, epi in [("qk", 1024, 256, 1, "P"), ("qkv_C", 1536, 256, 1, "C"), ("ff1_gelu", 1024, 256, 1, "PG"), ("ff2", 256, 1024, 1, "CR"),
                                   ("conv3_512", 256, 1536, 3, "C")]:
        cin = K // taps
        x, w, b = torch.randn(ROWS, T, cin, device=dev), torch.randn(N, K, device=dev) * 0.05, torch.randn(N, device=dev)
        r, out = torch.randn(ROWS, T, N, device=dev), torch.empty(ROWS, T, N, device=dev)
        xP, wP, outP = ops.split_planes(x.view(M, cin)), ops.split_planes(w), ops.Planes(M, N, dev)
        act = ops.GELU_ERF if "G" in epi else ops.NONE
        for tile in (1, 4, 8, 11, 15, 3):
            ops.lib.cbx_set_planes_tile(tile)
            line = f"{name:10s} t{tile:<2d}"
            for d in (0, 1, 2, 4, 3, 5, 6, 7):
                ops.GEMM_DIAG = d
                us = timeit(lambda: ops.conv1d_planes(xP, wP, B=ROWS, T=T, taps=taps, cin=cin, out=out if "C" in epi else None,
                                                      outp=outP if "P" in epi else None, bias=b, pad_left=taps - 1, act=act, residual=r if "R" in epi else None))
                line += f" {us:6.1f}"
            print(line, flush=True)
    ops.GEMM_DIAG = 0
else:
    Tp = (T + 7) // 8 * 8
    qkP, attP, vtP = ops.Planes(M, 1024, dev), ops.Planes(M, 512, dev), ops.Planes(ROWS * 512, Tp, dev, zero=True)
    ops.split_planes(torch.randn(M, 1024, device=dev), qkP)
    ops.split_planes(torch.randn(ROWS * 512, Tp, device=dev), vtP)
    lens = torch.full((ROWS,), T, dtype=torch.int32, device=dev)
    us = timeit(lambda: ops.flash_attn_planes(qkP.cols(0, 512), qkP.cols(512, 512), vtP, attP, Z=ROWS, H=8, T=T, vt_sb=512 * vtP.ld, scale=0.125, key_lens=lens))
    print(f"attention diag {os.environ.get('CBX_ATTN_DIAG')}: {us:6.1f} us", flush=True)
