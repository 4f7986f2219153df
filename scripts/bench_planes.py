"""Plane-format GEMM / attention / LayerNorm on the CFM shapes of the bench workload (rows = 16, T = 1000), every tile of the menu,
next to the fp32-operand f16x3 kernels they replace (run on the GPU box).

    CBX_PL_TILES=0,2,3,...   tiles to time (default: the whole menu)      CBX_REPS=50
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from chatterbox_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
ROWS, T = int(os.environ.get("CBX_ROWS", "16")), int(os.environ.get("CBX_T", "1000"))
M = ROWS * T
REPS = int(os.environ.get("CBX_REPS", "50"))
tiles = [int(t) for t in os.environ.get("CBX_PL_TILES", ",".join(str(i) for i in range(0, 18))).split(",")]


def timeit(fn, reps=REPS):
    """us per launch inside a captured hipGraph of `reps` back-to-back launches (no host launch cost: a Python-issued launch costs
    10-20 us, more than some of these kernels; includes the ~1.5 us dependent-kernel boundary)."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
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
    return e0.elapsed_time(e1) / (3 * reps) * 1e3  # us


# ramp the clocks
wa, wb, wc = torch.randn(8192, 1024, device=dev), torch.randn(8192, 1024, device=dev), torch.empty(8192, 8192, device=dev)
t0 = time.time()
while time.time() - t0 < 1.0:
    ops.linear(wa, wb, wc)
    torch.cuda.synchronize()
del wa, wb, wc

#        name        N     K     taps  epilogue
shapes = [("qk", 1024, 256, 1, "P"), ("qkv_old", 1536, 256, 1, "C"), ("attn_out", 256, 512, 1, "CR"), ("ff1_gelu", 1024, 256, 1, "PG"),
          ("ff2", 256, 1024, 1, "CR"), ("ff2_last", 256, 1024, 1, "PR"), ("conv3_256", 256, 768, 3, "C"), ("conv3_320", 256, 960, 3, "C"),
          ("conv3_512", 256, 1536, 3, "C"), ("res1x1", 256, 256, 1, "CR"), ("final_proj", 80, 256, 1, "C")]
print(f"# M = {M} (rows {ROWS} x T {T}); us per launch | TF fp32-equivalent (x3 = fp16 MFMA rate)", flush=True)
best = {}
for name, N, K, taps, epi in shapes:
    cin = K // taps
    x = torch.randn(ROWS, T, cin, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    r = torch.randn(ROWS, T, N, device=dev)
    out = torch.empty(ROWS, T, N, device=dev)
    xP, wP, outP = ops.split_planes(x.view(M, cin)), ops.split_planes(w), ops.Planes(M, N, dev)
    act = ops.GELU_ERF if "G" in epi else ops.NONE
    res = r if "R" in epi else None
    fl = 2.0 * M * N * K

    def old():
        ops.conv1d(x, w, out, taps=taps, cin=cin, bias=b, pad_left=taps - 1, act=act, residual=res)

    def new():
        ops.conv1d_planes(xP, wP, B=ROWS, T=T, taps=taps, cin=cin, out=out if "C" in epi else None, outp=outP if "P" in epi else None, bias=b,
                          pad_left=taps - 1, act=act, residual=res)

    with ops.gemm_precision(16):
        us = timeit(old)
    line = f"{name:11s} N={N:5d} K={K:5d} | split {us:6.1f} us {fl / us / 1e6:6.1f} TF |"
    ops.lib.cbx_set_planes_tile(0)
    ops.lib.cbx_set_planes_persist(0)
    line += f" one-tile-per-WG t0: {timeit(new):5.1f} | persistent"
    ops.lib.cbx_set_planes_persist(1)
    for t in tiles:
        ops.lib.cbx_set_planes_tile(t)
        try:
            us = timeit(new)
        except RuntimeError as e:  # a tile that does not serve the shape
            line += f" t{t}: n/a"
            continue
        line += f" t{t}: {us:5.1f}"
        if t and (name not in best or us < best[name][1]):
            best[name] = (t, us, fl / us / 1e6)
    ops.lib.cbx_set_planes_tile(0)
    print(line, flush=True)
print("# best tile per shape: " + ", ".join(f"{k}: t{v[0]} {v[1]:.1f} us {v[2]:.0f} TF" for k, v in best.items()), flush=True)

# swapped V^T product
Tp = (T + 7) // 8 * 8
h = torch.randn(M, 256, device=dev)
hP, wvP, vtP = ops.split_planes(h), ops.split_planes(torch.randn(512, 256, device=dev) * 0.05), ops.Planes(ROWS * 512, Tp, dev, zero=True)
line = "vT swapped  (512 x T x 256 per row) |"
for t in tiles:
    ops.lib.cbx_set_planes_tile(t)
    us = timeit(lambda: ops.gemm_planes(wvP, hP, M=512, N=T, K=256, nz1=ROWS, w_s1=T * hP.ld, P=vtP, p_s1=512 * vtP.ld))
    line += f" t{t}: {us:5.1f}"
ops.lib.cbx_set_planes_tile(0)
print(line, flush=True)

# attention
qkv = torch.randn(M, 1536, device=dev)
att = torch.empty(M, 512, device=dev)
q5 = qkv.view(ROWS, T, 3, 8, 64)
lens = torch.full((ROWS,), T, dtype=torch.int32, device=dev)
with ops.gemm_precision(16):
    us_old = timeit(lambda: ops.flash_attn(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2], att.view(ROWS, T, 8, 64), 0.125, key_lens=lens))
qkP, attP = ops.Planes(M, 1024, dev), ops.Planes(M, 512, dev)
ops.split_planes(qkv[:, :1024], qkP)
ops.split_planes(torch.randn(ROWS * 512, Tp, device=dev), vtP)
fl = 4.0 * ROWS * 8 * T * T * 64
line = f"flash attention: split {us_old:6.1f} us {fl / us_old / 1e6:6.1f} TF |"
for ver in (1, 2, 3, 4):
    ops.lib.cbx_set_attn_planes_version(ver)
    us_new = timeit(lambda: ops.flash_attn_planes(qkP.cols(0, 512), qkP.cols(512, 512), vtP, attP, Z=ROWS, H=8, T=T, vt_sb=512 * vtP.ld, scale=0.125,
                                                  key_lens=lens))
    line += f" planes v{ver} {us_new:6.1f} us {fl / us_new / 1e6:6.1f} TF fp32-equivalent |"
ops.lib.cbx_set_attn_planes_version(0)
print(line, flush=True)


# LayerNorm from the GEMM epilogue (cbx_gemm_pl_t.ln_w, round 5) against the two launches it replaces
for name, K in (("attn_out + norm3", 512), ("ff2 + next norm1", 1024)):
    aP, wP2 = ops.split_planes(torch.randn(M, K, device=dev)), ops.split_planes(torch.randn(256, K, device=dev) * 0.05)
    bb, xr, lw_, lb_ = torch.randn(256, device=dev), torch.randn(M, 256, device=dev), torch.ones(256, device=dev), torch.zeros(256, device=dev)
    hP_ = ops.Planes(M, 256, dev)

    def two():
        ops.linear_planes(aP, wP2, out=xr, bias=bb, residual=xr)
        ops.layernorm_planes(xr, lw_, lb_, hP_, 1e-5)

    def one():
        ops.linear_planes(aP, wP2, out=xr, bias=bb, residual=xr, ln=(lw_, lb_), lnp=hP_)

    print(f"{name:18s} K={K:5d} | Linear + layernorm_planes (2 launches) {timeit(two):6.1f} us | LayerNorm in the epilogue (1 launch) {timeit(one):6.1f} us", flush=True)

# LayerNorm
x = torch.randn(M, 256, device=dev)
w1, b1, y, st = torch.ones(256, device=dev), torch.zeros(256, device=dev), torch.empty(M, 256, device=dev), torch.empty(M, 2, device=dev)
yP = ops.Planes(M, 256, dev)
x320, x320P = torch.randn(M, 320, device=dev), ops.Planes(M, 320, dev)
print(f"layernorm fp32 {timeit(lambda: ops.layernorm(x, w1, b1, y)):5.1f} us | planes {timeit(lambda: ops.layernorm_planes(x, w1, b1, yP)):5.1f} us | "
      f"row_stats {timeit(lambda: ops.row_stats(x, st)):5.1f} us | split_planes(80 cols of 320) "
      f"{timeit(lambda: ops.split_planes(x320[:, :80], x320P.cols(0, 80))):5.1f} us", flush=True)
