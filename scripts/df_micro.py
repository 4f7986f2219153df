"""The deferred-epilogue forms of cbx_gemm_planes (tiles 41 / 42) next to their plain twins (32 / 35) and the automatic choice on the Linears of a CFM
transformer block as the flow issues them (run on the GPU box).  us per launch inside a hipGraph of back-to-back launches.

    CBX_ROWS=16 CBX_T=1000 CBX_REPS=40 CBX_PL_TILES=0,32,41,35,42
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from chatterbox_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
ROWS, T = int(os.environ.get("CBX_ROWS", "16")), int(os.environ.get("CBX_T", "1000"))
M = ROWS * T
REPS = int(os.environ.get("CBX_REPS", "40"))
tiles = [int(t) for t in os.environ.get("CBX_PL_TILES", "0,32,41,35,42").split(",")]


def capture(fn, reps=REPS):
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
    return g, s


def time_graph(gs, reps=REPS):
    g, s = gs
    with torch.cuda.stream(s):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


def interleaved(fns, rounds=int(os.environ.get("CBX_ROUNDS", "7"))):
    """median us per launch of each variant over `rounds` INTERLEAVED rounds (the first measurements of a process run at another clock / cache state than the
    later ones: a variant's position in a sequential sweep decides more than its code -- profiles/r06_n_*.log against r06_o_*.log)"""
    gs = [capture(f) for f in fns]
    ts = [[] for _ in fns]
    for _ in range(rounds):
        for i, g in enumerate(gs):
            ts[i].append(time_graph(g))
    return [sorted(t)[len(t) // 2] for t in ts]


wa, wb, wc = torch.randn(8192, 1024, device=dev), torch.randn(8192, 1024, device=dev), torch.empty(8192, 8192, device=dev)
for _ in range(30):
    ops.linear(wa, wb, wc)
torch.cuda.synchronize()
del wa, wb, wc

Tp = (T + 7) // 8 * 8
hP = ops.split_planes(torch.randn(M, 256, device=dev))
wqkv, w1 = ops.split_planes(torch.randn(1536, 256, device=dev) * 0.05), ops.split_planes(torch.randn(1024, 256, device=dev) * 0.05)
wo, w2 = ops.split_planes(torch.randn(256, 512, device=dev) * 0.05), ops.split_planes(torch.randn(256, 1024, device=dev) * 0.05)
qkP, vtP, ffP, attP = ops.Planes(M, 1024, dev), ops.Planes(ROWS * 512, Tp, dev, zero=True), ops.Planes(M, 1024, dev), ops.Planes(M, 512, dev)
ops.split_planes(torch.randn(M, 1024, device=dev), ffP)
ops.split_planes(torch.randn(M, 512, device=dev), attP)
b1, bo = torch.randn(1024, device=dev), torch.randn(256, device=dev)
x2 = torch.randn(M, 256, device=dev)
cases = [
    ("qkv  N=1536 K=256  P + V^T", 1536, 256, lambda t: ops.gemm_planes(hP, wqkv, M=M, N=1536, K=256, P=qkP, PT=vtP, pt_n0=1024, pt_T=T, pt_zs=512 * vtP.ld, tile=t)),
    ("ff1  N=1024 K=256  P, GELU", 1024, 256, lambda t: ops.gemm_planes(hP, w1, M=M, N=1024, K=256, P=ffP, bias=b1, act=ops.GELU_ERF, tile=t)),
    ("out  N=256  K=512  C + R", 256, 512, lambda t: ops.gemm_planes(attP, wo, M=M, N=256, K=512, C=x2, ldc=256, R=x2, ldr=256, bias=bo, tile=t)),
    ("ff2  N=256  K=1024 C + R", 256, 1024, lambda t: ops.gemm_planes(ffP, w2, M=M, N=256, K=1024, C=x2, ldc=256, R=x2, ldr=256, bias=bo, tile=t)),
]
if os.environ.get("CBX_DF_MICRO_LIB") == "1":  # imported by scripts/diag_loader_forms.py for `cases` / `interleaved`
    cases_only = True
else:
    cases_only = False
if not cases_only:
    print(f"# M = {M} (rows {ROWS} x T {T}); us per launch (fraction of the fp16 dense peak at 3 products per fp32 product)", flush=True)
for name, N, K, fn in ([] if cases_only else cases):
    line = f"{name:30s} |"
    ok = []
    for t in tiles:
        try:
            fn(t)
            ok.append(t)
        except RuntimeError:
            line += f" t{t}: n/a"
    for t, us in zip(ok, interleaved([(lambda t=t: fn(t)) for t in ok])):
        line += f" t{t}: {us:5.1f} ({2.0 * M * N * K * 3 / us / 1e6 / 2500:.3f})"
    print(line, flush=True)
