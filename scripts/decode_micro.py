"""Microbenchmarks of the T3 decode kernels (run on the GPU box):  python scripts/decode_micro.py [> gpurun_out/decode_micro.log]

Each configuration is a hipGraph of L = 30 dependent launches (one per layer, DISTINCT weights per layer unless "same" so that nothing
is cache-resident by accident), replayed REPS times and timed with events: the number is the average cost of one launch inside a
dependent chain, i.e. what a decode step pays.  Variants: row-major vs lane-ordered packed weights (w_packed), packed x operand,
cross-kernel prefetch of the next launch's weights into the Infinity Cache, split-K geometry.
"""
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatterbox_amd import ops  # noqa: E402

dev = torch.device("cuda")
L, REPS, M = 30, 20, 16
torch.manual_seed(0)


def chain_us(fns, reps=REPS):
    """fns: list of zero-arg launch closures -> average microseconds per launch when replayed as one graph."""
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(fns))


def pack_x(x):
    """(16, K) -> packed operand image (same layout as the weights)."""
    return ops.pack_gemv_weight(x)


def report(name, us, nbytes):
    print(f"{name:58s} {us:7.2f} us   {nbytes / us * 1e-6:6.2f} TB/s", flush=True)


def gemv_family(tag, N, K, swiglu, ksplit, nw):
    rows_w = 2 * N if swiglu else N
    Ws = [torch.randn(rows_w, K, device=dev) * 0.03 for _ in range(L)]
    Wp = [ops.pack_gemv_weight(w, swiglu=swiglu) for w in Ws]
    # row-major swiglu image of the existing kernel: [32 gate | 32 up] interleaved
    if swiglu:
        Wr = [torch.stack([w[:N].view(N // 32, 32, K), w[N:].view(N // 32, 32, K)], 1).reshape(2 * N, K).contiguous() for w in Ws]
    else:
        Wr = Ws
    x = torch.randn(M, K, device=dev)
    xp = pack_x(x)
    out = torch.empty(ksplit, M, N, device=dev) if ksplit > 1 else torch.empty(M, N, device=dev)
    nbytes = 4.0 * rows_w * K
    kw = dict(N=N, ksplit=ksplit, nw=nw, swiglu=swiglu)

    # correctness of the packed path against the row-major one (same arithmetic order -> bit-identical)
    o1 = torch.empty_like(out)
    o2 = torch.empty_like(out)
    o3 = torch.empty_like(out)
    ops.gemv(x, Wr[0], o1, **kw)
    ops.gemv(x, Wp[0], o2, w_packed=True, **kw)
    ops.gemv(xp, Wp[0], o3, w_packed=True, x_packed=True, M=M, K=K, **kw)
    torch.cuda.synchronize()
    ref = x @ Ws[0].t()
    if swiglu:
        ref = torch.nn.functional.silu(ref[:, :N]) * ref[:, N:]
    got = o1.sum(0) if ksplit > 1 else o1
    assert torch.equal(o1, o2) and torch.equal(o1, o3), f"{tag}: packed path differs from row-major"
    assert (got - ref).abs().max() < 2e-3, f"{tag}: {(got - ref).abs().max()}"

    report(f"{tag} row-major, rotating W", chain_us([lambda i=i: ops.gemv(x, Wr[i], out, **kw) for i in range(L)]), nbytes)
    report(f"{tag} row-major, same W (cache-resident)", chain_us([lambda: ops.gemv(x, Wr[0], out, **kw) for i in range(L)]), nbytes)
    report(f"{tag} packed W", chain_us([lambda i=i: ops.gemv(x, Wp[i], out, w_packed=True, **kw) for i in range(L)]), nbytes)
    report(f"{tag} packed W + packed x",
           chain_us([lambda i=i: ops.gemv(xp, Wp[i], out, w_packed=True, x_packed=True, M=M, K=K, **kw) for i in range(L)]), nbytes)
    report(f"{tag} packed W + packed x, same W",
           chain_us([lambda: ops.gemv(xp, Wp[0], out, w_packed=True, x_packed=True, M=M, K=K, **kw) for i in range(L)]), nbytes)
    for stride in (128, 64):
        report(f"{tag} packed W + packed x + prefetch next (stride {stride})",
               chain_us([lambda i=i: ops.gemv(xp, Wp[i], out, w_packed=True, x_packed=True, M=M, K=K, prefetch=Wp[(i + 1) % L],
                                              pf_stride=stride, **kw) for i in range(L)]), nbytes)
    del Ws, Wp, Wr


def main():
    print(torch.cuda.get_device_name(0))
    # floor: a trivial dependent chain
    t = torch.zeros(64, device=dev)
    report("trivial kernel chain (axpby on 64 floats)", chain_us([lambda: ops.axpby(t.view(1, -1), t.view(1, -1), a=1.0, b=0.0) for _ in range(L)]), 1.0)

    gemv_family("qkv  N=3072 K=1024 ks=1 nw=8", 3072, 1024, False, 1, 8)
    gemv_family("qkv  N=3072 K=1024 ks=2 nw=4", 3072, 1024, False, 2, 4)
    gemv_family("qkv  N=3072 K=1024 ks=4 nw=4", 3072, 1024, False, 4, 4)
    gemv_family("o    N=1024 K=1024 ks=4 nw=4", 1024, 1024, False, 4, 4)
    gemv_family("o    N=1024 K=1024 ks=2 nw=8", 1024, 1024, False, 2, 8)
    gemv_family("gu   F=4096 K=1024 ks=1 nw=8", 4096, 1024, True, 1, 8)
    gemv_family("gu   F=4096 K=1024 ks=1 nw=4", 4096, 1024, True, 1, 4)
    gemv_family("down N=1024 K=4096 ks=8 nw=4", 1024, 4096, False, 8, 4)
    gemv_family("down N=1024 K=4096 ks=4 nw=8", 1024, 4096, False, 4, 8)

    # add + rmsnorm (16 rows), attention at the bench's mean context
    x = torch.randn(M, 1024, device=dev)
    h = torch.empty_like(x)
    w = torch.ones(1024, device=dev)
    for ks in (4, 8):
        part = torch.randn(ks, M, 1024, device=dev)
        report(f"add_rmsnorm 16 rows, {ks} partials", chain_us([lambda: ops.add_rmsnorm(x, part, w, h) for _ in range(L)]), 4.0 * (ks + 2) * M * 1024)
    from chatterbox_amd.t3 import llama3_rope_tables
    cos, sin = llama3_rope_tables(1024)
    cos, sin = cos.to(dev), sin.to(dev)
    H = 16
    for ctx in (100, 225, 350):
        max_ctx = 384
        kc = [torch.randn(M, H, max_ctx, 64, device=dev) for _ in range(L)]
        vc = [torch.randn(M, H, max_ctx, 64, device=dev) for _ in range(L)]
        qkv = torch.randn(M, 3072, device=dev)
        att = torch.empty(M, 1024, device=dev)
        pos = torch.full((M,), ctx - 1, dtype=torch.int32, device=dev)
        report(f"decode_attn_rope ctx={ctx}", chain_us([lambda i=i: ops.decode_attn_rope(qkv, pos, cos, sin, kc[i], vc[i], att, 0.125) for i in range(L)]),
               2.0 * M * H * ctx * 256)
        del kc, vc


if __name__ == "__main__":
    main()
