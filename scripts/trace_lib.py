"""(shared by trace_decode.py / trace_decode_turbo.py)  Launch timeline of the T3 decode step from the CBX_TRACE side build (scripts/trace_decode.sh run): every workgroup of the decode kernels logs
the chip-wide 100 MHz counter at its phase boundaries; this script runs a few token steps at the bench shape (B = 8: 16 rows, 30 layers, context
~225) under the hipGraph and eagerly, groups the records into launches and prints, per kernel class, where the time of a launch goes:

    gap    = first workgroup's entry - last workgroup's exit of the PREVIOUS launch          (the dependent-launch boundary)
    ramp   = last workgroup's entry - first workgroup's entry                                (dispatch of the grid)
    phases = median over workgroups of (stamp k - own entry)                                 (see the stamp list of each kernel)
    span   = last exit - first entry                                                         (what rocprofv3 calls the kernel's duration)

Stamps.  gemv: 0 entry, 1 first load batch issued, 2 first K block multiplied (first bytes have landed), 3 K loop done (wave 0), 4 every wave's
partial tile in LDS, 5 epilogue stores issued, 6 stores acknowledged.  attention: 0 entry, 1 new token's q / k / v roped and in LDS, 2 context
walked, 3 output stored, 4 acknowledged.  sampler / embed: 0 entry, 1 done, 2 acknowledged.  Resolution 10 ns."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from chatterbox_amd import ops, synth

out_dir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r04/trace"
os.makedirs(out_dir, exist_ok=True)
lib = ops.lib
lib.cbx_trace_set.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
lib.cbx_trace_set.restype = ctypes.c_int
dev = torch.device("cuda:0")
L = int(os.environ.get("CBX_TRACE_LAYERS", "30"))
B, N = int(os.environ.get("CBX_TRACE_BATCH", "8")), 160
CAP = 1 << 19
buf = torch.zeros(CAP * 8, dtype=torch.int64, device=dev)
cnt = torch.zeros(1, dtype=torch.int32, device=dev)
KIND = {1: "gemv", 2: "attn", 3: "sample", 4: "embed"}


def name(tag):
    k = tag >> 28
    if k == 1:
        n = tag & 0xfffff
        wide_k = bool(tag & 0x8000000)  # K > N: the down / mlp c_proj projections
        base = {3072: "gemv qkv", 4096: "gemv gate|up / c_fc"}.get(n, ("gemv down" if wide_k else "gemv o") if n in (1024, 768) else f"gemv N={n}")
        return base + (" ct" if tag & 0x2000000 else "") + (" +np%d" % ((tag >> 20) & 7) if (tag >> 20) & 7 else "")
    return KIND.get(k, f"kind{k}") + (f" flags={tag & 15}" if k == 2 and tag & 15 else "")


def collect():
    torch.cuda.synchronize()
    n = min(int(cnt.item()), CAP)
    r = buf[: n * 8].view(n, 8).cpu().numpy().astype(np.int64)
    cnt.zero_()
    return r


def analyse(rec, label, f):
    """rec: (n, 8) [tag << 32 | wg, t0 .. t6] in 10 ns ticks."""
    tag, wg = (rec[:, 0] >> 32).astype(np.int64), rec[:, 0] & 0xffffffff
    t = rec[:, 1:].astype(np.float64) * 0.01  # us
    nst = {1: 7, 2: 5, 3: 3, 4: 3}
    order = np.argsort(t[:, 0], kind="stable")
    tag, wg, t = tag[order], wg[order], t[order]
    # launches: maximal runs of one tag in entry order whose workgroup index 0 appears once
    launches, start = [], 0
    for i in range(1, len(tag) + 1):
        if i == len(tag) or tag[i] != tag[start] or (wg[i] == 0 and (wg[start:i] == 0).any()):
            launches.append((start, i))
            start = i
    rows, prev_end = {}, None
    for a, b in launches:
        k = int(tag[a]) >> 28
        ns = nst.get(k, 3)
        te = t[a:b, ns - 1]
        ent = t[a:b, 0]
        d = dict(wgs=b - a, ramp=ent.max() - ent.min(), span=te.max() - ent.min(), gap=None if prev_end is None else ent.min() - prev_end,
                 phases=[float(np.median(t[a:b, j] - ent)) for j in range(1, ns)], p90=[float(np.percentile(t[a:b, j] - ent, 90)) for j in range(1, ns)])
        prev_end = te.max()
        rows.setdefault(name(int(tag[a])), []).append(d)
    print(f"\n== {label}: {len(rec)} records, {len(launches)} launches", file=f)
    print(f"{'kernel':22s} {'n':>4s} {'wgs':>5s} {'gap':>6s} {'ramp':>6s} {'span':>6s}   median (p90) of stamp - entry [us]", file=f)
    summ = {}
    for kname, ds in rows.items():
        gaps = [d["gap"] for d in ds if d["gap"] is not None and d["gap"] < 50]
        med = lambda xs: float(np.median(xs)) if len(xs) else float("nan")
        ph = np.median(np.array([d["phases"] for d in ds]), 0)
        p9 = np.median(np.array([d["p90"] for d in ds]), 0)
        s = dict(n=len(ds), wgs=int(np.median([d["wgs"] for d in ds])), gap=med(gaps), ramp=med([d["ramp"] for d in ds]), span=med([d["span"] for d in ds]),
                 phases=[round(float(x), 2) for x in ph], p90=[round(float(x), 2) for x in p9])
        summ[kname] = s
        print(f"{kname:22s} {s['n']:4d} {s['wgs']:5d} {s['gap']:6.2f} {s['ramp']:6.2f} {s['span']:6.2f}   " +
              "  ".join(f"{a:5.2f} ({b:5.2f})" for a, b in zip(s["phases"], s["p90"])), file=f)
    tot = t[:, :].max() - t[:, 0].min()
    print(f"first entry -> last stamp: {tot:.1f} us", file=f)
    return summ


