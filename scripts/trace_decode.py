"""Launch timeline of the T3 decode step from the CBX_TRACE side build (scripts/trace_decode.sh run): every workgroup of the decode kernels logs
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
from chatterbox_amd.t3 import T3Engine

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
        return {3072: "gemv qkv", 1024: "gemv o/down", 4096: "gemv gate|up"}.get(n, f"gemv N={n}") + (" ct" if tag & 0x2000000 else "") + (" +np%d" % ((tag >> 20) & 15) if (tag >> 20) & 15 else "")
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


eng = T3Engine(synth.t3_state_dict(L, 0), dev, n_layers=L)
texts = [synth.text_tokens(64, seed=b) for b in range(B)]
u = synth.rand((B, N), seed=1).to(dev)
kw = dict(max_new_tokens=N, uniforms=u, ban_eos=True, ban_from=6561)
variants = [("default", {}, {})]
if os.environ.get("CBX_TRACE_TUNE"):  # e.g. CBX_TRACE_TUNE="qkv_ks=4,qkv_ct=3,head_ct=2,da_pipe=7"
    from chatterbox_amd.autotune import split_variant
    tv, kv = split_variant({k: int(x) for k, x in (kv.split("=") for kv in os.environ["CBX_TRACE_TUNE"].split(","))})
    variants = [(os.environ["CBX_TRACE_TUNE"], tv, kv)]
if os.environ.get("CBX_TRACE_VARIANTS", "1") == "1":
    variants += [("da_pipe=7,pre_epi=1", {}, dict(da_pipe=7, pre_epi=1)), ("qkv_tc=12,od_tc=4,d_ks2=1,d_nw2=8", dict(qkv_tc=12, od_tc=4, d_ks2=1, d_nw2=8), {})]
result = {}
with open(os.path.join(out_dir, "trace_decode.txt"), "w") as f:
    for label, tune, knobs in variants:
        kn = dict(eng.knobs, da_pipe=0, pre_epi=0, deep=0, da_u=4)
        kn.update(knobs)
        eng.apply_variant(dict(T3Engine._TUNE, **tune), kn)
        for graph in (True, False):
            h = eng.generate(synth.t3_cond(), texts, async_mode=True, run_steps=N - 8, use_graph=graph, **kw)  # context ~ 100 + 150
            torch.cuda.synchronize()
            assert lib.cbx_trace_set(buf.data_ptr(), cnt.data_ptr(), CAP) == 0
            cnt.zero_()
            eng.advance(h, 4)
            rec = collect()
            assert lib.cbx_trace_set(None, None, 0) == 0
            np.save(os.path.join(out_dir, f"trace_{label.replace(',', '_').replace('=', '')}_{'graph' if graph else 'eager'}.npy"), rec)
            result[f"{label} | {'hipGraph' if graph else 'eager'}"] = analyse(rec, f"{label}, B = {B}, {L} layers, {'hipGraph replays' if graph else 'eager launches'}, 4 token steps", f)
            f.flush()
    json.dump(result, open(os.path.join(out_dir, "trace_decode.json"), "w"), indent=1)
print(open(os.path.join(out_dir, "trace_decode.txt")).read())
