"""Reads the rocprofv3 kernel trace of scripts/overlap_trace.py (argv[1] = *_kernel_trace.csv, argv[2] = the phase list it wrote) and prints, per phase:
wall span, per queue busy time (union of kernel intervals), time both queues are busy at once, and per kernel class launches / mean duration / mean
START-TO-START distance inside its own queue (what a dependent chain pays per launch) -- solo against overlapped.  Also the resources of every
kernel class (VGPRs, LDS, workgroup size, grid) that decide whether workgroups of the two streams can be co-resident on a CU."""
import csv
import json
import re
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append(r)
key = lambda r, *names: next((r[n] for n in names if n in r), None)
ev = []
for r in rows:
    name = key(r, "Kernel_Name", "Name")
    s, e = int(key(r, "Start_Timestamp", "BeginNs")), int(key(r, "End_Timestamp", "EndNs"))
    q = key(r, "Queue_Id", "Queue_ID", "queue_id", "Stream_Id")
    if q is None:  # no queue column: classify by kernel family (the T3 decode step's kernels against everything else)
        q = "t3" if re.search(r"gemv|decode_attn|t3_sample|embed_kernel|add_rmsnorm", name) else "flow"
    ev.append((s, e, q, name, r))
ev.sort()
phases = json.load(open(sys.argv[2]))
# split at idle gaps >= 60 ms; the LAST len(phases) segments are the announced phases (the model build / warm-up come first)
segs, cur, last_end = [], [], None
for x in ev:
    if last_end is not None and x[0] - last_end >= 60e6 and cur:
        segs.append(cur)
        cur = []
    cur.append(x)
    last_end = x[1] if last_end is None else max(last_end, x[1])
if cur:
    segs.append(cur)
print(f"# {len(ev)} kernel records, {len(segs)} segments, {len(phases)} phases", flush=True)
segs = segs[-len(phases):]


def cls(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z0-9_]+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:60]


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    if cs is not None:
        tot += ce - cs
    return tot


def both_busy(a, b):
    pts = [(s, 1, 0) for s, e in a] + [(e, -1, 0) for s, e in a] + [(s, 1, 1) for s, e in b] + [(e, -1, 1) for s, e in b]
    pts.sort()
    c, tot, prev = [0, 0], 0, None
    for t, d, w in pts:
        if prev is not None and c[0] > 0 and c[1] > 0:
            tot += t - prev
        c[w] += d
        prev = t
    return tot


res = {}
solo = {}
for ph, seg in zip(phases, segs):
    byq = defaultdict(list)
    for s, e, q, n, r in seg:
        byq[q].append((s, e, n, r))
    qs = sorted(byq, key=lambda q: -len(byq[q]))[:2]
    span = (max(e for s, e, *_ in seg) - min(s for s, *_ in seg)) / 1e6
    line = dict(config=ph["config"], what=ph["what"], host_wall_ms=ph["wall_ms"], gpu_span_ms=round(span, 1), kernels=len(seg), queues=len(byq))
    for i, q in enumerate(qs):
        line[f"queue{i}_kernels"] = len(byq[q])
        line[f"queue{i}_busy_ms"] = round(union([(s, e) for s, e, *_ in byq[q]]) / 1e6, 1)
    if len(qs) == 2:
        line["both_queues_busy_ms"] = round(both_busy([(s, e) for s, e, *_ in byq[qs[0]]], [(s, e) for s, e, *_ in byq[qs[1]]]) / 1e6, 1)
    print(json.dumps(line), flush=True)
    per = defaultdict(lambda: [0, 0.0, 0.0, 0])
    for q in byq:
        ks = sorted(byq[q])
        for i, (s, e, n, r) in enumerate(ks):
            c = cls(n)
            p = per[c]
            p[0] += 1
            p[1] += (e - s) / 1e3
            if i + 1 < len(ks):
                p[2] += (ks[i + 1][0] - s) / 1e3
                p[3] += 1
    res[(ph["config"], ph["what"])] = per
    top = sorted(per.items(), key=lambda kv: -kv[1][1])[:14]
    for c, (n, dur, s2s, ns) in top:
        ref = res.get((ph["config"], "t3"), {}).get(c) or res.get((ph["config"], "voc"), {}).get(c)
        extra = ""
        if ph["what"] == "both" and ref and ref[0]:
            extra = f"   solo {ref[1] / ref[0]:8.2f} us, start-to-start {ref[2] / max(1, ref[3]):8.2f}"
        print(f"    {c[:70]:70s} n {n:6d}  mean {dur / n:8.2f} us  start-to-start {s2s / max(1, ns):8.2f} us{extra}")
# resources per kernel class
seen = {}
for s, e, q, n, r in ev:
    c = cls(n)
    if c not in seen:
        seen[c] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size_X", "Grid_Size_X") if k in r}
print("# resources per kernel class (first launch seen)")
for c, v in sorted(seen.items()):
    print(f"    {c[:80]:80s} {v}")
