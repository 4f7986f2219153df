"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel (sum / launches / per-launch mean).  Usage: pmc_summary.py in.csv out.csv"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(sys.argv[1])):
    a = agg[r["Kernel_Name"]][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"])
    a[1] += 1
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "launches", "sum", "mean_per_launch"])
    for k, d in sorted(agg.items(), key=lambda kv: -max(v[0] for v in kv[1].values())):
        for c, (s, n) in d.items():
            w.writerow([k[:160], c, n, f"{s:.0f}", f"{s / n:.1f}"])
