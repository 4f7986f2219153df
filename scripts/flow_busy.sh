#!/bin/bash
# usage: [CBX_B=1] [CBX_N=25] bash scripts/flow_busy.sh
# GPU-busy fraction of the flow stage: sum of kernel durations (rocprofv3 kernel trace) of 4 flow + HiFT passes next to their wall time.
mkdir -p gpurun_out/r03b
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CBX_FLOW_NO_AB=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03b/flow_busy -o fb --output-format csv -- python $R/scripts/flow_time.py > $R/gpurun_out/r03b/flow_busy.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03b/flow_busy/**/fb_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print(f"kernel time total {tot/1e6:.1f} ms over 4 passes = {tot/4e6:.1f} ms per pass, {calls//4} launches per pass")
for r in rows[:14]:
    print(r["Name"][:100].ljust(100), r["Calls"], f'{float(r["AverageNs"])/1e3:.1f}', r["Percentage"])
PY
grep "flow/hift" gpurun_out/r03b/flow_busy.log
