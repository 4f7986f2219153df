#!/bin/bash
# Round 5, call H: per-stream start / end events of the overlapped run (scripts/overlap_probe.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/h
mkdir -p $O
cd $R
timeout 300 python scripts/overlap_probe.py > $O/overlap_probe.jsonl 2> $O/overlap_probe.err
tail -3 $O/overlap_probe.err; cat $O/overlap_probe.jsonl | cut -c1-330
