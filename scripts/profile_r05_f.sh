#!/bin/bash
# Round 5, call F: T3 decode beside ONE kind of flow kernel at a time (scripts/overlap_matrix.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/f
mkdir -p $O
cd $R
timeout 300 python scripts/overlap_matrix.py > $O/overlap_matrix.jsonl 2> $O/overlap_matrix.err
tail -3 $O/overlap_matrix.err; cat $O/overlap_matrix.jsonl | cut -c1-330
