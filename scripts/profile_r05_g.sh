#!/bin/bash
# Round 5, call G: the decode-step kernels raise their wave priority (s_setprio 3): co-residency experiment + per-kernel matrix again
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/g
mkdir -p $O
cd $R
CBX_OV_ATTN=5 timeout 300 python scripts/overlap_polite.py $O/overlap_polite.json > $O/overlap_polite.log 2> $O/overlap_polite.err
tail -3 $O/overlap_polite.err; cat $O/overlap_polite.log | cut -c1-250
timeout 300 python scripts/overlap_matrix.py > $O/overlap_matrix.jsonl 2> $O/overlap_matrix.err
tail -3 $O/overlap_matrix.err; cat $O/overlap_matrix.jsonl | cut -c1-250
