#!/bin/bash
# Round 5, call D: (1) the plane GEMM tile menu again after the loader waves got compile-time DMA streams (tile 21 = the shipped one-loader form must be back at
# its round-4 time; 31-37 = 2 / 4 loaders); (2) the co-residency experiment on the real kernels (scripts/overlap_polite.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/d
mkdir -p $O
cd $R
timeout 300 python scripts/overlap_polite.py $O/overlap_polite.json > $O/overlap_polite.log 2> $O/overlap_polite.err
tail -3 $O/overlap_polite.err; cat $O/overlap_polite.log
CBX_REPS=40 CBX_PL_TILES=0,4,14,17,21,32,33,35 timeout 300 python scripts/bench_planes.py > $O/bench_planes.log 2> $O/bench_planes.err
tail -3 $O/bench_planes.err; head -12 $O/bench_planes.log | cut -c1-330
