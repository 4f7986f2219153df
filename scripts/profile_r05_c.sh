#!/bin/bash
# Round 5, call C: (1) the plane GEMM with 2 / 4 loader waves (tiles 31-37) against the shipped forms, per CFM shape (scripts/bench_planes.py);
# (2) scripts/micro/concur.hip: can a chain of small dependent kernels keep its pace beside chip-filling kernels of another stream (UNPROFILED: the rocprofv3
#     kernel trace of call B shows zero concurrency between the two queues, but it may serialise dispatches itself)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/c
mkdir -p $O
cd $R
timeout 120 $R/scripts/micro/bin/concur 10 > $O/concur.jsonl 2> $O/concur.err
tail -3 $O/concur.err; head -c 2500 $O/concur.jsonl
CBX_REPS=40 CBX_PL_TILES=0,4,8,14,21,22,25,31,32,33,34,35,36,37 timeout 400 python scripts/bench_planes.py > $O/bench_planes_loader_waves.log 2> $O/bench_planes.err
tail -3 $O/bench_planes.err; cat $O/bench_planes_loader_waves.log | cut -c1-400
