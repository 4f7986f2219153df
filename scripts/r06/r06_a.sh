#!/bin/bash
# round 6, call a: same-box baselines of the tree at the start of the round (headline bench, Turbo / Nano at batch 1)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_a
mkdir -p $O
cd $R
timeout 500 python bench.py --steps 8 --warmup 2 > $O/bench_mtl.json 2> $O/bench_mtl.err
tail -1 $O/bench_mtl.json | cut -c1-300
timeout 300 python bench.py --workload turbo --batch 1 --steps 6 --warmup 2 --no-cpu-baseline --no-streaming > $O/bench_turbo_b1.json 2> $O/bench_turbo_b1.err
tail -1 $O/bench_turbo_b1.json | cut -c1-300
timeout 300 python bench.py --workload nano --batch 1 --steps 6 --warmup 2 --no-cpu-baseline --no-streaming > $O/bench_nano_b1.json 2> $O/bench_nano_b1.err
tail -1 $O/bench_nano_b1.json | cut -c1-300
