#!/bin/bash
# FIRST GPU call of round 4: the whole -m gpu suite on the ABI v10 build WITHOUT -x (per-test results under profiles/),
# the launch timeline of the decode step (CBX_TRACE side build), the hardware-green decode geometries, one bench line.
#   gpurun --timeout 1500 -- 'bash scripts/profile_r04_first.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/first
mkdir -p $O
cd $R
timeout 700 python -m pytest tests -q -m gpu -rfE --durations=12 -p no:cacheprovider --junitxml=$O/pytest_gpu.xml > $O/pytest_gpu.log 2>&1
tail -25 $O/pytest_gpu.log
timeout 150 bash scripts/trace_decode.sh run $O/trace > $O/trace.log 2>&1; tail -40 $O/trace.log
CBX_GREEN_BUDGET_S=200 timeout 300 python scripts/green_variants.py $O > $O/green.log 2>&1; tail -5 $O/green.log
timeout 300 python bench.py --steps 8 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
