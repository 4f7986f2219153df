#!/bin/bash
# Second GPU call of round 4: the three tests the first call lost to a NULL-workspace bug, the launch timeline, the green list, one bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/second
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_baseline_shapes_gpu.py tests/test_zz_abi_v9_gpu.py -q -m gpu -rfE -p no:cacheprovider \
    -k "b8_250 or autotuner or two_engines or caller_owned" --junitxml=$O/pytest_gpu.xml > $O/pytest_gpu.log 2>&1
tail -8 $O/pytest_gpu.log
timeout 200 bash scripts/trace_decode.sh run $O/trace > $O/trace.log 2>&1; tail -60 $O/trace.log
CBX_GREEN_BUDGET_S=220 timeout 320 python scripts/green_variants.py $O > $O/green.log 2>&1; tail -4 $O/green.log
timeout 300 python bench.py --steps 8 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -c 1200 $O/bench.json; tail -c 400 $O/bench.err
