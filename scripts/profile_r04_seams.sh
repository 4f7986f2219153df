#!/bin/bash
# Round 4, the remaining GPU minutes: the ABI v12 stage seams (cbx_cfm_solve, cbx_hift_decode) on the hardware -- their own tests, the flow / vocoder model
# tests with the seams switched on, and the bench line with and without them on one box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/seams
mkdir -p $O
cd $R
timeout 120 python -m pytest tests/test_zzz_stage_seams_gpu.py -q -m gpu -rfE -p no:cacheprovider > $O/pytest_seams.log 2>&1
tail -3 $O/pytest_seams.log
CBX_FLOW_CSEAM=1 CBX_HIFT_CSEAM=1 timeout 100 python bench.py --steps 4 --warmup 2 --no-alt-precisions --no-streaming --no-cpu-baseline > $O/bench_seams_on.json 2> $O/bench_on.err
timeout 100 python bench.py --steps 4 --warmup 2 --no-alt-precisions --no-streaming --no-cpu-baseline > $O/bench_seams_off.json 2> $O/bench_off.err
for f in on off; do python -c "import json; d=json.load(open('$O/bench_seams_$f.json')); print('$f', d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['config']['stage_seams'])"; done
CBX_FLOW_CSEAM=1 CBX_HIFT_CSEAM=1 timeout 110 python -m pytest tests/test_models_gpu.py -q -m gpu -rfE -p no:cacheprovider -x -k "flow or hift or meanflow or end_to_end or voice or minimum" > $O/pytest_models_seams_on.log 2>&1
tail -3 $O/pytest_models_seams_on.log
