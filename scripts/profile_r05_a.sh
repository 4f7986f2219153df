#!/bin/bash
# Round 5, call A: the two stage seams that had only run on the emulator (cbx_s3gen_encode, cbx_hift_f0_source), then every model golden THROUGH the
# four C entry points, then a same-box bench pair seams off / on.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/a
mkdir -p $O
cd $R
CBX_TEST_PENDING_SEAMS=1 timeout 200 python -m pytest tests/test_zzz_stage_seams_gpu.py -q -m gpu -rfEs -p no:cacheprovider > $O/pytest_seams.log 2>&1
tail -4 $O/pytest_seams.log
CBX_FLOW_CSEAM=1 CBX_HIFT_CSEAM=1 timeout 300 python -m pytest tests/test_models_gpu.py tests/test_baseline_shapes_gpu.py -q -m gpu -rfEs -p no:cacheprovider > $O/pytest_models_seams_on.log 2>&1
tail -4 $O/pytest_models_seams_on.log
CBX_FLOW_CSEAM=1 CBX_HIFT_CSEAM=1 timeout 200 python bench.py --steps 5 --warmup 2 --no-alt-precisions --no-streaming --no-cpu-baseline > $O/bench_all_seams.json 2> $O/bench_seams.err
python -c "import json; d=json.load(open('$O/bench_all_seams.json')); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['config'].get('stage_seams'), d['parity'])"
timeout 200 python bench.py --steps 5 --warmup 2 --no-alt-precisions --no-streaming --no-cpu-baseline > $O/bench_seams_off.json 2> $O/bench_off.err
python -c "import json; d=json.load(open('$O/bench_seams_off.json')); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['parity'])"
