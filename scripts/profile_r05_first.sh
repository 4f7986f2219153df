#!/bin/bash
# First GPU call of round 5 (prepared at the end of round 4, when its GPU minutes were spent): the whole -m gpu suite INCLUDING the two stage seams that
# have only run on the emulator (cbx_s3gen_encode, cbx_hift_f0_source: CBX_TEST_PENDING_SEAMS=1), no -x, per-test results; then the driver's bench command.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/first
mkdir -p $O
cd $R
CBX_TEST_PENDING_SEAMS=1 timeout 1000 python -m pytest tests -q -m gpu -rfEs -p no:cacheprovider --junitxml=$O/pytest_gpu.xml > $O/pytest_gpu.log 2>&1
tail -6 $O/pytest_gpu.log
CBX_BENCH_VERBOSE=1 timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench.err
python -c "import json; d=json.load(open('$O/bench_steps20_warmup5.json')); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['t3_geometry'], d['decode_step']['ms_per_step'], d['decode_step']['frac'], d['roofline']['frac'], d['parity'])"
# every S3Gen / HiFT stage through its C sequencer, same box
CBX_FLOW_CSEAM=1 CBX_HIFT_CSEAM=1 timeout 200 python bench.py --steps 5 --warmup 2 --no-alt-precisions --no-streaming > $O/bench_all_seams.json 2> $O/bench_seams.err
python -c "import json; d=json.load(open('$O/bench_all_seams.json')); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['config']['stage_seams'], d['parity'])"
