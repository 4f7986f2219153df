#!/bin/bash
# round 6, call g: same-box A/B of the streaming schedules; the pipelined test with one host thread
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_g
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_models_gpu.py tests/test_stream_gpu.py -q -m gpu -k "pipelined or stream" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
timeout 900 python scripts/stream_ab.py > $O/stream_ab.jsonl 2> $O/stream_ab.err
cat $O/stream_ab.jsonl
tail -3 $O/stream_ab.err
