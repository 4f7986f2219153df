#!/bin/bash
# Round 5, call E: the co-residency experiment with the 4-wave plane attention (version 5) + its parity tests + per-launch time
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/e
mkdir -p $O
cd $R
timeout 200 python -m pytest tests/test_planes_gpu.py -q -m gpu -x -p no:cacheprovider -k "flash_attn_planes" > $O/pytest_attn.log 2>&1; tail -2 $O/pytest_attn.log
CBX_OV_ATTN=5 timeout 300 python scripts/overlap_polite.py $O/overlap_polite.json > $O/overlap_polite.log 2> $O/overlap_polite.err
tail -3 $O/overlap_polite.err; cat $O/overlap_polite.log
