#!/bin/bash
# Third GPU call of round 4: the ABI v11 kernels (column-tile / split-K q/k/v + head GEMV, attention folding the partial sums): tests, A/B, timeline.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/third
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_zz_abi_v9_gpu.py tests/test_ops_gpu.py -q -m gpu -rfE -p no:cacheprovider \
    -k "col_tiles or folds_qkv or tile_variants or c_level_decode_step or decode_attn" --junitxml=$O/pytest_gpu.xml > $O/pytest_gpu.log 2>&1
tail -8 $O/pytest_gpu.log
timeout 300 python scripts/decode_ab.py $O/decode_ab.json > $O/decode_ab.log 2>&1; cat $O/decode_ab.log | grep -v amdgpu.ids
CBX_TRACE_VARIANTS=0 CBX_TRACE_TUNE="qkv_ks=4,qkv_ct=3,head_ct=2" timeout 200 bash scripts/trace_decode.sh run $O/trace > $O/trace.log 2>&1; tail -24 $O/trace.log
