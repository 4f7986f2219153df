#!/bin/bash
# Fourth GPU call of round 4: rope_rows (cos | sin gathered once per token step), the ABI v11 tests again, A/B.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/fourth
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_zz_abi_v9_gpu.py tests/test_ops_gpu.py -q -m gpu -rfE -p no:cacheprovider \
    -k "col_tiles or folds_qkv or tile_variants or c_level_decode_step or embed" --junitxml=$O/pytest_gpu.xml > $O/pytest_gpu.log 2>&1
tail -6 $O/pytest_gpu.log
timeout 300 python scripts/decode_ab.py $O/decode_ab.json > $O/decode_ab.log 2>&1; cat $O/decode_ab.log | grep -v amdgpu.ids
