#!/bin/bash
# Eleventh GPU call: rows past M read row 0's bytes in the packed x operand (gemv_kernel / gemv_ct_kernel): tests, batch-1 lines, B = 8 sanity.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/eleventh
mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py tests/test_zz_abi_v9_gpu.py -q -m gpu -rfE -p no:cacheprovider -k "gemv or t3 or turbo or nano or tile_variants or col_tiles or two_engines" > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
for w in "--workload turbo --batch 1:turbo_b1" "--workload nano --batch 1:nano_b1" "--batch 1:mtl_b1"; do
  flags=${w%%:*}; tag=${w##*:}
  timeout 300 python bench.py $flags --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precisions --no-streaming --no-autotune > $O/bench_$tag.json 2> $O/bench_$tag.err
  python -c "import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['config'].get('stage_ms_per_step'), d.get('decode_step', {}).get('ms_per_step'), d.get('decode_step', {}).get('frac'))"
done
CBX_AB_SHORT=1 timeout 200 python scripts/decode_ab.py 2>&1 | grep -v amdgpu.ids | head -3
