#!/bin/bash
# Last GPU call of round 4: the C prefill seam on the hardware (its own test, the T3 goldens that now run through it), a short bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/last
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_zz_abi_v9_gpu.py tests/test_baseline_shapes_gpu.py tests/test_models_gpu.py -q -m gpu -rfE -p no:cacheprovider -k "prefill or b8_250 or t3_vs_reference or ragged or bf16_weight" > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-alt-precisions --no-streaming > $O/bench_steps5.json 2> $O/bench.err
python -c "import json; d=json.load(open('$O/bench_steps5.json')); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['t3_geometry'], d['parity']['tokens_equal'], d['parity']['wav_rmse'])"
