#!/bin/bash
# round 6, call b: the batch-1 row path (cbx_gemv_row_f32 / cbx_decode_attn_parts) -- hardware tests + same-box A/B of Turbo / Nano at batch 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_b
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py -q -m gpu -x -k "gemv_row or attn_parts or turbo" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
for wl in turbo nano; do
  for rp in 0 1; do
    CBX_TURBO_TUNE="row_path=$rp" timeout 300 python bench.py --workload $wl --batch 1 --steps 6 --warmup 2 --no-cpu-baseline --no-streaming > $O/bench_${wl}_rp$rp.json 2> $O/bench_${wl}_rp$rp.err
    python - <<PY
import json
d=json.loads(open("$O/bench_${wl}_rp$rp.json").read().strip().splitlines()[-1])
print("$wl row_path=$rp", d["value"], d["stage_ms"], d["decode_step"]["ms_per_step"], d["decode_step"]["frac"])
PY
  done
done
for sp in "row_splits=16,row_chunks=4" "row_splits=16,row_chunks=2" "row_splits=8,row_chunks=8" "row_splits=4,row_chunks=8"; do
  CBX_TURBO_TUNE="$sp" timeout 300 python bench.py --workload turbo --batch 1 --steps 4 --warmup 1 --no-cpu-baseline --no-streaming --no-parity > $O/bench_turbo_$sp.json 2> $O/bench_turbo_$sp.err
  python - <<PY
import json
d=json.loads(open("$O/bench_turbo_$sp.json").read().strip().splitlines()[-1])
print("turbo $sp", d["value"], d["decode_step"]["ms_per_step"])
PY
done
