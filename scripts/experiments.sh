#!/bin/bash
# The four opt-in experiments written after the round-1 GPU budget was spent: correctness first, then an
# A/B of the bench step on ONE box.  Usage on the GPU box:  bash scripts/experiments.sh   (about 4 minutes)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out; L=gpurun_out/experiments.log; : > $L
echo "== gated tests (multi-stream T3, fused add+norm GEMV)" >> $L
CBX_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests -x -q -m gpu -k "streams_equal or fused_norm or gemv_norm_fused or minimum_sizes" 2>&1 | tail -6 >> $L
echo "== gated test (A-stationary K=256 split GEMM)" >> $L
CBX_TEST_EXPERIMENTAL=1 CBX_SPLIT_AK=1 timeout 200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "split_gemm" 2>&1 | tail -4 >> $L
run() {  # label, env..., bench args
  local label=$1; shift
  echo "== $label" >> $L
  env "$@" CBX_BENCH_VERBOSE=1 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $EXTRA 2>&1 | grep -E "step seed=[01]|\"metric\"" | cut -c1-260 >> $L
}
EXTRA="" run "default" X=1
EXTRA="--t3-streams 2" run "T3 as 2 concurrent sub-batches" X=1
EXTRA="--t3-streams 4" run "T3 as 4 concurrent sub-batches" X=1
EXTRA="" run "fused add+norm GEMV (5 launches per layer)" CBX_T3_FUSED=1
EXTRA="--t3-streams 2" run "fused + 2 streams" CBX_T3_FUSED=1
EXTRA="" run "A-stationary K=256 split GEMM" CBX_SPLIT_AK=1
cat $L
