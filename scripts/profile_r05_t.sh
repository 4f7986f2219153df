#!/bin/bash
# Round 5, call T: LayerNorm from the GEMM epilogue (cbx_gemm_pl_t.ln_w): parity tests, per-launch micro-benchmark, same-box bench pairs (serial + throughput schedule)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/t
mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_planes_gpu.py tests/test_zzz_stage_seams_gpu.py tests/test_baseline_shapes_gpu.py tests/test_models_gpu.py -q -m gpu -p no:cacheprovider -rfE -k "layernorm_epilogue or cfm_solve or flow_t1000 or flow or meanflow or end_to_end or pipelined or vc" > $O/pytest_ln.log 2>&1; tail -4 $O/pytest_ln.log
CBX_REPS=40 CBX_PL_TILES=0 timeout 200 python scripts/bench_planes.py 2> $O/bench_planes.err | grep -E "norm|attn_out|ff2 " | cut -c1-200 | tee $O/bench_planes_ln.log
for f in 1 0; do
CBX_FUSED_LN=$f timeout 300 python bench.py --steps 12 --warmup 3 --no-alt-precisions --no-streaming --no-cpu-baseline --no-autotune > $O/bench_fused_ln_$f.json 2> $O/bench_$f.err
python -c "
import json; d=json.load(open('$O/bench_fused_ln_$f.json'))
print('fused_ln $f: throughput schedule', d['value'], d['ms_per_step'], '| serial', d['other_schedule']['value'], d['other_schedule']['ms_per_step'], d['stage_ms'])
"
done
