#!/bin/bash
# round 6, call o: automatic plane-GEMM choice per output kind (q | k | v -> form 35, ff1 -> form 42) in situ; N = 256 shapes on a part-filled chip (streaming round 0: T = 530)
mkdir -p gpurun_out/r06_o
timeout 600 python -m pytest tests/test_planes_gpu.py tests/test_baseline_shapes_gpu.py -x -q -m gpu -k "deferred_epilogue or transposed_column or s3gen_t1000 or estimator_planes" 2>&1 | tail -4
CBX_PL_TILES=0,32,35,41,42 timeout 300 python scripts/df_micro.py 2>&1 | tee gpurun_out/r06_o/df_micro_auto.log
CBX_T=530 CBX_PL_TILES=0,1,2,9,16,32 timeout 300 python scripts/df_micro.py 2>&1 | tee gpurun_out/r06_o/df_micro_t530.log
CBX_ROWS=2 CBX_PL_TILES=0,1,2,9,16,32 timeout 300 python scripts/df_micro.py 2>&1 | tee gpurun_out/r06_o/df_micro_rows2.log
timeout 900 python bench.py --schedule serial --steps 6 --warmup 2 > gpurun_out/r06_o/bench_serial.json 2> gpurun_out/r06_o/bench_serial.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r06_o/bench_serial.json'))
print('serial', d['value'], d['ms_per_step'], d.get('stage_ms'))
for r in d.get('roofline_secondary',[]):
    if 'gemm_pl' in r['kernel']: print(r['frac'], r['avg_launch_us'])
P
