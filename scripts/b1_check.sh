mkdir -p gpurun_out/r03b
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "decode_attn" > gpurun_out/r03b/t_da.log 2>&1; tail -3 gpurun_out/r03b/t_da.log
for w in "--batch 1:mtl_b1" "--workload turbo --batch 1:turbo_b1" "--workload nano --batch 1:nano_b1"; do
  flags=${w%%:*}; tag=${w##*:}
  python bench.py $flags --steps 3 --warmup 1 --no-cpu-baseline --no-streaming --no-alt-precisions --no-fast-mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tag', d['value'], d['ms_per_step'], d['config'].get('stage_ms_per_step'), d.get('decode_step',{}).get('ms_per_step'))"
done
