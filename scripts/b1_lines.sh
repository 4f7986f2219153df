O=gpurun_out/r03/final
mkdir -p $O
for w in "--workload turbo --batch 1:turbo_b1" "--workload nano --batch 1:nano_b1"; do
  flags=${w%%:*}; tag=${w##*:}
  timeout 200 python bench.py $flags --no-cpu-baseline --no-alt-precisions --no-streaming > $O/bench_$tag.json 2> $O/bench_$tag.err
  tail -1 $O/bench_$tag.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'], d['config'].get('stage_ms_per_step'), d['decode_step']['ms_per_step'], d['decode_step']['frac'])"
done
