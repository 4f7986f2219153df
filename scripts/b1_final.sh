O=gpurun_out/r03/final
mkdir -p $O gpurun_out/r03b
timeout 300 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py -x -q -m gpu -k "decode_attn or turbo or nano or t3" > gpurun_out/r03b/t_da2.log 2>&1; tail -3 gpurun_out/r03b/t_da2.log
for w in "--workload turbo --batch 1:turbo_b1" "--workload nano --batch 1:nano_b1" "--batch 1:mtl_b1"; do
  flags=${w%%:*}; tag=${w##*:}
  timeout 600 python bench.py $flags --no-cpu-baseline --no-alt-precisions --no-streaming > $O/bench_$tag.json 2> $O/bench_$tag.err
  tail -1 $O/bench_$tag.json | cut -c1-200
done
