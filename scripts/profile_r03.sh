#!/bin/bash
# Round-3 measurement set (one GPU call): the HBM-traffic PMC passes FIRST (separate FETCH_SIZE / WRITE_SIZE runs, kernel-trace only, of the S3Gen
# pass and an eager T3 decode: bench.py reads roofline.traffic from their summaries), then the driver's bench command, rocprofv3 kernel
# stats of the SAME command, and the other workloads' lines.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_flow_$c -o p -- python $R/scripts/flow_only.py > /tmp/log_flow_$c.txt 2>&1
  f=$(find /tmp/pmc_flow_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py $f $O/flow_only_pmc_$c.csv
  CBX_STEPS=6 timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_t3_$c -o p -- python $R/scripts/prof_t3_eager.py > /tmp/log_t3_$c.txt 2>&1
  f=$(find /tmp/pmc_t3_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py $f $O/t3_eager_pmc_$c.csv
done
# the bench reads the PMC summaries from profiles/: make this call's available to it
cp $O/flow_only_pmc_FETCH_SIZE.csv $R/profiles/r03_flow_only_pmc_FETCH_SIZE.csv 2>/dev/null
cp $O/flow_only_pmc_WRITE_SIZE.csv $R/profiles/r03_flow_only_pmc_WRITE_SIZE.csv 2>/dev/null
cp $O/t3_eager_pmc_FETCH_SIZE.csv $R/profiles/r03_t3_eager_pmc_FETCH_SIZE.csv 2>/dev/null
cp $O/t3_eager_pmc_WRITE_SIZE.csv $R/profiles/r03_t3_eager_pmc_WRITE_SIZE.csv 2>/dev/null
cd $R
CBX_BENCH_VERBOSE=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench.err
tail -1 $O/bench_steps20_warmup5.json | cut -c1-300
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> /tmp/rocprof_bench.err
cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) $O/bench_steps20_warmup5_kernel_stats.csv
head -14 $O/bench_steps20_warmup5_kernel_stats.csv | cut -c1-170
cd $R
for w in "--workload turbo --batch 1:turbo_b1" "--workload nano --batch 1:nano_b1" "--batch 1:mtl_b1" "--batch 32 --steps 4 --warmup 1:mtl_b32" "--config3 --steps 3 --warmup 1:config3"; do
  flags=${w%%:*}; tag=${w##*:}
  timeout 600 python bench.py $flags --no-cpu-baseline --no-alt-precisions --no-streaming > $O/bench_$tag.json 2> $O/bench_$tag.err
  tail -1 $O/bench_$tag.json | cut -c1-200
done
CBX_GEMM_SHAPES=t3_prefill_qkv,t3_prefill_o,t3_prefill_down,big CBX_PRECS=1 timeout 200 python scripts/bench_gemm.py > $O/bench_gemm_f32_prefill.log 2>&1
cat $O/bench_gemm_f32_prefill.log | tail -5
ls -la $O
