mkdir -p gpurun_out/r03b
timeout 240 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py -x -q -m gpu -k "sampl or decode_attn or turbo or nano or t3" > gpurun_out/r03b/t_final.log 2>&1; tail -3 gpurun_out/r03b/t_final.log
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_turbo -o t -- python $GRAFT_REPO_ROOT/bench.py --workload turbo --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-streaming --no-alt-precisions --no-fast-mode --no-parity > /tmp/turbo_line.json 2>/dev/null
cp $(find /tmp/prof_turbo -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r03b/turbo_b1_kernel_stats_final.csv
grep -h "t3_sample\|decode_attn" $GRAFT_REPO_ROOT/gpurun_out/r03b/turbo_b1_kernel_stats_final.csv | cut -c1-160
python -c "
import json
d=json.loads(open('/tmp/turbo_line.json').read().strip().splitlines()[-1]); print('turbo under rocprof', d['value'], d['ms_per_step'], d['decode_step']['ms_per_step'])"
