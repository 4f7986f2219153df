# Turbo batch 1 with 1000 speech tokens (context up to ~1450): decode attention with / without the context split, and a kernel profile of the default run
mkdir -p gpurun_out/r03b
run() {
  env $1 python bench.py --workload turbo --batch 1 --tokens 1000 --steps 2 --warmup 1 --no-cpu-baseline --no-streaming --no-alt-precisions --no-fast-mode --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('turbo 1000 tokens [$1]', d['value'], d['ms_per_step'], d.get('decode_step',{}).get('ms_per_step'))"
}
run "X=1"; run "CBX_DA_NO_SPLIT=1"; run "CBX_DA_SPLIT_MIN=512"; run "CBX_DA_NO_SPLIT=1 CBX_DA_U=16"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_turbo -o t -- python $GRAFT_REPO_ROOT/bench.py --workload turbo --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-streaming --no-alt-precisions --no-fast-mode --no-parity > /dev/null 2>&1
cp $(find /tmp/prof_turbo -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r03b/turbo_b1_kernel_stats.csv
head -12 $GRAFT_REPO_ROOT/gpurun_out/r03b/turbo_b1_kernel_stats.csv | cut -c1-150
