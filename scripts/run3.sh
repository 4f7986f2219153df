cd /tmp && export TMPDIR=/tmp
export CBX_BENCH_VERBOSE=1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -v "simple_timer\|amdgpu.ids" | tail -8
ls -la $GRAFT_REPO_ROOT/gpurun_out/prof_bench | head
rm -f $GRAFT_REPO_ROOT/gpurun_out/prof_bench/*kernel_trace.csv
