cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_t3 -o t3 -- python $GRAFT_REPO_ROOT/scripts/prof_t3.py 30 8 2>&1 | tail -5
ls $GRAFT_REPO_ROOT/gpurun_out/prof_t3 | head
