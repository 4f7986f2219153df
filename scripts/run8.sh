cd /tmp && export TMPDIR=/tmp
export CBX_GEMM_SHAPES="qkv,attn_out"
R=$GRAFT_REPO_ROOT
timeout 100 python $R/scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids
timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $R/gpurun_out/pmc2 -o g -- python $R/scripts/bench_gemm.py 2>&1 | grep -v "simple_timer\|amdgpu" | tail -3
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc3 -o g -- python $R/scripts/bench_gemm.py 2>&1 | grep -v "simple_timer\|amdgpu" | tail -3
rm -f $R/gpurun_out/pmc2/*trace.csv $R/gpurun_out/pmc3/*trace.csv
