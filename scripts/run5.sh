cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z0-9_]+|GRBM_[A-Z_]+|TCC_[A-Z0-9_]+|TCP_[A-Z0-9_]+)\b" | sort -u | tr '\n' ' ' | head -c 6000
echo
export CBX_GEMM_SHAPES="attn_out,big,qkv"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc1 -o g -- python $GRAFT_REPO_ROOT/scripts/bench_gemm.py 2>&1 | grep -v simple_timer | tail -5
ls $GRAFT_REPO_ROOT/gpurun_out/pmc1
