export CBX_GEMM_SHAPES="qkv,attn_out,ff2,res1x1,big"
for dbg in 0 3; do echo "DBG=$dbg (1=no global loads in loop, 2=no MFMA)"; CBX_GEMM_DBG=$dbg timeout 100 python scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids; done
timeout 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -x 2>&1 | tail -3
