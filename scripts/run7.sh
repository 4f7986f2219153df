timeout 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -x 2>&1 | tail -3
export CBX_GEMM_SHAPES="qkv,attn_out,ff1+gelu,ff2,conv3_256,res1x1,t3_prefill_o,big"
timeout 100 python scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids
export CBX_BENCH_VERBOSE=1
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -3
