// MFMA issue rate of v_mfma_f32_32x32x16_f16 under the accumulator dependency patterns of the f16x3 kernels, one or two waves per SIMD,
// and beside a VALU-only partner wave (does the matrix pipe overlap with another wave's VALU?).  Registers only, no memory.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_dep.hip -o /tmp/md && /tmp/md
//   P = 0: the mma3 order of attention_planes / gemm_planes:  accc += ah bl; accc += al bh (depends on the previous one); acc += ah bh
//   P = 1: same work, no two consecutive MFMAs on one accumulator (two tiles interleaved: c0, c1, c0, c1, a0, a1)
//   P = 2: three accumulators per tile (hh, hl, lh): no dependent pair at all inside a step
//   P = 3: 8 independent accumulators round robin (the issue-rate ceiling)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

// VALU_WAVES: 0 none; 1 fma + exp + add stream (softmax-like); 2 fma-only stream; 3 = 1 with s_setprio 1 on the MFMA waves; 4 = 2 with s_setprio 1;
//             5 / 6: the VALU waves of 1 / 2 ALONE (the MFMA waves exit at once)
template <int P, int VALU_WAVES>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    const int wid = threadIdx.x >> 6;
    f16x8 ah, al, bh, bl;
    for (int e = 0; e < 8; ++e) { ah[e] = (_Float16)(seed + e); al[e] = (_Float16)(seed * 0.5f + e); bh[e] = (_Float16)(seed - e); bl[e] = (_Float16)(seed * 0.25f); }
    f32x16 c[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    if (VALU_WAVES && wid >= 4) {  // the second wave of every SIMD: VALU only
        float x[16];
        for (int r = 0; r < 16; ++r) x[r] = seed + r;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int rep = 0; rep < 6; ++rep)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (VALU_WAVES == 1 || VALU_WAVES == 3 || VALU_WAVES == 5) x[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[r], 0.999f, -0.001f)) + x[(r + 1) & 15];
                    else x[r] = __builtin_fmaf(__builtin_fmaf(x[r], 0.999f, -0.001f), 1.0001f, x[(r + 1) & 15]) * 0.5f;
                }
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += x[r];
        out[blockIdx.x * 512 + threadIdx.x] = s;
        return;
    }
    if (VALU_WAVES >= 5) return;
    if (VALU_WAVES == 3 || VALU_WAVES == 4) __builtin_amdgcn_s_setprio(1);
    for (int it = 0; it < iters; ++it) {
        if (P == 0) {
#pragma unroll
            for (int rep = 0; rep < 8; ++rep) { MFMA(ah, bl, c[1]); MFMA(al, bh, c[1]); MFMA(ah, bh, c[0]); MFMA(ah, bl, c[3]); MFMA(al, bh, c[3]); MFMA(ah, bh, c[2]); }
        } else if (P == 1) {
#pragma unroll
            for (int rep = 0; rep < 8; ++rep) { MFMA(ah, bl, c[1]); MFMA(ah, bl, c[3]); MFMA(al, bh, c[1]); MFMA(al, bh, c[3]); MFMA(ah, bh, c[0]); MFMA(ah, bh, c[2]); }
        } else if (P == 2) {
#pragma unroll
            for (int rep = 0; rep < 8; ++rep) { MFMA(ah, bl, c[1]); MFMA(al, bh, c[4]); MFMA(ah, bh, c[0]); MFMA(ah, bl, c[3]); MFMA(al, bh, c[5]); MFMA(ah, bh, c[2]); }
        } else {
#pragma unroll
            for (int rep = 0; rep < 6; ++rep)
#pragma unroll
                for (int i = 0; i < 8; ++i) MFMA(ah, bh, c[i]);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int P, int VW>
void run(float* out, int threads, const char* what) {
    const int iters = 400, grid = 256;
    hipLaunchKernelGGL((k<P, VW>), dim3(grid), dim3(threads), 0, 0, out, iters, 1.0f);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<P, VW>), dim3(grid), dim3(threads), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int mfma_waves = VW ? 4 : threads / 64;
    const double n_mfma = (double)iters * 48 * mfma_waves * grid;  // per launch
    const double t = ms / 5 * 1e-3;
    printf("pattern %d  %s: %7.1f us  %6.0f TF fp16  (%.1f ns per MFMA and SIMD)\n", P, what, t * 1e6, n_mfma * 32 * 32 * 16 * 2 / t / 1e12,
           t / ((double)iters * 48 * mfma_waves / 4) * 1e9);
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    run<0, 0>(out, 256, "1 wave / SIMD            "); run<1, 0>(out, 256, "1 wave / SIMD            "); run<2, 0>(out, 256, "1 wave / SIMD            "); run<3, 0>(out, 256, "1 wave / SIMD            ");
    run<0, 0>(out, 512, "2 waves / SIMD           "); run<1, 0>(out, 512, "2 waves / SIMD           "); run<2, 0>(out, 512, "2 waves / SIMD           "); run<3, 0>(out, 512, "2 waves / SIMD           ");
    run<0, 1>(out, 512, "MFMA + VALU(exp) wave    "); run<3, 1>(out, 512, "MFMA + VALU(exp) wave    ");
    run<0, 2>(out, 512, "MFMA + VALU(fma) wave    ");
    run<0, 3>(out, 512, "MFMA prio1 + VALU(exp)   "); run<0, 4>(out, 512, "MFMA prio1 + VALU(fma)   ");
    run<0, 5>(out, 512, "VALU(exp) wave alone     "); run<0, 6>(out, 512, "VALU(fma) wave alone     ");
    printf("(VALU stream per iteration: exp form 96 x (fma, exp, add); fma form 96 x (fma, fma, mul); MFMA stream 48 MFMAs per iteration)\n");
    return 0;
}
