// Does VALU work issued BY THE SAME WAVE between its MFMAs hide under them (v_mfma_f32_32x32x16_f16 = 32 cycles of matrix pipe, 4 of issue)?
// One wave streams 48 MFMAs and 48 * K softmax-like VALU ops per iteration either as two blocks (all MFMAs, then all VALU: the structure of
// flash_attn_pl2_kernel's matrix / vector blocks) or interleaved 1 MFMA : K VALU.  Registers only, no memory, 1 or 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_valu_inwave.hip -o /tmp/mv && /tmp/mv
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

// one softmax-like group on 2 scores: 2 fma, 2 exp, 2 add (row sum), 1 cvt_pk, 2 fma_mix = 9 VALU (2 of them transcendental)
__device__ __forceinline__ void vgroup(float& x0, float& x1, float& ls, unsigned& hp, unsigned& lp, float sc, float sh) {
    const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(x0, sc, sh));
    const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(x1, sc, sh));
    ls += p0 + p1;
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 h2t __attribute__((ext_vector_type(2)));
    const unsigned h2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{p0, p1}, h2t));
    unsigned l2;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l2) : "v"(h2), "v"(p0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l2) : "v"(h2), "v"(p1));
    hp ^= h2, lp ^= l2;  // +2 VALU: 11 per group
    x0 = p0 * 0.5f, x1 = p1 * 0.5f;  // +2: 13 per group (keeps the chain alive)
}

// MODE 0: MFMA only; 1: VALU only; 2: blocks (48 MFMA, then G groups); 3: interleaved (1 MFMA, then G/48 groups... see below)
template <int MODE, int GPM2>  // GPM2 = VALU groups per 2 MFMAs (1 -> 6.5 VALU per MFMA, 2 -> 13 per MFMA)
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    f16x8 ah, bh;
    for (int e = 0; e < 8; ++e) { ah[e] = (_Float16)(seed + e); bh[e] = (_Float16)(seed - e); }
    f32x16 c[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    float x[16], ls = 0.f;
    unsigned hp = 0, lp = 0;
    for (int r = 0; r < 16; ++r) x[r] = seed * 0.01f * r;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2 || MODE == 0 || MODE == 1) {
            if (MODE != 1) {
#pragma unroll
                for (int i = 0; i < 48; ++i) MFMA(ah, bh, c[i & 7]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE != 0) {
#pragma unroll
                for (int g = 0; g < 24 * GPM2; ++g) vgroup(x[(2 * g) & 15], x[(2 * g + 1) & 15], ls, hp, lp, 0.999f, -0.01f);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                MFMA(ah, bh, c[(2 * i) & 7]);
                __builtin_amdgcn_sched_barrier(0);
                vgroup(x[(2 * i) & 15], x[(2 * i + 1) & 15], ls, hp, lp, 0.999f, -0.01f);
                __builtin_amdgcn_sched_barrier(0);
                MFMA(ah, bh, c[(2 * i + 1) & 7]);
                __builtin_amdgcn_sched_barrier(0);
                if (GPM2 == 2) vgroup(x[(2 * i + 8) & 15], x[(2 * i + 9) & 15], ls, hp, lp, 0.999f, -0.01f);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = ls + __uint_as_float(hp & 0x3fffffff) + __uint_as_float(lp & 0x3fffffff);
    for (int r = 0; r < 16; ++r) s += x[r];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int GPM2>
void run(float* out, int threads, const char* what) {
    const int iters = 400, grid = 256;
    hipLaunchKernelGGL((k<MODE, GPM2>), dim3(grid), dim3(threads), 0, 0, out, iters, 1.0f);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE, GPM2>), dim3(grid), dim3(threads), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %d waves/SIMD: %7.1f us\n", what, threads / 256, ms / 5 * 1e3);
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    for (int th = 256; th <= 512; th += 256) {
        run<0, 1>(out, th, "48 MFMA per iteration, nothing else");
        run<1, 1>(out, th, "24 VALU groups (312 VALU, 48 of them exp) per iteration alone");
        run<2, 1>(out, th, "blocks: 48 MFMA, then 24 groups");
        run<3, 1>(out, th, "interleaved: (MFMA, group, MFMA) x 24");
        run<1, 2>(out, th, "48 VALU groups (624 VALU, 96 exp) alone");
        run<2, 2>(out, th, "blocks: 48 MFMA, then 48 groups");
        run<3, 2>(out, th, "interleaved: (MFMA, group) x 48");
    }
    return 0;
}
