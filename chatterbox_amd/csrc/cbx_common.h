// Shared device/host helpers for libcbx_hip.so (gfx950 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/cbx.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Stores of the decode step's outputs (activation images, split-K partial images, the KV append).  A dependent launch starts only after its predecessor's end-of-kernel
// release has written the dirty lines of every XCD's L2 back; a line stored write-through has left the L2 by then (MI355X_MICROARCH.md "publish-large": 64 KB per
// workgroup, plain stores + release 8.2 us against sc1 stores 3.0).  Same values at the same addresses: results are unchanged bit for bit.
// CBX_WT: 1 sc1 (agent-scope write-through; the default), 0 plain stores, 2 sc0 sc1 (system scope), 3 nt -- side builds: scripts/wt_build.sh.  Measured on one box, interleaved
// processes (profiles/r06_az_decode_store_policy_ab.log): the B = 8 token step 1.1285 ms plain, 1.1050 sc1, 1.1043 sc0 sc1, 1.1398 nt; identical logits.  The flow's kernels
// (16 MB outputs per launch: plane GEMM epilogues, plane attention, narrow LayerNorm) LOSE 3.3 % with write-through stores (profiles/r06_ba_flow_store_policy_ab.log) and keep
// plain stores.
#ifndef CBX_WT
#define CBX_WT 1
#endif
__device__ __forceinline__ void cbx_store_out(float* p, float v) {
#if CBX_WT == 1
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#elif CBX_WT == 2
    asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
#elif CBX_WT == 3
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ void cbx_store_out4(float* p, f32x4 v) {
#if CBX_WT == 1
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");  // the nop: a VALU write of the data registers of a > 64-bit store needs wait states the compiler does not insert after inline asm
#elif CBX_WT == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#elif CBX_WT == 3
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
    *reinterpret_cast<f32x4*>(p) = v;
#endif
}
#define CBX_F16_LO_SCALE 2048.0f  /* scale of the second fp16 plane of the f16x3 forms (gemm_split.hip) */
int* cbx_range_flag();            /* gemm_split.hip: device word of cbx_set_range_flag, or NULL */
#ifdef __HIPCC__
// amax = max(amax, |v0..3|) in two instructions (fmaxf chains compile to one canonicalising v_max per operand)
__device__ __forceinline__ void cbx_amax4(float& amax, const f32x4 v) {
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(v[0]), "v"(v[1]));
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(v[2]), "v"(v[3]));
}
// Plane pair of two fp32 values (gemm_planes.hip): h = {RNE16(v0), RNE16(v1)}, l = {RNE16(2048 (v0 - h0)), RNE16(2048 (v1 - h1))}, both packed
// lo | hi << 16.  The residual comes out of ONE mixed-precision fma per element (v_fma_mix{lo,hi}_f16: fp16 operand h, fp32 operands -2048 and
// 2048 v; the fma is exact, so its fp16 rounding is the only one) instead of cvt + fma + cvt.  The trailing s_nop covers the
// VALU-write -> MFMA-read wait states hipcc does not pad after an asm statement (the planes often feed an MFMA directly).
__device__ __forceinline__ void cbx_split2(float v0, float v1, unsigned& h, unsigned& l) {
    typedef _Float16 cbx_h2 __attribute__((ext_vector_type(2)));
    typedef float cbx_f2 __attribute__((ext_vector_type(2)));
    const cbx_f2 v = {v0, v1};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, cbx_h2));
    const cbx_f2 t = v * CBX_F16_LO_SCALE;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "s"(-CBX_F16_LO_SCALE), "v"(t[0]));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 1" : "+v"(l) : "v"(h), "s"(-CBX_F16_LO_SCALE), "v"(t[1]));
}
#endif

// ordinal of the calling thread's current device, clamped to [0, 64): index of the per-device tables (range flag, "LDS opted into" bits)
static inline int cbx_device() {
    int d = 0;
    return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64) ? d : 0;
}

// ---- CO-RESIDENT streams (ABI v13, cbx_set_stream_coresident): a stream ATTRIBUTE, like its priority.  Launches on such a stream that would otherwise fill a CU
// with several workgroups of one kernel (LayerNorm: 5 per CU; the split GEMM: 2) reserve enough dynamic LDS that at most `max_wg` fit, so that half of every
// SIMD's register file stays free for the workgroups of another stream's latency-bound kernel chain (engine.synthesize_pipelined; profiles/r05_overlap_*).
// Returns the dynamic LDS bytes to request for a kernel that itself needs `own` bytes per workgroup.
int cbx_stream_coresident(hipStream_t st);
static inline size_t cbx_coresident_lds(hipStream_t st, size_t own, int max_wg) {
    if (!cbx_stream_coresident(st)) return own;
    const size_t want = (size_t)(160 * 1024) / (size_t)(max_wg + 1) + 1024;  // more than a (max_wg + 1)-th of the CU's 160 KiB
    return own > want ? own : want;
}

// ---- CBX_TRACE: launch-timeline instrumentation of a SIDE build (scripts/trace_decode.sh -> build/libcbx_hip_trace.so; the product library is
// compiled without it and its instruction streams do not change).  Thread 0 of every workgroup of an instrumented kernel reads the chip-wide
// 100 MHz counter (s_memrealtime) at up to 7 points and appends ONE 64-byte record {tag, t0 .. t6} to a device log at exit; the host sorts the
// records by time.  A poor man's thread trace: where a dependent chain of 5-9 us kernels spends its time (dispatch ramp, first bytes, stream,
// reduce, epilogue, boundary to the next launch).
#if defined(CBX_TRACE) && defined(__HIPCC__)
#define CBX_TRC_TU                                                                                                                          \
    __device__ unsigned long long* g_trc_buf = nullptr;                                                                                    \
    __device__ unsigned* g_trc_cnt = nullptr;                                                                                              \
    __device__ unsigned g_trc_cap = 0;
#define CBX_TRC_SETTER(name)                                                                                                                \
    extern "C" int name(unsigned long long* buf, unsigned* cnt, unsigned cap) {                                                             \
        hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_trc_buf), &buf, sizeof(buf));                                                         \
        if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_trc_cnt), &cnt, sizeof(cnt));                                               \
        if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_trc_cap), &cap, sizeof(cap));                                               \
        return (int)e;                                                                                                                      \
    }
#define CBX_TRC_DECL unsigned long long trc_t[7] = {0, 0, 0, 0, 0, 0, 0}
#define CBX_TRC_STAMP(k)                                                  \
    do {                                                                  \
        if (threadIdx.x == 0) trc_t[k] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define CBX_TRC_FLUSH(tag)                                                                                                   \
    do {                                                                                                                     \
        if (threadIdx.x == 0 && g_trc_buf) {                                                                                 \
            const unsigned i = atomicAdd(g_trc_cnt, 1u);                                                                     \
            if (i < g_trc_cap) {                                                                                             \
                unsigned long long* r = g_trc_buf + (unsigned long long)i * 8;                                               \
                r[0] = (unsigned long long)(tag) << 32 | (unsigned long long)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)); \
                for (int k_ = 0; k_ < 7; ++k_) r[1 + k_] = trc_t[k_];                                                        \
            }                                                                                                                \
        }                                                                                                                    \
    } while (0)
#else
#define CBX_TRC_TU
#define CBX_TRC_SETTER(name)
#define CBX_TRC_DECL
#define CBX_TRC_STAMP(k)
#define CBX_TRC_FLUSH(tag)
#endif

extern thread_local char cbx_err_buf[512];
int cbx_set_error(int code, const char* fmt, ...);
int cbx_check_launch(const char* what);

#define CBX_REQUIRE(cond, ...)                                     \
    do {                                                           \
        if (!(cond)) return cbx_set_error(CBX_EINVAL, __VA_ARGS__); \
    } while (0)

// GELU (erf form: F.gelu default, diffusers GELU of the CFM feed-forward): 0.5 v (1 + erf(v / sqrt 2)), branch-free.
//   |x| < 1:  erf(x) = x + x q(x^2)                      (degree-6 fit, relative weight)
//   |x| >= 1: erf(x) = 1 - 2^(-log2(e) r),  r = a + a s(a) = -ln erfc(a), a = min(|x|, 4)  (degree-6 fit weighted by erfc(a) a; erf(4) = 1 in fp32)
// Max |erf error| 7.4e-8 (1.5 ulp) and max |GELU error| 5.9e-8 against fp64 over [-6, 6] and N(0, 2) samples (the fit and the check:
// DESIGN.md section 3); ~26 VALU + 1 v_exp instead of the ~50 of the library erff (both of its branches run in a mixed wave).
__device__ __forceinline__ float cbx_gelu_erf(float v) {
    const float x = v * 0.70710678118654752f;
    const float a = fminf(fabsf(x), 4.0f), t = x * x;
    float q = 7.847259257687256e-05f;
    q = __builtin_fmaf(q, t, -0.0008008189033716917f);
    q = __builtin_fmaf(q, t, 0.005188099108636379f);
    q = __builtin_fmaf(q, t, -0.026853691786527634f);
    q = __builtin_fmaf(q, t, 0.1128358244895935f);
    q = __builtin_fmaf(q, t, -0.3761262595653534f);
    q = __builtin_fmaf(q, t, 0.12837916612625122f);
    const float e_small = __builtin_fmaf(a, q, a);
    float sp = 1.4670923519588541e-05f;
    sp = __builtin_fmaf(sp, a, -0.00035766823566518724f);
    sp = __builtin_fmaf(sp, a, 0.0037862290628254414f);
    sp = __builtin_fmaf(sp, a, -0.024070942774415016f);
    sp = __builtin_fmaf(sp, a, 0.10660174489021301f);
    sp = __builtin_fmaf(sp, a, 0.634926438331604f);
    sp = __builtin_fmaf(sp, a, 0.12870502471923828f);
    const float r = __builtin_fmaf(a, sp, a);
    const float e_big = 1.0f - __builtin_amdgcn_exp2f(r * -1.4426950408889634f);
    const float e = copysignf(a < 1.0f ? e_small : e_big, x);
    return 0.5f * v * (1.0f + e);
}

__device__ __forceinline__ float cbx_act(float v, int act, float slope, float param) {
    switch (act) {
        case CBX_ACT_SILU: return v / (1.0f + __expf(-v));
        case CBX_ACT_GELU_ERF: return cbx_gelu_erf(v);
        case CBX_ACT_GELU_TANH: {
            float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
            return 0.5f * v * (1.0f + tanhf(u));
        }
        case CBX_ACT_MISH: {
            // x * tanh(softplus(x)) = x * n / (n + 2), n = e^x (e^x + 2): one exp and one reciprocal instead of exp + log1p + tanh
            // (max rel. error 2.8e-7 against fp64 over [-30, 30], the same as torch's fp32 mish; x > 20: tanh(softplus) = 1 in fp32)
            if (v > 20.0f) return v;
            const float e = expf(v), n = e * (e + 2.0f);
            return v * (n / (n + 2.0f));
        }
        case CBX_ACT_LRELU: return v > 0.0f ? v : v * slope;
        case CBX_ACT_ELU: return v > 0.0f ? v : expm1f(v);
        case CBX_ACT_TANH: return tanhf(v);
        case CBX_ACT_SNAKE: {
            float s = sinf(v * param);
            return v + (1.0f / (param + 1e-9f)) * s * s;
        }
        case CBX_ACT_ABS: return fabsf(v);
        default: return v;
    }
}

// x of lane (l ^ MASK), MASK a power of two < 64, WITHOUT the LDS crossbar: hipcc lowers __shfl_xor to ds_bpermute_b32 (an LDS round trip, ~100
// cycles each; a 6-step butterfly is a dependent chain of six) -- here 1 / 2 are quad_perm DPP moves, 4 is a row_shl:4 / row_shr:4 pair under
// bank masks, 8 is row_ror:8 (a rotation by half a 16-lane row IS the xor), 16 / 32 are v_permlane16_swap / v_permlane32_swap (gfx950).  Same
// partner lane as __shfl_xor, so every reduction built on it keeps its results bit for bit (round 6).  The two operands of a swap must be
// DIFFERENT registers (hipcc folds swap(x, x) into "both results equal"): the copy is made opaque.
template <int MASK>
__device__ __forceinline__ float cbx_xor_lane(float x) {
    static_assert(MASK == 1 || MASK == 2 || MASK == 4 || MASK == 8 || MASK == 16 || MASK == 32, "xor partner: a power of two below 64");
    const int v = __builtin_bit_cast(int, x);
    if constexpr (MASK == 1) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    else if constexpr (MASK == 2) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    else if constexpr (MASK == 4) {
        int r = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0x5, false);  // row_shl:4 -> banks 0, 2 (lanes 0-3, 8-11 of a row) take lane l + 4
        r = __builtin_amdgcn_update_dpp(r, v, 0x114, 0xF, 0xA, false);      // row_shr:4 -> banks 1, 3 take lane l - 4
        return __builtin_bit_cast(float, r);
    } else if constexpr (MASK == 8) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true));  // row_ror:8
    else {
        unsigned a = (unsigned)v, b = a;
        asm volatile("" : "+v"(b));
        if constexpr (MASK == 16) {
            const auto sw = __builtin_amdgcn_permlane16_swap(a, b, false, false);  // odd rows of a <-> even rows of b
            return __builtin_bit_cast(float, (threadIdx.x & 16) ? sw[0] : sw[1]);
        } else {
            const auto sw = __builtin_amdgcn_permlane32_swap(a, b, false, false);  // lanes 32-63 of a <-> lanes 0-31 of b
            return __builtin_bit_cast(float, (threadIdx.x & 32) ? sw[0] : sw[1]);
        }
    }
}
__device__ __forceinline__ float wave_sum(float v) {  // the xor butterfly 32, 16, .. 1 (order and partners of rounds 1-5: same bits)
    v += cbx_xor_lane<32>(v);
    v += cbx_xor_lane<16>(v);
    v += cbx_xor_lane<8>(v);
    v += cbx_xor_lane<4>(v);
    v += cbx_xor_lane<2>(v);
    v += cbx_xor_lane<1>(v);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, cbx_xor_lane<32>(v));
    v = fmaxf(v, cbx_xor_lane<16>(v));
    v = fmaxf(v, cbx_xor_lane<8>(v));
    v = fmaxf(v, cbx_xor_lane<4>(v));
    v = fmaxf(v, cbx_xor_lane<2>(v));
    v = fmaxf(v, cbx_xor_lane<1>(v));
    return v;
}

// XCD-aware workgroup remap (MI355X: 8 XCDs with private L2s; workgroup b is dispatched to XCD b % 8).  Returns the
// logical tile id for this workgroup such that the workgroups resident on one XCD own CONSECUTIVE tile ids, so tiles
// that share an operand panel (same A rows / same K,V head) hit that XCD's L2 instead of each XCD re-fetching the
// panel.  Bijective for any grid size; placement only affects speed, never results.
__device__ __forceinline__ int cbx_xcd_remap(int orig, int nwg) {
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}
