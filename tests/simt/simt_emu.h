// SIMT emulator for the gfx950 kernels of chatterbox_amd/csrc -- TEST INFRASTRUCTURE, not a product path.
//
// The .hip sources are compiled AS THEY ARE (tests/simt/build_emu.py only rewrites the handful of inline-asm statements and the
// `extern __shared__` declarations) for the x86 host against this header, which stands in for <hip/hip_runtime.h>:
//   * a workgroup runs as blockDim.x cooperative fibers on ONE OS thread (simt_core.cpp: hand-written context switch, round-robin
//     scheduler, deadlock detection); `__shared__` is a function-level static, so workgroups run one after the other;
//   * __syncthreads / s_barrier, the wave-level exchanges (__shfl_*, DPP, permlane, ballot) and the MFMA instructions are rendezvous
//     points: every lane deposits its operands, waits for its wave, then computes ITS OWN destination registers from the deposited
//     operands with the register layout of the gfx950 instruction;
//   * buffer resources carry {base, num_records}; out-of-range loads return zeros and out-of-range stores are dropped, as the hardware's
//     bounds check does; LDS-DMA (`buffer_load ... lds`, `global_load_lds`) copies synchronously -- one legal completion order;
//   * s_waitcnt / s_nop / sched_barrier / s_setprio do nothing.
// What this can and cannot show: index arithmetic, layouts, barriers, reductions and the arithmetic itself (to fp32 rounding) are
// exercised by the same source the GPU runs; timing, memory-model visibility and missing-wait bugs are not.
// The library built this way (tests/simt/libcbx_emu.so) exports the SAME C ABI (include/cbx.h) on host pointers.  Only tests load it.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>

#define CBX_SIMT_EMU 1
#ifndef __HIPCC__
#define __HIPCC__ 1
#endif
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __constant__

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { unsigned x, y, z, w; } __attribute__((aligned(16)));
struct int4 { int x, y, z, w; } __attribute__((aligned(16)));
struct float4 { float x, y, z, w; } __attribute__((aligned(16)));
struct float2 { float x, y; } __attribute__((aligned(8)));
struct int2 { int x, y; } __attribute__((aligned(8)));
struct uint2 { unsigned x, y; } __attribute__((aligned(8)));
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }

// ---------------------------------------------------------------------------------------------------------------- runtime stubs
typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorLaunchFailure = 719 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { int multiProcessorCount = 256; size_t sharedMemPerBlock = 160 * 1024; int warpSize = 64; };

namespace simt {
constexpr int WAVE = 64;
constexpr int XSLOT = 128;  // bytes per lane of a wave's exchange buffer
struct Lane {
    dim3 tid;
    int flat, lane, wave;
    int xpar;               // parity of the exchange buffer this lane uses for its next collective
    int wave_lanes;         // lanes of this lane's wave (64 except in a ragged last wave)
    unsigned char* wave_x;  // this wave's exchange buffers [2][WAVE][XSLOT]
};
extern Lane* g_cur;
extern dim3 g_blockIdx, g_blockDim, g_gridDim;
extern int g_last_error;

int launch(dim3 grid, dim3 block, size_t dyn_lds, const std::function<void()>& body);
void* dyn_lds();
void block_sync();
void wave_sync();
void yield();
void note_progress();
// this lane's slot / another lane's slot of the wave's exchange buffer for the collective in flight
inline unsigned char* xslot_of(int lane) { return g_cur->wave_x + (g_cur->xpar * WAVE + lane) * XSLOT; }
inline unsigned char* xslot_mine() { return xslot_of(g_cur->lane); }
inline void xflip() { g_cur->xpar ^= 1; }
inline int wave_lanes() { return g_cur->wave_lanes; }
// Vector-memory completion order (CBX_EMU_DMA=deferred): an LDS-DMA does not land when it is issued but when its lane waits for it --
// `s_waitcnt vmcnt(n)` (the explicit asm waits of the sources), `__syncthreads()` (which carries a vmcnt(0)) or the end of the kernel --
// i.e. as LATE as the kernel's own waits allow.  Buffer loads / stores through the builtins occupy a slot of the same in-order queue.
// A kernel whose counted waits are too weak then reads stale LDS deterministically, instead of once in a while on the hardware.
extern bool g_dma_deferred;
void vm_push_dma(void* dst, const void* src, int size);  // src == nullptr: zeros (out-of-range lanes)
void vm_push_other();
void vm_wait(int n);
// pairwise mailbox of the power-of-two xor shuffles (lane groups of one wave may have diverged: decode attention)
unsigned long long shfl_xor_pair(unsigned long long bits, int mask_log2);
}  // namespace simt

#define threadIdx (simt::g_cur->tid)
#define blockIdx (simt::g_blockIdx)
#define blockDim (simt::g_blockDim)
#define gridDim (simt::g_gridDim)

static inline hipError_t hipGetLastError() {
    const int e = simt::g_last_error;
    simt::g_last_error = 0;
    return e;
}
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "simt emulator: launch failed (deadlock or fault, see stderr)"; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { *p = hipDeviceProp_t(); return hipSuccess; }
template <class F> static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
template <class F> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t) { *n = 1; return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    ((void)(stream), simt::launch(dim3(grid), dim3(block), (size_t)(shmem), [=]() { kern(__VA_ARGS__); }))

static inline void __syncthreads() {
    if (simt::g_dma_deferred) simt::vm_wait(0);  // the compiler's __syncthreads drains the wave's vector-memory counter
    simt::block_sync();
}

// ---------------------------------------------------------------------------------------------------------------- scalar helpers
#define __expf(x) expf(x)
#define __logf(x) logf(x)
#define __log2f(x) log2f(x)
#define __exp2f(x) exp2f(x)
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.0f / a; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __frsqrt_rn(float x) { return 1.0f / sqrtf(x); }
static inline float __fsqrt_rn(float x) { return sqrtf(x); }
#define __sinf(x) sinf(x)
#define __cosf(x) cosf(x)
static inline unsigned __float_as_uint(float f) { return __builtin_bit_cast(unsigned, f); }
static inline int __float_as_int(float f) { return __builtin_bit_cast(int, f); }
static inline float __uint_as_float(unsigned u) { return __builtin_bit_cast(float, u); }
static inline float __int_as_float(int u) { return __builtin_bit_cast(float, u); }
static inline long long __double_as_longlong(double d) { return __builtin_bit_cast(long long, d); }
static inline double __longlong_as_double(long long d) { return __builtin_bit_cast(double, d); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline float __saturatef(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
using std::max;
using std::min;
static inline long min(long a, int b) { return a < b ? a : b; }
static inline long min(int a, long b) { return a < b ? a : b; }
static inline long max(long a, int b) { return a > b ? a : b; }
static inline long max(int a, long b) { return a > b ? a : b; }

// ---------------------------------------------------------------------------------------------------------------- atomics
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add(p, v, order)
#define __hip_atomic_fetch_or(p, v, order, scope) __atomic_fetch_or(p, v, order)
#define __hip_atomic_fetch_max(p, v, order, scope) simt_atomic_max(p, v)
#define __hip_atomic_load(p, order, scope) simt_atomic_load(p)
#define __hip_atomic_store(p, v, order, scope) simt_atomic_store(p, v)
template <class T> static inline T simt_atomic_load(const T* p) { return *(const volatile T*)p; }
template <class T, class U> static inline void simt_atomic_store(T* p, U v) { *(volatile T*)p = (T)v; }
template <class T> static inline T simt_atomic_max(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicOr(int* p, int v) { int o = *p; *p = o | v; return o; }
static inline unsigned atomicOr(unsigned* p, unsigned v) { unsigned o = *p; *p = o | v; return o; }
static inline int atomicMax(int* p, int v) { int o = *p; if (v > o) *p = v; return o; }
static inline int atomicMin(int* p, int v) { int o = *p; if (v < o) *p = v; return o; }
static inline int atomicExch(int* p, int v) { int o = *p; *p = v; return o; }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

// ---------------------------------------------------------------------------------------------------------------- wave exchanges


template <class T> static inline T __shfl_xor(T v, int mask, int /*width*/ = 64) {
    static_assert(sizeof(T) <= 8, "shuffle of at most 8 bytes");
    unsigned long long bits = 0;
    memcpy(&bits, &v, sizeof(T));
    if (mask > 0 && (mask & (mask - 1)) == 0 && mask < 64) {
        bits = simt::shfl_xor_pair(bits, __builtin_ctz((unsigned)mask));
    } else {
        memcpy(simt::xslot_mine(), &bits, 8);
        simt::wave_sync();
        const int src = simt::g_cur->lane ^ mask;
        if (src < simt::wave_lanes()) memcpy(&bits, simt::xslot_of(src), 8);
        simt::xflip();
    }
    T r;
    memcpy(&r, &bits, sizeof(T));
    return r;
}
template <class T, class F> static inline T simt_shfl_by(T v, F src_of) {
    static_assert(sizeof(T) <= 8, "shuffle of at most 8 bytes");
    unsigned long long bits = 0;
    memcpy(&bits, &v, sizeof(T));
    memcpy(simt::xslot_mine(), &bits, 8);
    simt::wave_sync();
    const int src = src_of(simt::g_cur->lane);
    if (src >= 0 && src < simt::wave_lanes()) memcpy(&bits, simt::xslot_of(src), 8);
    simt::xflip();
    T r;
    memcpy(&r, &bits, sizeof(T));
    return r;
}
template <class T> static inline T __shfl(T v, int src, int width = 64) {
    return simt_shfl_by(v, [=](int l) { return (l & ~(width - 1)) | (src & (width - 1)); });
}
template <class T> static inline T __shfl_up(T v, unsigned delta, int width = 64) {
    return simt_shfl_by(v, [=](int l) { return (l & (width - 1)) >= (int)delta ? l - (int)delta : -1; });
}
template <class T> static inline T __shfl_down(T v, unsigned delta, int width = 64) {
    return simt_shfl_by(v, [=](int l) { return (l & (width - 1)) + (int)delta < width ? l + (int)delta : -1; });
}
static inline unsigned long long simt_ballot(bool pred) {
    const unsigned char b = pred ? 1 : 0;
    memcpy(simt::xslot_mine(), &b, 1);
    simt::wave_sync();
    unsigned long long m = 0;
    for (int l = 0; l < simt::wave_lanes(); ++l)
        if (*simt::xslot_of(l)) m |= 1ull << l;
    simt::xflip();
    return m;
}
static inline unsigned long long __ballot(int pred) { return simt_ballot(pred != 0); }
static inline int __all(int pred) { return simt_ballot(pred == 0) == 0; }
static inline int __any(int pred) { return simt_ballot(pred != 0) != 0; }
#define __builtin_amdgcn_ballot_w64(...) simt_ballot(__VA_ARGS__)
#define __builtin_amdgcn_readfirstlane(x) (x) /* the sources only use it on wave-uniform values */

// v_mov_b32 with a DPP control: quad_perm (ctrl < 0x100), row_shl:n (0x100 + n: lane l of a 16-lane row receives from l + n), row_shr:n (0x110 + n: from
// l - n), row_ror:n (0x120 + n: from (l - n) mod 16); row_mask / bank_mask select the rows (16 lanes) / banks (4 lanes of a row) that are WRITTEN, the
// others keep `old`; a written lane without a source (shifted in from outside the row) gets 0 with bound_ctrl, `old` without.  Anything else fails loudly.
static inline int simt_update_dpp(int old, int v, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const int l = simt::g_cur->lane;
    int src = -1;
    if (ctrl >= 0 && ctrl <= 0xFF) src = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
    else if (ctrl >= 0x101 && ctrl <= 0x10F) src = (l & 15) + (ctrl & 0xF) < 16 ? l + (ctrl & 0xF) : -1;
    else if (ctrl >= 0x111 && ctrl <= 0x11F) src = (l & 15) - (ctrl & 0xF) >= 0 ? l - (ctrl & 0xF) : -1;
    else if (ctrl >= 0x121 && ctrl <= 0x12F) src = (l & ~15) | (((l & 15) - (ctrl & 0xF)) & 15);
    else {
        fprintf(stderr, "simt: unsupported DPP control 0x%x\n", ctrl);
        abort();
    }
    // the xor-partner forms (cbx_xor_lane: quad_perm xor 1 / 2, the row_shl:4 | row_shr:4 pair under bank masks, row_ror:8) synchronise PAIRWISE like
    // __shfl_xor does here -- 16-lane groups of a wave may sit in different loop iterations (decode attention) --, everything else wave-wide
    int xb = -1;
    if (ctrl == 0xB1) xb = 0;
    else if (ctrl == 0x4E) xb = 1;
    else if ((ctrl == 0x104 && bank_mask == 0x5) || (ctrl == 0x114 && bank_mask == 0xA)) xb = 2;
    else if (ctrl == 0x128) xb = 3;
    int got;
    if (xb >= 0) {
        got = (int)(unsigned)simt::shfl_xor_pair((unsigned long long)(unsigned)v, xb);
        if (src != (l ^ (1 << xb))) src = -2;  // (a lane the bank mask does not write: whatever it received is dropped below)
    } else {
        got = simt_shfl_by(v, [=](int) { return src; });  // (every lane takes part in the exchange)
    }
    const bool enabled = ((row_mask >> (l >> 4)) & 1) && ((bank_mask >> ((l >> 2) & 3)) & 1);
    if (!enabled) return old;
    if (src < 0) {
        if (src == -2) {
            fprintf(stderr, "simt: DPP xor form writes a lane whose source is not its xor partner (ctrl 0x%x)\n", ctrl);
            abort();
        }
        return bound_ctrl ? 0 : old;
    }
    return got;
}
#define __builtin_amdgcn_mov_dpp(v, ctrl, rm, bm, bc) simt_update_dpp(0, v, ctrl, rm, bm, bc)
#define __builtin_amdgcn_update_dpp(old, v, ctrl, rm, bm, bc) simt_update_dpp(old, v, ctrl, rm, bm, bc)

typedef unsigned simt_u32x2 __attribute__((ext_vector_type(2)));
// v_permlane32_swap vdst, src: lanes 32-63 of vdst <-> lanes 0-31 of src.  Returns {new vdst, new src}.
static inline simt_u32x2 simt_permlane_swap(unsigned vdst, unsigned src, int bit) {
    const unsigned long long mine = (unsigned long long)vdst | ((unsigned long long)src << 32);
    const unsigned long long other = simt::shfl_xor_pair(mine, bit);  // pairwise: lane l <-> lane l ^ (1 << bit)
    const int l = simt::g_cur->lane;
    simt_u32x2 r;
    if (!(l & (1 << bit))) {  // lower half / even row: my src <- vdst of the partner; my vdst unchanged
        r[0] = vdst;
        r[1] = (unsigned)other;
    } else {  // upper half / odd row: my vdst <- src of the partner; my src unchanged
        r[0] = (unsigned)(other >> 32);
        r[1] = src;
    }
    return r;
}
static inline simt_u32x2 simt_permlane32_swap(unsigned vdst, unsigned src, bool, bool) { return simt_permlane_swap(vdst, src, 5); }
#define __builtin_amdgcn_permlane32_swap(...) simt_permlane32_swap(__VA_ARGS__)
// v_permlane16_swap vdst, src: odd rows (lanes 16-31, 48-63) of vdst <-> even rows (lanes 0-15, 32-47) of src.  Returns {new vdst, new src}.
static inline simt_u32x2 simt_permlane16_swap(unsigned vdst, unsigned src, bool, bool) { return simt_permlane_swap(vdst, src, 4); }
#define __builtin_amdgcn_permlane16_swap(...) simt_permlane16_swap(__VA_ARGS__)

// ---------------------------------------------------------------------------------------------------------------- no-op instructions
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) (simt::note_progress(), simt::yield()) /* a polling lane is not stuck: bounded spins end by themselves */
#define __builtin_amdgcn_s_nop(x) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_s_barrier() simt::block_sync()
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define __builtin_amdgcn_exp2f(...) exp2f(__VA_ARGS__)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_rsqf(x) (1.0f / sqrtf(x))
#define __builtin_amdgcn_fmed3f(a, b, c) fmaxf(fminf(fmaxf(a, b), c), fminf(a, b))

// ---------------------------------------------------------------------------------------------------------------- MFMA
typedef float simt_f32x4 __attribute__((ext_vector_type(4)));
typedef float simt_f32x16 __attribute__((ext_vector_type(16)));

template <class E> static inline float simt_to_f32(E e) {
    if constexpr (sizeof(E) == 4) return (float)e;
    else if constexpr (__is_same(E, _Float16)) return (float)e;
    else {  // bf16 carried as a 16-bit integer or __bf16
        unsigned short b;
        memcpy(&b, &e, 2);
        return __builtin_bit_cast(float, (unsigned)b << 16);
    }
}

// The wave's LAST arriving lane computes the whole D tile from the deposited operands (one conversion per operand element, plain loops the
// host compiler vectorises) into the wave's result area; every lane then adds its own registers' share to its C operand.
namespace simt {
bool wave_arrive_last();   // true for the lane that completes the wave's rendezvous (it must call wave_release() when the result is ready)
void wave_release();
void wave_wait_released();
float* wave_result();      // [32 * 32] floats per wave
}
// v_mfma_f32_16x16x4_f32: A[i][k] in lane 16k + i, B[k][j] in lane 16k + j, D[4(l / 16) + r][l % 16] in register r of lane l.
// Each D element is the fmaf chain  fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, c))))  -- what the fp32 MFMA computes bit for bit.
static inline simt_f32x4 simt_mfma_f32_16x16x4f32(float a, float b, simt_f32x4 c, int, int, int) {
    unsigned char* m = simt::xslot_mine();
    const float ab[2] = {a, b};
    memcpy(m, ab, 8);
    memcpy(m + 16, &c, 16);
    if (simt::wave_arrive_last()) {
        float A[16][4], B[16][4];
        for (int ln = 0; ln < 64; ++ln) {
            float v[2];
            memcpy(v, simt::xslot_of(ln), 8);
            A[ln & 15][ln >> 4] = v[0];
            B[ln & 15][ln >> 4] = v[1];
        }
        float* R = simt::wave_result();
        for (int ln = 0; ln < 64; ++ln) {
            float cc[4];
            memcpy(cc, simt::xslot_of(ln) + 16, 16);
            const int j = ln & 15;
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * (ln >> 4) + r;
                float s = cc[r];
                for (int k = 0; k < 4; ++k) s = fmaf(A[i][k], B[j][k], s);
                R[ln * 4 + r] = s;
            }
        }
        simt::wave_release();
    } else {
        simt::wave_wait_released();
    }
    simt_f32x4 d;
    memcpy(&d, simt::wave_result() + simt::g_cur->lane * 4, 16);
    simt::xflip();
    return d;
}
// 32x32 accumulator layout: register r of lane l holds D[8 (r / 4) + 4 (l / 32) + r % 4][l % 32].
static inline simt_f32x16 simt_mfma_f32_32x32x2f32(float a, float b, simt_f32x16 c, int, int, int) {
    unsigned char* m = simt::xslot_mine();
    const float ab[2] = {a, b};
    memcpy(m, ab, 8);
    memcpy(m + 16, &c, 64);
    if (simt::wave_arrive_last()) {
        float A[32][2], B[32][2];
        for (int ln = 0; ln < 64; ++ln) {
            float v[2];
            memcpy(v, simt::xslot_of(ln), 8);
            A[ln & 31][ln >> 5] = v[0];
            B[ln & 31][ln >> 5] = v[1];
        }
        float* R = simt::wave_result();
        for (int ln = 0; ln < 64; ++ln) {
            float cc[16];
            memcpy(cc, simt::xslot_of(ln) + 16, 64);
            const int j = ln & 31;
            for (int r = 0; r < 16; ++r) {
                const int i = 8 * (r >> 2) + 4 * (ln >> 5) + (r & 3);
                R[ln * 16 + r] = fmaf(A[i][1], B[j][1], fmaf(A[i][0], B[j][0], cc[r]));
            }
        }
        simt::wave_release();
    } else {
        simt::wave_wait_released();
    }
    simt_f32x16 d;
    memcpy(&d, simt::wave_result() + simt::g_cur->lane * 16, 64);
    simt::xflip();
    return d;
}
// v_mfma_f32_32x32x16_{f16,bf16}: A[i][8 (l / 32) + e] in element e of lane l (i = l % 32), B likewise with j; products are exact in fp32, the
// 16-term sum is formed in double and added to C once (the hardware's internal order is not architected; tests compare with tolerances).
template <class V8> static inline simt_f32x16 simt_mfma_32x32x16(V8 a, V8 b, simt_f32x16 c) {
    static_assert(sizeof(V8) == 16, "8 x 16-bit operands");
    unsigned char* m = simt::xslot_mine();
    memcpy(m, &a, 16);
    memcpy(m + 16, &b, 16);
    if (simt::wave_arrive_last()) {
        float A[32][16], B[32][16];  // A[i][k], B[j][k]
        for (int ln = 0; ln < 64; ++ln) {
            V8 av, bv;
            memcpy(&av, simt::xslot_of(ln), 16);
            memcpy(&bv, simt::xslot_of(ln) + 16, 16);
            for (int e = 0; e < 8; ++e) {
                A[ln & 31][8 * (ln >> 5) + e] = simt_to_f32(av[e]);
                B[ln & 31][8 * (ln >> 5) + e] = simt_to_f32(bv[e]);
            }
        }
        float* R = simt::wave_result();
        for (int i = 0; i < 32; ++i)
            for (int jj = 0; jj < 32; ++jj) {
                double s = 0.0;
                for (int k = 0; k < 16; ++k) s += (double)A[i][k] * (double)B[jj][k];
                R[i * 32 + jj] = (float)s;  // the exact sum of 16 exact products, rounded once; added to C in fp32 below
            }
        simt::wave_release();
    } else {
        simt::wave_wait_released();
    }
    const int l = simt::g_cur->lane, j = l & 31;
    const float* R = simt::wave_result();
    simt_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int i = 8 * (r >> 2) + 4 * (l >> 5) + (r & 3);
        d[r] = d[r] + R[i * 32 + j];
    }
    simt::xflip();
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32(...) simt_mfma_f32_16x16x4f32(__VA_ARGS__)
#define __builtin_amdgcn_mfma_f32_32x32x2f32(...) simt_mfma_f32_32x32x2f32(__VA_ARGS__)
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) simt_mfma_32x32x16(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) simt_mfma_32x32x16(a, b, c)

// ---------------------------------------------------------------------------------------------------------------- buffer resources
struct simt_rsrc {
    char* base;
    unsigned num_records;
};
#define __amdgpu_buffer_rsrc_t simt_rsrc
template <class P> static inline simt_rsrc simt_make_rsrc(P* p, short /*stride*/, int num_records, int /*flags*/) {
    return simt_rsrc{(char*)const_cast<typename std::remove_const<P>::type*>(p), (unsigned)num_records};
}
#define __builtin_amdgcn_make_buffer_rsrc(...) simt_make_rsrc(__VA_ARGS__)
typedef unsigned simt_u32x4 __attribute__((ext_vector_type(4)));
static inline bool simt_in_range(const simt_rsrc& rs, long off, int bytes) { return off >= 0 && (unsigned long)off + (unsigned)bytes <= rs.num_records; }
static inline simt_u32x4 simt_buffer_load_b128(simt_rsrc rs, int voff, int soff, int) {
    simt_u32x4 v = {0, 0, 0, 0};
    if (simt::g_dma_deferred) simt::vm_push_other();
    const long off = (long)(unsigned)voff + (long)(unsigned)soff;  // 32-bit unsigned offsets, as the hardware adds them
    if (simt_in_range(rs, off, 16)) memcpy(&v, rs.base + off, 16);
    return v;
}
static inline unsigned simt_buffer_load_b32(simt_rsrc rs, int voff, int soff, int) {
    unsigned v = 0;
    if (simt::g_dma_deferred) simt::vm_push_other();
    const long off = (long)(unsigned)voff + (long)(unsigned)soff;
    if (simt_in_range(rs, off, 4)) memcpy(&v, rs.base + off, 4);
    return v;
}
template <class T> static inline void simt_buffer_store(T v, simt_rsrc rs, int voff, int soff, int) {
    if (simt::g_dma_deferred) simt::vm_push_other();
    const long off = (long)(unsigned)voff + (long)(unsigned)soff;
    if (simt_in_range(rs, off, (int)sizeof(T))) memcpy(rs.base + off, &v, sizeof(T));
}
#define __builtin_amdgcn_raw_buffer_load_b128(...) simt_buffer_load_b128(__VA_ARGS__)
#define __builtin_amdgcn_raw_buffer_load_b32(...) simt_buffer_load_b32(__VA_ARGS__)
#define __builtin_amdgcn_raw_buffer_store_b32(...) simt_buffer_store(__VA_ARGS__)
#define __builtin_amdgcn_raw_buffer_store_b64(...) simt_buffer_store(__VA_ARGS__)
#define __builtin_amdgcn_raw_buffer_store_b128(...) simt_buffer_store(__VA_ARGS__)
// LDS-DMA: every lane moves `size` bytes to lds + lane * size (the instruction's LDS address is wave-uniform, M0-based)
template <class L> static inline void simt_buffer_load_lds(simt_rsrc rs, L lds, int size, int voff, int soff, int imm, int) {
    char* dst = (char*)(uintptr_t)lds + (long)simt::g_cur->lane * size;
    const long off = (long)(unsigned)voff + (long)(unsigned)soff + imm;
    const bool ok = simt_in_range(rs, off, size);
    if (simt::g_dma_deferred) return simt::vm_push_dma(dst, ok ? rs.base + off : nullptr, size);
    if (ok) memcpy(dst, rs.base + off, size);
    else memset(dst, 0, size);
}
template <class G, class L> static inline void simt_global_load_lds(G src, L lds, int size, int imm, int) {
    char* dst = (char*)(uintptr_t)lds + (long)simt::g_cur->lane * size;
    if (simt::g_dma_deferred) return simt::vm_push_dma(dst, (const char*)(uintptr_t)src + imm, size);
    memcpy(dst, (const char*)(uintptr_t)src + imm, size);
}
#define __builtin_amdgcn_raw_ptr_buffer_load_lds(...) simt_buffer_load_lds(__VA_ARGS__)
#define __builtin_amdgcn_global_load_lds(...) simt_global_load_lds(__VA_ARGS__)

// ---------------------------------------------------------------------------------------------------------------- rewritten inline asm
// (tests/simt/build_emu.py maps the sources' asm statements onto these)
static inline float simt_max3_abs(float acc, float a, float b) { return fmaxf(fmaxf(fabsf(a), fabsf(b)), acc); }
// global_store_dword{,x2,x4} ... sc1: a write-through store is a store (the cache policy has no meaning here)
template <class V>
static inline void simt_store_wt(void* p, V v) { memcpy(p, &v, sizeof(V)); }
// v_fma_mix{lo,hi}_f16 with an fp16 first operand (low / high half of h2), fp32 second and third operands: one rounding, to fp16
static inline unsigned simt_fma_mix_f16(unsigned dst, unsigned h2, float s, float t, int hi) {
    _Float16 h;
    const unsigned short hb = (unsigned short)(hi ? (h2 >> 16) : (h2 & 0xffffu));
    memcpy(&h, &hb, 2);
    const _Float16 r = (_Float16)((double)(float)h * (double)s + (double)t);  // the fp32 fma of these operands is exact in double
    unsigned short rb;
    memcpy(&rb, &r, 2);
    return hi ? ((dst & 0x0000ffffu) | ((unsigned)rb << 16)) : ((dst & 0xffff0000u) | rb);
}
