// f16x3 implicit GEMM on PLANE-FORMAT operands (round 3): the arithmetic of gemm_split.hip's F16 form without its per-tile conversion.
//
//   An fp32 tensor X lives in memory as two fp16 planes  X = h + l / 2048,  h = RNE16(X), l = RNE16(2048 (X - h))  (22 significand
//   bits, see gemm_split.hip).  Weights are split ONCE at load; activations are written in plane format by the kernel that produces
//   them (GEMM / LayerNorm / attention epilogues, cbx_split_planes_f32) -- the same 4 bytes per element as fp32.  The consumer's K loop
//   is then a plain fp16 MFMA loop:  acc += Ah Bh,  accc += Ah Bl + Al Bh,  C = acc + accc / 2048 (+ epilogue), with
//     * operand tiles moved global -> LDS by global_load_lds (16 B per lane, no VGPR round trip, no VALU, no ds_write), NS = 2 or 3 LDS
//       stages, ONE barrier per K tile: the loads of tiles t+1 (.. t+NS-1) are in flight while tile t is multiplied; the wait in front of
//       the barrier is a COUNTED vmcnt (raw s_barrier: __syncthreads() would drain every DMA in flight);
//     * the LDS image lane-linear as the DMA writes it, bank conflicts removed by an XOR swizzle applied to the SOURCE address and to
//       the ds_read_b128 address (cdna_hip_programming.md rule 21): chunk c of row r sits at slot r*CH + (c ^ f(r)),
//       f(r) = (r / (16 / CH)) & (CH - 1), CH = 16-byte chunks per row and plane (BK / 8);
//     * masked rows (M tail, causal-conv left padding, rows past a ragged length) fetched from a 16-byte zero page instead of being
//       predicated: every lane of every load instruction writes its LDS slot, so no tile is ever partially stale.
//   Same implicit-GEMM address map as gemm_split.hip for Linear and Conv1d (taps, dilation, stride, left pad, per-batch lengths);
//   no upsampling, W in [N][K] layout, K % BK == 0, Cin % BK == 0.
//
// Replaces F.linear / F.conv1d of the CFM estimator (reference models/s3gen/decoder.py:243-333, matcha/transformer.py:243-316,
// matcha/decoder.py:56-61).
#include <stdlib.h>
#include "cbx_common.h"

namespace {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __attribute__((aligned(16))) const unsigned cbx_zero_page[4] = {0u, 0u, 0u, 0u};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// epilogue activations of the CFM / encoder GEMMs (a compile-time-small switch: the generic cbx_act() unrolled 16 x TM x TN times is
// most of a kernel's code)
__device__ __forceinline__ float pl_act(float v, int act, float slope) {
    if (act == CBX_ACT_GELU_ERF) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    if (act == CBX_ACT_SILU) return v / (1.0f + __expf(-v));
    if (act == CBX_ACT_LRELU) return v > 0.0f ? v : v * slope;
    return v;
}

template <int BM, int BN, int WARPS_M, int WARPS_N, int BK, int NS>
__global__ __launch_bounds__(WARPS_M * WARPS_N * 64) void gemm_pl_kernel(const cbx_gemm_pl_t p, int* range_flag) {
    constexpr int NWV = WARPS_M * WARPS_N;
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int CH = BK / 8;                      // 16-byte chunks per row and plane
    constexpr int RPS = 16 / CH;                    // rows per swizzle step
    constexpr int PLANE_SLOTS = (BM + BN) * CH;     // A rows then B rows
    constexpr int STAGE_SLOTS = 2 * PLANE_SLOTS;    // plane h, plane l
    constexpr int STAGE_BYTES = STAGE_SLOTS * 16;
    constexpr int NLOAD = STAGE_SLOTS / 64;         // wave-level DMA instructions (1 KiB each) per stage
    static_assert((BM * CH) % 64 == 0 && (BN * CH) % 64 == 0, "a DMA instruction must not straddle the A / B boundary");
    static_assert(NLOAD % NWV == 0, "DMA instructions must divide evenly over the waves");
    static_assert(WM % 32 == 0 && WN % 32 == 0 && (CH == 4 || CH == 8), "tile shape");
    constexpr int LPW = NLOAD / NWV;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WARPS_N, wn = wid % WARPS_N;
    const int z = blockIdx.z;
    const int tile = cbx_xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int n0 = (tile % gridDim.x) * BN, m0 = (tile / gridDim.x) * BM;

    const _Float16* Ab = reinterpret_cast<const _Float16*>(p.A) + (long)z * p.a_s1;
    const _Float16* Wb = reinterpret_cast<const _Float16*>(p.W) + (long)z * p.w_s1;
    const int lim = p.lens ? min(p.Tin, p.lens[z]) : p.Tin;
    const int nk = p.K / BK;

    // ---- DMA descriptors: load i of this wave fills slots [L*64, L*64 + 64) of a stage, L = wid * LPW + i; the lane's slot fixes
    //      (plane, row, chunk) for the whole K walk
    const _Float16* ptr[LPW];
    int trow[LPW];        // A: input row of the current tap (validity 0 <= trow < lim); B: 0
    int tlim[LPW];        // A: lim (or 0 for rows >= M); B: 1 / 0
    int tstep[LPW];       // A: dil; B: 0
    long wstep[LPW];      // pointer step at a tap wrap (halves): A: dil * lda - Cin; B: 0
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int s = (wid * LPW + i) * 64 + lane;
        const int q = s / PLANE_SLOTS, rs = s % PLANE_SLOTS;
        const int row = rs / CH, pc = rs % CH;
        const int c = pc ^ ((row / RPS) & (CH - 1));
        if (row < BM) {
            const int m = m0 + row;
            const int t0 = m * p.stride - p.pad_left;
            trow[i] = t0;
            tlim[i] = m < p.M ? lim : 0;
            tstep[i] = p.dil;
            wstep[i] = (long)p.dil * p.lda - p.Cin;
            ptr[i] = Ab + (long)t0 * p.lda + (q ? p.a_lo : 0) + c * 8;
        } else {
            const int n = n0 + row - BM;
            trow[i] = 0;
            tlim[i] = n < p.N ? 1 : 0;
            tstep[i] = 0;
            wstep[i] = 0;
            ptr[i] = Wb + (long)(n < p.N ? n : 0) * p.ldw + (q ? p.w_lo : 0) + c * 8;
        }
    }
    int ld_c0 = 0;
    const _Float16* zp = reinterpret_cast<const _Float16*>(cbx_zero_page);
    auto issue = [&](int stage) {
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const bool ok = (unsigned)trow[i] < (unsigned)tlim[i];
            const _Float16* src = ok ? ptr[i] : zp;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + stage * STAGE_BYTES + (wid * LPW + i) * 1024), 16, 0, 0);
        }
        ld_c0 += BK;
        const bool wrap = ld_c0 >= p.Cin;  // wave-uniform: next tile starts the next conv tap
        ld_c0 = wrap ? 0 : ld_c0;
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            ptr[i] += BK + (wrap ? wstep[i] : 0L);
            trow[i] += wrap ? tstep[i] : 0;
        }
    };

    f32x16 acc[TM][TN], accc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = accc[i][j][r] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    const int swz = (lr / RPS) & (CH - 1);  // rows of a fragment are base + lr with base % 32 == 0: f(row) = f(lr)
    const int a_off = (wm * WM + lr) * CH * 16, b_off = (BM + wn * WN + lr) * CH * 16;
    auto compute = [&](int stage) {
        const unsigned char* st = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int kc = 0; kc < BK / 16; ++kc) {
            const int co = ((kc * 2 + lh) ^ swz) << 4;
            f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const f16x8*>(st + a_off + i * 32 * CH * 16 + co);
                al[i] = *reinterpret_cast<const f16x8*>(st + PLANE_SLOTS * 16 + a_off + i * 32 * CH * 16 + co);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const f16x8*>(st + b_off + j * 32 * CH * 16 + co);
                bl[j] = *reinterpret_cast<const f16x8*>(st + PLANE_SLOTS * 16 + b_off + j * 32 * CH * 16 + co);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accc[i][j], 0, 0, 0);
                    accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    };

    // ---- main loop: NS LDS stages, one barrier per K tile.  Before the barrier every wave waits until ITS DMAs of tile kt have landed
    //      (counted: the NS - 2 younger tiles stay in flight); after it tile kt is complete for everybody and nobody reads the stage
    //      that tile kt + NS - 1 is about to overwrite (it held tile kt - 1).
    static_assert(NS == 2 || NS == 3, "two or three LDS stages");
#ifdef CBX_DIAG  // scripts/diag_planes.sh: parts of the kernel switched off (1 no DMA after the prologue, 2 no ds_read / MFMA, 4 no epilogue stores)
    const int dg = p.reserved0;
#define DG(bit) (dg & (bit))
#else
#define DG(bit) 0
#endif
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) issue(s);
    for (int kt = 0; kt < nk; ++kt) {
        if (NS == 3 && kt + 1 < nk && !DG(1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + NS - 1 < nk && !DG(1)) issue((kt + NS - 1) % NS);
        if (!DG(2)) compute(kt % NS);
    }

    // ---- epilogue.  C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    //      Lean by construction: every store / residual load is ONE buffer instruction (32-bit lane offset + a scalar row offset); rows >= M fall
    //      behind num_records and are dropped by the hardware, lanes whose column is >= N carry an out-of-range offset; the activation
    //      and the output kinds are branched on once per tile, never per element.
    constexpr int OOB = (int)0x80000000;
    const int Mrem = p.M;
    const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc(
        p.C ? p.C + (long)z * p.c_s1 : nullptr, 0, p.C ? (int)((((long)Mrem - 1) * p.ldc + p.N) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.R ? p.R + (long)z * p.r_s1 : nullptr), 0, p.R ? (int)((((long)Mrem - 1) * p.ldr + p.N) * 4) : 0, 0x00020000);
    // plane output: the l plane lies p_lo halves after the h plane of the same row, so one descriptor covers both
    const __amdgpu_buffer_rsrc_t p_rs = __builtin_amdgcn_make_buffer_rsrc(
        p.P ? reinterpret_cast<_Float16*>(p.P) + (long)z * p.p_s1 : nullptr, 0, p.P ? (int)((((long)Mrem - 1) * p.ldp + p.p_lo + p.N) * 2) : 0, 0x00020000);
    const bool hasC = p.C != nullptr, hasP = p.P != nullptr, hasR = p.R != nullptr;
    const int ldc4 = (int)p.ldc * 4, ldr4 = (int)p.ldr * 4, ldp2 = (int)p.ldp * 2;
    const bool odd = lr & 1;
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + lr;
        const bool nok = n < p.N;
        const float bia = p.bias ? p.bias[nok ? n : 0] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * WM + i * 32 + 4 * lh;  // row of register 0
            float v[16], res[16];
            if (hasR) {
                const int ro = nok ? mb * ldr4 + n * 4 : OOB;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    res[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rs, ro, ((r & 3) + 8 * (r >> 2)) * ldr4, 0));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = __builtin_fmaf(accc[i][j][r], 1.0f / CBX_F16_LO_SCALE, acc[i][j][r]) + bia;
            if (p.act == CBX_ACT_GELU_ERF) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = 0.5f * v[r] * (1.0f + erff(v[r] * 0.70710678118654752f));
            } else if (p.act != CBX_ACT_NONE) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = pl_act(v[r], p.act, p.act_slope);
            }
            if (hasR) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += res[r];
            }
            if (p.alpha != 1.0f) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] *= p.alpha;
            }
            if (hasC && !DG(4)) {
                const int co = nok ? mb * ldc4 + n * 4 : OOB;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), c_rs, co, ((r & 3) + 8 * (r >> 2)) * ldc4, 0);
            }
            if (hasP) {
                // The lane pair (even, odd column) owns columns n, n + 1 of every row.  Of each register pair (rows m, m + 1) the even lane
                // converts row m and the odd lane row m + 1: one exchange gives each lane both columns of its row, which it splits into the h
                // word and the l word (two 4-byte stores per lane and register pair instead of a conversion per element).
                const int ne = n & ~1;  // first column of the pair
                const int po = (ne < p.N ? mb * ldp2 + ne * 2 : OOB) + (odd ? ldp2 : 0);
                const int plo2 = (int)p.p_lo * 2;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float give = odd ? v[r] : v[r + 1];
                    const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
                    const float c0 = odd ? got : v[r], c1 = odd ? v[r + 1] : got;  // columns ne, ne + 1 of this lane's row
                    unsigned h2, l2;
                    cbx_split2(c0, c1, h2, l2);
                    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(c0), "v"(c1));
                    if (!DG(4)) {
                        const int so = ((r & 3) + 8 * (r >> 2)) * ldp2;
                        __builtin_amdgcn_raw_buffer_store_b32(h2, p_rs, po, so, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(l2, p_rs, po + plo2, so, 0);
                    }
                }
            }
        }
    }
    if (hasP && amax > 65504.f && range_flag) atomicOr(range_flag, 1);
}

// x (rows, C) fp32 -> planes.  One float4 per thread; two 8-byte stores.
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, _Float16* __restrict__ P, long rows, int C4,
                                                           long ldx, long ldp, long p_lo, int* range_flag) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * C4) return;
    const long r = i / C4;
    const int c = (int)(i - r * C4) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
    const f16x4 h = __builtin_convertvector(v, f16x4);
    const f32x4 t = v * CBX_F16_LO_SCALE;
    f32x4 d;
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = __builtin_fmaf((float)h[e], -CBX_F16_LO_SCALE, t[e]);
    *reinterpret_cast<f16x4*>(P + r * ldp + c) = h;
    *reinterpret_cast<f16x4*>(P + r * ldp + p_lo + c) = __builtin_convertvector(d, f16x4);
    float amax = 0.f;
    cbx_amax4(amax, v);
    if (amax > 65504.f && range_flag) atomicOr(range_flag, 1);
}

template <int BM, int BN, int WARPS_M, int WARPS_N, int BK, int NS = 2>
int launch_pl(const cbx_gemm_pl_t& p, hipStream_t st) {
    constexpr size_t lds = (size_t)NS * 2 * (BM + BN) * (BK / 8) * 16;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = gemm_pl_kernel<BM, BN, WARPS_M, WARPS_N, BK, NS>;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return cbx_set_error((int)e, "gemm_planes: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
        configured = true;
    }
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, p.nz1);
    hipLaunchKernelGGL(kern, grid, dim3(WARPS_M * WARPS_N * 64), lds, st, p, cbx_range_flag());
    return cbx_check_launch("gemm_planes");
}

}  // namespace

static int g_pl_tile = getenv("CBX_PL_TILE") ? atoi(getenv("CBX_PL_TILE")) : 0;
extern "C" int cbx_set_planes_tile(int t) {
    g_pl_tile = t;
    return 0;
}

extern "C" int cbx_gemm_planes(const cbx_gemm_pl_t* pp, void* stream) {
    cbx_gemm_pl_t p = *pp;
    if (p.taps < 1) p.taps = 1;
    if (p.dil < 1) p.dil = 1;
    if (p.stride < 1) p.stride = 1;
    if (p.nz1 < 1) p.nz1 = 1;
    if (p.Cin <= 0) p.Cin = p.K / p.taps;
    if (p.Tin <= 0) p.Tin = p.M;
    if (p.alpha == 0.f) p.alpha = 1.f;
    CBX_REQUIRE(p.A && p.W && (p.C || p.P), "gemm_planes: null operand");
    CBX_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0 && p.K == p.taps * p.Cin, "gemm_planes: bad shape M=%d N=%d K=%d taps=%d Cin=%d", p.M, p.N, p.K, p.taps, p.Cin);
    CBX_REQUIRE(p.Cin % 32 == 0, "gemm_planes: Cin=%d must be a multiple of 32 (a K tile must not straddle two conv taps)", p.Cin);
    CBX_REQUIRE((p.lda | p.a_lo | p.a_s1 | p.ldw | p.w_lo) % 8 == 0 && (((uintptr_t)p.A | (uintptr_t)p.W) & 15) == 0,
                "gemm_planes: operand planes must be 16-byte aligned (strides / plane offsets multiples of 8 halves)");
    CBX_REQUIRE(!p.P || ((p.N | p.ldp | p.p_lo | p.p_s1) % 2 == 0 && ((uintptr_t)p.P & 3) == 0), "gemm_planes: plane output needs even N / strides");
    CBX_REQUIRE((!p.C || ((long)p.M * p.ldc + p.N) * 4 < 0x7fffffffL) && (!p.R || ((long)p.M * p.ldr + p.N) * 4 < 0x7fffffffL) &&
                    (!p.P || ((long)p.M * p.ldp + p.p_lo + p.N) * 2 < 0x7fffffffL),
                "gemm_planes: one batch of an output / residual must span less than 2 GiB (32-bit buffer offsets)");
    CBX_REQUIRE((!p.C || p.ldc >= p.N) && (!p.R || p.ldr >= p.N) && (!p.P || (p.p_lo > 0 && p.ldp >= p.p_lo + p.N)),
                "gemm_planes: rows must not overlap (ldc, ldr >= N; plane output rows hold [h | l]: ldp >= p_lo + N)");
    hipStream_t st = (hipStream_t)stream;
    const int force = g_pl_tile;
    const bool k64 = p.Cin % 64 == 0;
    // tile menu (BM x BN, waves, wave tile, BK, LDS stages); the automatic choice below is the measured one: profiles/r03_gemm_planes_tiles.log
    switch (force) {
        case 1: return launch_pl<128, 64, 4, 2, 32>(p, st);       // 8 waves 32x32, 48 KB
        case 2: return launch_pl<128, 64, 2, 2, 32>(p, st);       // 4 waves 64x32, 48 KB
        case 3: return launch_pl<128, 128, 2, 2, 32>(p, st);      // 4 waves 64x64, 64 KB
        case 4: return launch_pl<128, 128, 4, 2, 32>(p, st);      // 8 waves 32x64, 64 KB
        case 5: if (k64) return launch_pl<128, 64, 4, 2, 64>(p, st); break;    // 96 KB
        case 6: if (k64) return launch_pl<128, 64, 2, 2, 64>(p, st); break;    // 96 KB
        case 7: if (k64) return launch_pl<128, 128, 2, 2, 64>(p, st); break;   // 128 KB
        case 8: if (k64) return launch_pl<128, 128, 4, 2, 64>(p, st); break;   // 128 KB
        case 9: return launch_pl<64, 64, 2, 2, 32>(p, st);        // 4 waves 32x32, 32 KB
        case 10: return launch_pl<256, 64, 4, 2, 32>(p, st);      // 8 waves 64x32, 80 KB
        case 11: return launch_pl<256, 128, 4, 2, 32>(p, st);     // 8 waves 64x64, 96 KB
        case 12: return launch_pl<128, 128, 2, 2, 32, 3>(p, st);  // 4 waves 64x64, 3 stages, 96 KB
        case 13: return launch_pl<128, 64, 2, 2, 32, 3>(p, st);   // 4 waves 64x32, 3 stages, 72 KB
        case 14: return launch_pl<128, 128, 2, 4, 32>(p, st);     // 8 waves 64x32, 64 KB
        case 15: return launch_pl<128, 256, 2, 4, 32>(p, st);     // 8 waves 64x64, 96 KB
        case 16: return launch_pl<64, 128, 2, 2, 32>(p, st);      // 4 waves 32x64, 48 KB
        case 17: return launch_pl<128, 128, 4, 2, 32, 3>(p, st);  // 8 waves 32x64, 3 stages, 96 KB
        default: break;
    }
    // automatic choice (profiles/r03_bench_planes_tiles.log, rows 16 x T 1000): 8-wave workgroups everywhere (more waves to overlap DMA issue,
    // MFMA and the epilogue); wide outputs 2 x 4 waves of 64 x 32, narrow outputs with a long K the BK = 64 form, N <= 96 half-width tiles
    const long g128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.nz1;
    if (g128 < 64) return launch_pl<64, 64, 2, 2, 32>(p, st);
    if (p.N <= 96) return launch_pl<128, 64, 4, 2, 32>(p, st);
    if (p.N >= 512) return launch_pl<128, 128, 2, 4, 32>(p, st);
    if (k64 && p.K >= 512) return launch_pl<128, 128, 4, 2, 64>(p, st);
    return launch_pl<128, 128, 4, 2, 32>(p, st);
}

extern "C" int cbx_split_planes_f32(const float* x, void* planes, long rows, int C, long ldx, long ldp, long p_lo, void* stream) {
    CBX_REQUIRE(x && planes && C % 4 == 0 && ldx % 4 == 0 && ldp % 4 == 0 && p_lo % 4 == 0, "split_planes: C, ldx, ldp, p_lo must be multiples of 4");
    CBX_REQUIRE((((uintptr_t)x) & 15) == 0 && (((uintptr_t)planes) & 7) == 0, "split_planes: alignment");
    if (rows <= 0) return 0;
    const long n = rows * (C / 4);
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       reinterpret_cast<_Float16*>(planes), rows, C / 4, ldx, ldp, p_lo, cbx_range_flag());
    return cbx_check_launch("split_planes");
}
