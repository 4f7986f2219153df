// fp32-accurate implicit GEMM on the bf16 matrix cores of gfx950 (split-operand "bf16x3" / "bf16x6").
//
//   v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the exact v_mfma_f32_32x32x2_f32 (2.5 PF vs 157 TF chip peak).
//   Every fp32 operand is split on the fly into NP bf16 planes  a = a0 + a1 (+ a2)  (each plane the RNE bf16 of the
//   running residual, so NP = 3 represents all 24 significand bits exactly) and the product is rebuilt from the
//   plane products whose weight is above the fp32 rounding level, accumulated in fp32 by the MFMA:
//       NP = 2 (bf16x3):  a0b0 + a0b1 + a1b0                              rel. error ~4e-6  (3 MFMA per k16, 5.3x fp32 rate)
//       NP = 3 (bf16x6):  a0b0 + a0b1 + a1b0 + a0b2 + a2b0 + a1b1         rel. error ~1e-7  (6 MFMA per k16, 2.7x fp32 rate)
//   F16 (f16x3): two fp16 planes, the second one SCALED: a = h + l / 2048 with h = RNE fp16(a), l = RNE fp16(2048 (a - h)).  h and l carry
//   11 significand bits each and the scaling keeps l normal whenever h is, so the pair represents a to 2^-24 |a| like fp32 does;
//   hh goes to one accumulator, hl + lh to a second one that is folded in as acc + acc_c / 2048 in the epilogue (ll < 2^-24 |ab| is
//   dropped).  rel. error ~8e-8 of the plane arithmetic (below the fp32 MFMA's own ~2.5e-7) at the cost of bf16x3: 3 MFMA per k16 and
//   two planes of LDS traffic.  Range: |a| <= 65504 (fp16); an operand beyond that sets *range_flag (cbx_set_range_flag) so that the
//   caller can repeat the computation with bf16x6, which has the fp32 exponent range.  |a| < 6e-5 degrades gracefully (absolute
//   precision 3e-11), a concern only for tensors that are tiny throughout.
//   (fp32 MFMA itself: ~2.5e-7 at K = 256.)  Same address generator / epilogue contract as gemm_f32.hip: Linear, Conv1d,
//   upsample+conv, phase-packed ConvTranspose1d, ragged masking -- only W in [N][K] layout, K tiles of 32.
//
// Workgroup = 8 waves on a 128 x 128 tile (64 x 32 per wave) or 4 waves on 64 x 64; K tiles are register-prefetched TWO
// tiles ahead (the MFMA time of one tile is ~0.3 us, less than one HBM round trip) and staged through a double-buffered
// LDS image of bf16 planes: row stride 80 B keeps both the 8-byte plane stores and the ds_read_b128 operand fetches
// (8 consecutive k per lane: row = lane&31, k = 8*(lane>>5)..+8) conflict-free.
#include <stdlib.h>
#include "cbx_common.h"

namespace {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int SBK = 32;       // K tile (fp32 elements)
constexpr int SLD = SBK + 8;  // LDS row stride in bf16 elements (80 B)

template <int NP, bool F16>
__device__ __forceinline__ void split_store(const f32x4 v, __bf16* dst, int plane_stride, float& amax, bool raw) {
#ifdef CBX_DIAG
    if (raw) {  // diagnosis: no conversion arithmetic, raw halves of the fp32 words
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<f32x2*>(dst) = f32x2{v[0], v[1]};
        *reinterpret_cast<f32x2*>(dst + plane_stride) = f32x2{v[2], v[3]};
        return;
    }
#endif
    if constexpr (F16) {
        // 12 VALU per 4 elements: 2 cvt_pk (h), 2 pk_mul + 4 cvt + 2 pk_fma (2048 v - 2048 h, exact), 2 cvt_pk (l); + 2 max3 (range check)
        const f16x4 h = __builtin_convertvector(v, f16x4);
        *reinterpret_cast<f16x4*>(dst) = h;
        const f32x4 t = v * CBX_F16_LO_SCALE;
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = __builtin_fmaf((float)h[e], -CBX_F16_LO_SCALE, t[e]);
        *reinterpret_cast<f16x4*>(dst + plane_stride) = __builtin_convertvector(r, f16x4);
        cbx_amax4(amax, v);
        return;
    }
    bf16x4 h = __builtin_convertvector(v, bf16x4);
    *reinterpret_cast<bf16x4*>(dst) = h;
    f32x4 r = v - __builtin_convertvector(h, f32x4);
    bf16x4 m = __builtin_convertvector(r, bf16x4);
    *reinterpret_cast<bf16x4*>(dst + plane_stride) = m;
    if constexpr (NP == 3) {
        r = r - __builtin_convertvector(m, f32x4);
        *reinterpret_cast<bf16x4*>(dst + 2 * plane_stride) = __builtin_convertvector(r, bf16x4);
    }
}

// NS = LDS stages.  2: one barrier per K tile (the next tile is written to the other stage while this one is read).  1: half the
// LDS (two workgroups per CU also with three planes) at the price of a second barrier per K tile -- the co-resident workgroup fills
// the bubbles; the register prefetch (two K tiles ahead) is the same.
//
// LD = loader.  0: generic (per-lane pointers, validity mask per K tile, operands zeroed on the way to LDS; serves upsampled inputs and
// K % 32 != 0).  1 / 2 (Linear / Conv1d with up == 1, K % 32 == 0): raw buffer loads -- the row offsets are 32-bit VGPRs computed once
// (per tap for convolutions), the K advance is a scalar offset, and rows outside the tensor / past the ragged length / past K get an
// offset beyond num_records, for which the hardware returns zeros: no address arithmetic, no mask and no select in the K loop
// (the generic loader spends ~180 VALU instructions per K tile and wave, of which ~45 are the plane conversion itself).
// LN (with LD = 1): LayerNorm folded into the A operand -- (a - mean[m]) * rstd[m] * ln_w[k] + ln_b[k] (the expression of norm.hip) is applied
// to the fp32 tile in registers just before the plane split; mean / rstd come from cbx_row_stats_f32, ln_w / ln_b tiles ride along with
// the operand tiles (buffer loads, zeros past K).  Rows past M carry garbage that only reaches output rows that are never stored.
template <int BM, int BN, int WARPS_M, int WARPS_N, int NP, int NS = 2, bool F16 = false, int LD = 0, bool LN = false>
__global__ __launch_bounds__(WARPS_M * WARPS_N * 64) void gemm_split_kernel(const cbx_gemm_t p, int* range_flag) {
    static_assert(!LN || LD == 1, "the folded LayerNorm rides on the Linear buffer-load loader");
    static_assert(!F16 || NP == 2, "the fp16 form has two planes");
    constexpr int BK = SBK;
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int TPR = BK / 4;                 // threads (float4) per tile row
    constexpr int NT = WARPS_M * WARPS_N * 64;
    constexpr int RP = NT / TPR;                // tile rows covered by one pass of float4 loads
    static_assert(BM % RP == 0 && BN % RP == 0, "tile rows must be a multiple of the loader pass");
    constexpr int A_IT = BM / RP, B_IT = BN / RP;
    constexpr int PLANE = (BM + BN) * SLD;      // bf16 elements of one plane of one stage (A rows then B rows)
    constexpr int STAGE = NP * PLANE;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* smem = reinterpret_cast<__bf16*>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WARPS_N, wn = wid % WARPS_N;
    const int z = blockIdx.z, z1 = z / p.nz2, z2 = z - z1 * p.nz2;
    const int tile = cbx_xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int n0 = (tile % gridDim.x) * BN, m0 = (tile / gridDim.x) * BM;

    const float* __restrict__ Ab = p.A + (long)z1 * p.a_s1 + (long)z2 * p.a_s2;
    const float* __restrict__ Wb = p.W + (long)z1 * p.w_s1 + (long)z2 * p.w_s2;
    const int lim = p.lens ? min(p.Tin, p.lens[z1]) : p.Tin;
    const int K = p.K;

    // ---- loader state (see gemm_f32.hip): everything constant along K lives in base pointers, the K walk is incremental
    const int l_row = tid / TPR, a_c4 = (tid % TPR) * 4;
    const float* a_ptr[A_IT];
    int a_row[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + l_row + RP * i;
        a_ok[i] = m < p.M;
        a_row[i] = m * p.stride - p.pad_left;
        a_ptr[i] = Ab + (long)a_row[i] * p.lda + a_c4;
    }
    const float* b_ptr[B_IT];
    bool b_ok[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int n = n0 + l_row + RP * i;
        b_ok[i] = n < p.N;
        b_ptr[i] = Wb + (long)(b_ok[i] ? n : 0) * p.ldw + a_c4;
    }
    int ld_kt = 0, ld_tap = 0, ld_c0 = 0;
    long ld_aoff = 0;
    const long tap_step = (long)p.dil * p.lda - p.Cin;

    // Loads are UNCONDITIONAL (out-of-range lanes read the tensor base and are zeroed when the tile is written to LDS): a
    // predicated load makes hipcc wait for the data at the join, which would serialise the whole prefetch.
    auto load_tiles = [&](f32x4(&ra)[A_IT], f32x4(&rb)[B_IT], unsigned& okmask) {
        const int k0 = ld_kt * BK;
        const int kk = k0 + a_c4;
        const bool kin = kk < K;
        const int tap_rows = ld_tap * p.dil;
        unsigned mask = 0;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int rr = a_row[i] + tap_rows;
            bool ok = a_ok[i] && rr >= 0 && kin;
            const float* src;
            if (p.up > 1) {  // nearest-neighbour upsampled input: row index is not linear in the tap
                const int row = rr / p.up;
                ok = ok && row < lim;
                src = Ab + (long)row * p.lda + ld_c0 + a_c4;
            } else {
                ok = ok && rr < lim;
                src = a_ptr[i] + ld_aoff;
            }
            src = ok ? src : Ab;
            ra[i] = *reinterpret_cast<const f32x4*>(src);
            mask |= ok ? (1u << i) : 0u;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const bool ok = b_ok[i] && kin;
            const float* src = ok ? b_ptr[i] + k0 : Wb;
            rb[i] = *reinterpret_cast<const f32x4*>(src);
            mask |= ok ? (1u << (16 + i)) : 0u;
        }
        okmask = mask;
        ld_kt += 1;
        ld_c0 += BK;
        ld_aoff += BK;
        if (p.taps > 1 && ld_c0 >= p.Cin) {
            ld_c0 = 0;
            ld_tap += 1;
            ld_aoff += tap_step;
        }
    };

    float amax = 0.f;  // F16: largest operand magnitude seen (fp16 range check)
    bool raw = false;
#ifdef CBX_DIAG
    raw = p.reserved0 & 2;
#endif
    // ---- fast loader (LD != 0)
    constexpr int OOB = (int)0x80000000;  // >= num_records: the load returns 0
    const int nk = (K + BK - 1) / BK;
    [[maybe_unused]] __amdgpu_buffer_rsrc_t a_rs, b_rs;
    [[maybe_unused]] int a_vo[A_IT], b_vo[B_IT], a_rr[A_IT], a_lin[A_IT];
    if constexpr (LD != 0) {
        a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ab), 0, OOB, 0x00020000);
        b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wb), 0, OOB, 0x00020000);
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            a_rr[i] = a_ok[i] ? a_row[i] : OOB;                         // input row of tap 0 (negative: never valid)
            a_lin[i] = (a_row[i] * (int)p.lda + a_c4) * 4;              // its byte offset (dispatch guarantees 31 bits)
            a_vo[i] = (a_rr[i] >= 0 && a_rr[i] < lim) ? a_lin[i] : OOB;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) b_vo[i] = b_ok[i] ? ((n0 + l_row + RP * i) * (int)p.ldw + a_c4) * 4 : OOB;
    }
    [[maybe_unused]] __amdgpu_buffer_rsrc_t lw_rs, lb_rs;
    [[maybe_unused]] float ln_mean[A_IT], ln_rstd[A_IT];
    if constexpr (LN) {
        lw_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ln_w), 0, OOB, 0x00020000);
        lb_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ln_b), 0, OOB, 0x00020000);
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int m = m0 + l_row + RP * i;
            ln_mean[i] = a_ok[i] ? p.ln_stats[2 * (long)m] : 0.f;
            ln_rstd[i] = a_ok[i] ? p.ln_stats[2 * (long)m + 1] : 0.f;
        }
    }
    const int tap_bytes = p.dil * (int)p.lda * 4;
    auto load_fast = [&](f32x4(&ra)[A_IT], f32x4(&rb)[B_IT], f32x4& lg, f32x4& lb) {
        const int pe = ld_kt < nk ? 0 : OOB;  // the unrolled loop touches up to three tiles past the end: zeros
        if constexpr (LN) {
            lg = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lw_rs, (a_c4 * 4) | pe, ld_kt * (BK * 4), 0));
            lb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lb_rs, (a_c4 * 4) | pe, ld_kt * (BK * 4), 0));
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_vo[i] | pe, ld_c0 * 4, 0));
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_vo[i] | pe, ld_kt * (BK * 4), 0));
        ld_kt += 1;
        ld_c0 += BK;
        if constexpr (LD == 2) {  // next tap: rows move by dil, validity is re-derived (branch-free: the K loop stays one basic block)
            const bool wrap = ld_c0 >= p.Cin;
            ld_c0 = wrap ? 0 : ld_c0;
            const int dr = wrap ? p.dil : 0, db = wrap ? tap_bytes : 0;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                a_rr[i] += dr;
                a_lin[i] += db;
                a_vo[i] = (a_rr[i] >= 0 && a_rr[i] < lim) ? a_lin[i] : OOB;
            }
        }
    };
    auto load_any = [&](f32x4(&ra)[A_IT], f32x4(&rb)[B_IT], unsigned& okmask, f32x4& lg, f32x4& lb) {
        if constexpr (LD != 0) load_fast(ra, rb, lg, lb);
        else load_tiles(ra, rb, okmask);
    };

    auto store_tiles = [&](int buf, const f32x4(&ra)[A_IT], const f32x4(&rb)[B_IT], unsigned okmask, const f32x4 lg, const f32x4 lb) {
        __bf16* st = smem + buf * STAGE;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            f32x4 a = LD || ((okmask >> i) & 1u) ? ra[i] : zero;
            if constexpr (LN) {
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = (a[e] - ln_mean[i]) * ln_rstd[i] * lg[e] + lb[e];
            }
            split_store<NP, F16>(a, st + (l_row + RP * i) * SLD + a_c4, PLANE, amax, raw);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            split_store<NP, F16>(LD || ((okmask >> (16 + i)) & 1u) ? rb[i] : zero, st + (BM + l_row + RP * i) * SLD + a_c4, PLANE, amax, raw);
    };

    f32x16 acc[TM][TN];
    f32x16 accc[F16 ? TM : 1][F16 ? TN : 1];  // F16: the cross products h*l + l*h, 2048 times too large
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                if constexpr (F16) accc[i][j][r] = 0.f;
            }

    const int lr = lane & 31, lh = lane >> 5;
    auto compute = [&](int buf) {
        const __bf16* as = smem + buf * STAGE + (wm * WM + lr) * SLD + 8 * lh;
        const __bf16* bs = smem + buf * STAGE + (BM + wn * WN + lr) * SLD + 8 * lh;
#pragma unroll
        for (int kc = 0; kc < BK / 16; ++kc) {
            bf16x8 af[TM][NP], bf[TN][NP];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < NP; ++q) af[i][q] = *reinterpret_cast<const bf16x8*>(as + q * PLANE + i * 32 * SLD + kc * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < NP; ++q) bf[j][q] = *reinterpret_cast<const bf16x8*>(bs + q * PLANE + j * 32 * SLD + kc * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (F16) {
                        const f16x8 ah = __builtin_bit_cast(f16x8, af[i][0]), al = __builtin_bit_cast(f16x8, af[i][1]);
                        const f16x8 bh = __builtin_bit_cast(f16x8, bf[j][0]), bl = __builtin_bit_cast(f16x8, bf[j][1]);
                        accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accc[i][j], 0, 0, 0);
                        accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][j], 0, 0, 0);
                        continue;
                    }
                    // smallest-weight products first
                    if constexpr (NP == 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
                }
        }
    };

    // ---- main loop: registers hold tiles kt+1 and kt+2 in flight while tile kt is consumed from LDS
    f32x4 ra[2][A_IT], rb[2][B_IT], lg[2], lbt[2];
    unsigned ok0 = 0, ok1 = 0;
    load_any(ra[0], rb[0], ok0, lg[0], lbt[0]);
    store_tiles(0, ra[0], rb[0], ok0, lg[0], lbt[0]);
    load_any(ra[1], rb[1], ok1, lg[1], lbt[1]);
    __syncthreads();
    // Branch-free body: tiles past the end of K load the tensor base with an all-false mask (zeros in LDS, adds 0), so the
    // number of loads in flight at every wait is the same on every path and hipcc emits exact vmcnt(N) instead of vmcnt(0).
#ifdef CBX_DIAG
    const int dg = p.reserved0;
#define DG(bit) (dg & (bit))
#else
#define DG(bit) 0
#endif
    for (int kt = 0; kt < nk; kt += 2) {
        if (!DG(1)) load_any(ra[0], rb[0], ok0, lg[0], lbt[0]);  // tile kt+2
        if (!DG(4)) compute(0);                          // tile kt
        if constexpr (NS == 1) __syncthreads();  // every wave is done reading the single stage
        if (!DG(8)) store_tiles(NS == 1 ? 0 : 1, ra[1], rb[1], ok1, lg[1], lbt[1]);   // tile kt+1
        __syncthreads();
        if (!DG(1)) load_any(ra[1], rb[1], ok1, lg[1], lbt[1]);  // tile kt+3
        if (!DG(4)) compute(NS == 1 ? 0 : 1);            // tile kt+1 (all zero when nk is odd and this is past the end)
        if constexpr (NS == 1) __syncthreads();
        if (!DG(8)) store_tiles(0, ra[0], rb[0], ok0, lg[0], lbt[0]);   // tile kt+2
        __syncthreads();
    }
#undef DG

    if constexpr (F16) {
        if (amax > 65504.f && range_flag) atomicOr(range_flag, 1);
    }

    // ---- epilogue.  C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* Cb = p.C + (long)z1 * p.c_s1 + (long)z2 * p.c_s2;
    const float* Rb = p.R ? p.R + (long)z1 * p.r_s1 + (long)z2 * p.r_s2 : nullptr;
    float* C2b = p.C2 ? p.C2 + (long)z1 * p.c2_s1 + (long)z2 * p.c2_s2 : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + lr;
        if (n >= p.N) continue;
        const float bia = p.bias ? p.bias[n] : 0.f;
        const float a1 = p.act1_param ? p.act1_param[n] : 0.f;
        const float a2 = p.act2_param ? p.act2_param[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * WM + i * 32 + 4 * lh;
            // residual / beta operands of the 16 rows are fetched back to back (clamped rows), not one wait per element
            float res[16], old[16];
            if (Rb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) res[r] = Rb[(long)min(mb + (r & 3) + 8 * (r >> 2), p.M - 1) * p.ldr + n];
            }
            if (p.beta != 0.f) {
#pragma unroll
                for (int r = 0; r < 16; ++r) old[r] = Cb[(long)min(mb + (r & 3) + 8 * (r >> 2), p.M - 1) * p.ldc + n];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                float v = acc[i][j][r];
                if constexpr (F16) v += accc[i][j][r] * (1.0f / CBX_F16_LO_SCALE);
                v += bia;
                v = cbx_act(v, p.act1, p.act1_slope, a1);
                if (Rb) v += res[r];
                v *= p.alpha;
                if (p.beta != 0.f) v += p.beta * old[r];
                if (m < p.M) {
                    Cb[(long)m * p.ldc + n] = v;
                    if (C2b) C2b[(long)m * p.ldc2 + n] = cbx_act(v, p.act2, p.act2_slope, a2);
                }
            }
        }
    }
}

int* g_range_flags[64] = {nullptr};  // one word per device ordinal (cbx_set_range_flag)

template <int BM, int BN, int WARPS_M, int WARPS_N, int NP, int NS = 2, bool F16 = false, int LD = 0, bool LN = false>
int launch_split(const cbx_gemm_t& p, hipStream_t st) {
    constexpr size_t lds = (size_t)NS * NP * (BM + BN) * SLD * sizeof(__bf16);
    auto kern = gemm_split_kernel<BM, BN, WARPS_M, WARPS_N, NP, NS, F16, LD, LN>;
    static unsigned long long configured = 0;  // > 64 KiB of dynamic LDS has to be opted into once per kernel AND device (one bit per ordinal)
    const int dev = cbx_device();
    if (!(configured >> dev & 1)) {
        const size_t most = lds > 96 * 1024 ? lds : 96 * 1024;  // (a co-resident stream asks for up to 81 KiB: one workgroup per CU)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)most);
        if (e != hipSuccess) return cbx_set_error((int)e, "gemm_split: cannot reserve %zu B of LDS: %s", most, hipGetErrorString(e));
        configured |= 1ull << dev;
    }
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, p.nz1 * p.nz2);
    hipLaunchKernelGGL(kern, grid, dim3(WARPS_M * WARPS_N * 64), cbx_coresident_lds(st, lds, 1), st, p, cbx_range_flag());
    return cbx_check_launch("gemm_split");
}

}  // namespace

static int g_split_tile = 0;
// tuning knob: 0 = automatic, 64 / 12864 / 128 force a tile shape, 1286401 / 12801 = the single-LDS-stage forms
extern "C" int cbx_set_split_tile(int t) {
    g_split_tile = t;
    return 0;
}

// Device word that the fp16 forms (precision 16: here, gemm_planes.hip, attention_split.hip, norm.hip) OR a 1 into when an operand exceeds
// the fp16 range.  NULL (default) = not reported.  ONE WORD PER DEVICE: cbx_set_range_flag registers the word for the calling thread's
// current device, a launch reports into the word of the device it is launched on (an engine per GPU may share one process).
int* cbx_range_flag() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return nullptr;
    return g_range_flags[d];
}
extern "C" int cbx_set_range_flag(int* dev_flag) {
    int d = 0;
    hipError_t e = hipGetDevice(&d);
    if (e != hipSuccess || d < 0 || d >= 64) return cbx_set_error(CBX_EINVAL, "set_range_flag: no current device");
    g_range_flags[d] = dev_flag;
    return 0;
}

// Would a Linear (M, N?, K) with row stride lda fold a LayerNorm into its A operand (cbx_gemm_t.ln_stats)?  Mirrors the conditions of the
// dispatcher below that do not depend on the caller's other operands: no forced tile, the buffer-load loader usable (31-bit offsets),
// K % 32 == 0.  The caller additionally needs precision 16, taps == 1, one batch, no lens.
static int split_generic_loader_forced() {
    return 0;  // (the generic loader of round 1 is kept for shapes the buffer-load loader does not serve)
}
extern "C" int cbx_gemm_ln_fusable(long M, int K, long lda) {
    if (g_split_tile || split_generic_loader_forced()) return 0;
    return M > 32 && K > 0 && K % SBK == 0 && (M + 1) * lda * 4 < 0x7fffffffL;
}

// Called by cbx_gemm_f32 after argument validation.  planes = 2 (bf16x3), 3 (bf16x6) or 16 (f16x3).  Returns -1 when the shape is not
// served by this kernel (caller falls back to the exact fp32 MFMA kernel).
int cbx_gemm_split_dispatch(const cbx_gemm_t& p, int planes, hipStream_t st) {
    if (p.w_kn || p.swiglu) return -1;
    if (p.taps > 1 && p.Cin % SBK != 0) return -1;  // a K tile must not straddle two conv taps
    const int force = g_split_tile;
    const long g128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.nz1 * p.nz2;
    // measured on the CFM shapes (scripts/bench_gemm.py): 128x64 / 8 waves / 2 workgroups per CU beats 128x128 (1 per CU) by
    // 5-45 % and 64x64 by 0-15 %
    int tile = force ? force : (g128 >= 64 && p.N >= 64 ? 12864 : 64);
    // loader: 1 / 2 = raw buffer loads (Linear / Conv1d), needs K tiles that never straddle the end of K and 31-bit byte offsets
    const int no_fast = split_generic_loader_forced();
    const bool fast = !no_fast && p.up == 1 && p.K % SBK == 0 && (long)(p.Tin + 1) * p.lda * 4 < 0x7fffffffL &&
                      (long)(p.N + 128) * p.ldw * 4 < 0x7fffffffL && (long)p.pad_left * p.lda * 4 < 0x3fffffffL;
    const int ld = !fast ? 0 : p.taps == 1 ? 1 : 2;
    if (p.ln_stats) {  // LayerNorm folded into the A operand: f16x3 Linear on the buffer-load loader only -- anything else is a caller error
        if (planes != 16 || ld != 1 || force || p.nz1 * p.nz2 != 1 || p.lens || !p.ln_w || !p.ln_b)
            return cbx_set_error(CBX_EINVAL, "gemm: ln_stats needs precision 16, taps == 1, up == 1, K %% 32 == 0, one batch, no lens, ln_w and ln_b");
        if (tile == 12864 && p.K <= 256) return launch_split<128, 64, 4, 2, 2, 1, true, 1, true>(p, st);
        if (tile == 12864) return launch_split<128, 64, 4, 2, 2, 2, true, 1, true>(p, st);
        return launch_split<64, 64, 2, 2, 2, 2, true, 1, true>(p, st);
    }
    if (ld && !force) {  // the default tiles with the fast loader
        const bool wide = tile == 12864;
        if (planes == 16) {
            // K <= 256 (8 K tiles: the transformer blocks' q/k/v and ff1 projections): the single-stage form is 3-10 % faster; other wave
            // shapes (64x32 / 64x64 per wave, 128x128 and 256x64 tiles) all measured slower: profiles/r02_bench_gemm_f16x3_wave_shapes_rejected.log
            if (wide && ld == 1 && p.K <= 256) return launch_split<128, 64, 4, 2, 2, 1, true, 1>(p, st);
            if (wide) return ld == 1 ? launch_split<128, 64, 4, 2, 2, 2, true, 1>(p, st) : launch_split<128, 64, 4, 2, 2, 2, true, 2>(p, st);
            return ld == 1 ? launch_split<64, 64, 2, 2, 2, 2, true, 1>(p, st) : launch_split<64, 64, 2, 2, 2, 2, true, 2>(p, st);
        }
        if (planes == 2) {
            if (wide) return ld == 1 ? launch_split<128, 64, 4, 2, 2, 2, false, 1>(p, st) : launch_split<128, 64, 4, 2, 2, 2, false, 2>(p, st);
            return ld == 1 ? launch_split<64, 64, 2, 2, 2, 2, false, 1>(p, st) : launch_split<64, 64, 2, 2, 2, 2, false, 2>(p, st);
        }
        if (wide) return ld == 1 ? launch_split<128, 64, 4, 2, 3, 1, false, 1>(p, st) : launch_split<128, 64, 4, 2, 3, 1, false, 2>(p, st);
        return ld == 1 ? launch_split<64, 64, 2, 2, 3, 2, false, 1>(p, st) : launch_split<64, 64, 2, 2, 3, 2, false, 2>(p, st);
    }
    if (planes == 16) {
        if (tile == 12801) return launch_split<128, 128, 2, 4, 2, 1, true>(p, st);
        if (tile == 1286401) return launch_split<128, 64, 4, 2, 2, 1, true>(p, st);
        if (tile == 128) return launch_split<128, 128, 2, 4, 2, 2, true>(p, st);
        if (tile == 12864) return launch_split<128, 64, 4, 2, 2, 2, true>(p, st);
        return launch_split<64, 64, 2, 2, 2, 2, true>(p, st);
    }
    if (planes == 2) {
        if (tile == 12801) return launch_split<128, 128, 2, 4, 2, 1>(p, st);
        if (tile == 1286401) return launch_split<128, 64, 4, 2, 2, 1>(p, st);
        if (tile == 128) return launch_split<128, 128, 2, 4, 2>(p, st);
        if (tile == 12864) return launch_split<128, 64, 4, 2, 2>(p, st);
        return launch_split<64, 64, 2, 2, 2>(p, st);
    }
    if (tile == 1286401) return launch_split<128, 64, 4, 2, 3, 1>(p, st);
    if (tile == 12801) return launch_split<128, 128, 2, 4, 3, 1>(p, st);
    if (tile == 128) return launch_split<128, 128, 2, 4, 3>(p, st);
    // three planes: the double-buffered 128x64 image is 92 KB (one workgroup per CU); the single-stage form (46 KB, two per CU) is
    // 20-55 % faster on every CFM / encoder shape (scripts/bench_gemm.py, gpurun_out/bench_gemm_ns1.log)
    if (tile == 12864) return force ? launch_split<128, 64, 4, 2, 3>(p, st) : launch_split<128, 64, 4, 2, 3, 1>(p, st);
    return launch_split<64, 64, 2, 2, 3>(p, st);
}
