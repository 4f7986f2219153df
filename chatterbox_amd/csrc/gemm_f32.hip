// Implicit-GEMM linear / conv1d / batched matmul on the fp32 matrix cores of gfx950.
//
//   v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain), 64 FLOP/clk/SIMD = 157 TF chip peak.
//
// One workgroup (4 or 8 waves) owns a BM x BN output tile; K is walked in 16-float
// tiles that are register-prefetched from HBM while the previous tile is consumed from a double-buffered
// LDS image (one s_barrier per K tile).  A wave's MFMA operand for k-step s is one float per lane; the two
// half-waves take k = 4h+s of each 8-deep k-block (h = lane>>5), so every lane fetches its four operands with
// ONE ds_read_b128 from a row of stride 20 floats (conflict-free for the b128 lane groups).
//
// The A operand is generated on the fly from a channel-last activation tensor: K = taps*Cin and the row of tap j
// for output row m is (m*stride + j*dil - pad_left)/up, zero outside [0, min(Tin, lens[z1])) -- that single
// address generator gives Linear, Conv1d (any stride/dilation/padding, causal or not), nearest-upsample+conv,
// phase-packed ConvTranspose1d and ragged-batch masking without ever materialising im2col.
#include <stdlib.h>
#include "cbx_common.h"

namespace {

// K tile: 16 or 32 floats.  LDS row stride BK+4 floats (80 / 144 B): 16-B aligned and conflict-free for the b128 lane groups

// LD = 1 (only with !W_KN; up == 1, K % BK == 0, 31-bit byte offsets -- checked by the dispatcher): the loader of gemm_split.hip.  Row offsets
// are 32-bit VGPRs computed once (per tap for convolutions), the K advance is a scalar offset of raw buffer loads, rows outside the
// tensor / ragged length / K read zeros from the hardware; tiles are prefetched TWO ahead and the K loop is one basic block (the generic
// loader predicates every load, which makes hipcc wait for all of them at the join, and prefetches one tile ahead).  Same MFMA order:
// results are bit-identical to LD = 0.
template <int BM, int BN, int WARPS_M, int WARPS_N, bool W_KN, int BK, int LD = 0>
__global__ __launch_bounds__(WARPS_M * WARPS_N * 64) void gemm_f32_kernel(const cbx_gemm_t p) {
    static_assert(!LD || !W_KN, "the buffer-load loader serves W in [N][K] layout");
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int LDS_LD = BK + 4;
    constexpr int TPR = BK / 4;                // threads (float4) per tile row
    constexpr int NT = WARPS_M * WARPS_N * 64;  // threads per workgroup (4 or 8 waves)
    constexpr int RP = NT / TPR;               // tile rows covered by one pass of float4 loads
    constexpr int A_IT = (BM + RP - 1) / RP;   // float4 loads per thread per K tile (A)
    constexpr int B_IT = (BN + RP - 1) / RP;
    constexpr bool A_PART = (BM % RP) != 0;    // e.g. BM = 32 with 256 threads: only threads 0..127 carry A rows
    constexpr bool B_PART = (BN % RP) != 0;

    __shared__ __attribute__((aligned(16))) float As[2][BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDS_LD];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WARPS_N, wn = wid % WARPS_N;
    const int z = blockIdx.z, z1 = z / p.nz2, z2 = z - z1 * p.nz2;
    // XCD-aware tile order: an XCD's L2 sees a contiguous run of tile ids.  With n fastest the run shares A panels and walks ALL of W
    // (every XCD fetches the whole weight: 8 W + A bytes in total); with m fastest it shares W panels and walks all of A (8 A + W).  The
    // bigger operand is the one to fetch once: m fastest whenever the weight is the larger operand (N > M: the T3 prefill projections,
    // where the n-fastest order measured a 3.1x over-fetch, profiles/r02_t3_eager_pmc_FETCH_SIZE.csv).
    const int tile = cbx_xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const bool mfast = p.N > p.M && p.nz1 * p.nz2 == 1;
    const int n0 = (mfast ? tile / gridDim.y : tile % gridDim.x) * BN, m0 = (mfast ? tile % gridDim.y : tile / gridDim.x) * BM;

    const float* __restrict__ Ab = p.A + (long)z1 * p.a_s1 + (long)z2 * p.a_s2;
    const float* __restrict__ Wb = p.W + (long)z1 * p.w_s1 + (long)z2 * p.w_s2;
    const int lim = p.lens ? min(p.Tin, p.lens[z1]) : p.Tin;
    const int K = p.K;

    // ---- per-thread loader state.  Everything that does not change along K is folded into base pointers; the K walk
    // itself is incremental (column offset + tap counter), so a tile costs a handful of integer ops per load.
    const int a_c4 = (tid % TPR) * 4;
    const float* a_ptr[A_IT];   // row pointer at tap 0, column a_c4
    int a_row[A_IT];            // input row at tap 0 (may be negative: left padding)
    bool a_ok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + (tid / TPR) + RP * i;
        a_ok[i] = m < p.M && (!A_PART || (tid / TPR) + RP * i < BM);
        a_row[i] = m * p.stride - p.pad_left;
        a_ptr[i] = Ab + (long)a_row[i] * p.lda + a_c4;
    }
    const float* b_ptr[B_IT];
    bool b_ok[B_IT];
    if constexpr (!W_KN) {
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int n = n0 + (tid / TPR) + RP * i;
            b_ok[i] = n < p.N && (!B_PART || (tid / TPR) + RP * i < BN);
            b_ptr[i] = Wb + (long)(b_ok[i] ? n : 0) * p.ldw + a_c4;
        }
    }
    int ld_tap = 0, ld_c0 = 0;          // tap / channel offset of the NEXT tile to load
    long ld_aoff = 0;                   // = ld_tap*dil*lda + ld_c0
    const long tap_step = (long)p.dil * p.lda - p.Cin;
    f32x4 ra[A_IT], rb[B_IT];

    auto load_tiles = [&](int kt) {
        const int k0 = kt * BK;
        const int kk = k0 + a_c4;
        const bool kfull = (k0 + BK) <= K;  // wave-uniform: only the last tile of a ragged K takes the guarded path
        const int tap_rows = ld_tap * p.dil;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int rr = a_row[i] + tap_rows;
            bool ok = a_ok[i] && rr >= 0 && (kfull || kk < K);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (p.up > 1) {  // nearest-neighbour upsampled input (conformer Upsample1D): row index is not linear in the tap
                const int row = rr / p.up;
                if (ok && row < lim) v = *reinterpret_cast<const f32x4*>(Ab + (long)row * p.lda + ld_c0 + a_c4);
            } else {
                if (ok && rr < lim) v = *reinterpret_cast<const f32x4*>(a_ptr[i] + ld_aoff);
            }
            ra[i] = v;
        }
        if constexpr (!W_KN) {
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (b_ok[i] && (kfull || kk < K)) v = *reinterpret_cast<const f32x4*>(b_ptr[i] + k0);
                rb[i] = v;
            }
        } else {
            constexpr int PER = BN / 4;  // float4 per k row
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                int idx = tid + NT * i;
                int kr = idx / PER, c4 = (idx - kr * PER) * 4;
                int k = k0 + kr, n = n0 + c4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k < K && kr < BK) {
                    const float* src = Wb + (long)k * p.ldw + n;
                    if (n + 3 < p.N) {
                        v = *reinterpret_cast<const f32x4*>(src);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < p.N) v[e] = src[e];
                    }
                }
                rb[i] = v;
            }
        }
        // advance the K walk
        ld_c0 += BK;
        ld_aoff += BK;
        if (p.taps > 1 && ld_c0 >= p.Cin) {
            ld_c0 = 0;
            ld_tap += 1;
            ld_aoff += tap_step;
        }
    };

    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            if (!A_PART || (tid / TPR) + RP * i < BM)
                *reinterpret_cast<f32x4*>(&As[buf][((tid / TPR) + RP * i) * LDS_LD + a_c4]) = ra[i];
        if constexpr (!W_KN) {
#pragma unroll
            for (int i = 0; i < B_IT; ++i)
                if (!B_PART || (tid / TPR) + RP * i < BN)
                    *reinterpret_cast<f32x4*>(&Bs[buf][((tid / TPR) + RP * i) * LDS_LD + a_c4]) = rb[i];
        } else {
            constexpr int PER = BN / 4;
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                int idx = tid + NT * i;
                int kr = idx / PER, c4 = (idx - kr * PER) * 4;
                if (kr < BK) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) Bs[buf][(c4 + e) * LDS_LD + kr] = rb[i][e];
                }
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (K + BK - 1) / BK;
    const int lr = lane & 31, lh = lane >> 5;
    auto compute = [&](int cur) {
        const float* as = &As[cur][(wm * WM + lr) * LDS_LD + 4 * lh];
        const float* bs = &Bs[cur][(wn * WN + lr) * LDS_LD + 4 * lh];
#pragma unroll
        for (int kb = 0; kb < BK / 8; ++kb) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(as + i * 32 * LDS_LD + kb * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(bs + j * 32 * LDS_LD + kb * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
    };

    if constexpr (LD != 0) {
        constexpr int OOB = (int)0x80000000;  // byte offset >= num_records: the load returns 0
        const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ab), 0, OOB, 0x00020000);
        const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wb), 0, OOB, 0x00020000);
        int a_vo[A_IT], a_rr[A_IT], a_lin[A_IT], b_vo[B_IT];
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            a_rr[i] = a_ok[i] ? a_row[i] : OOB;                 // input row of tap 0 (negative: never valid)
            a_lin[i] = (a_row[i] * (int)p.lda + a_c4) * 4;
            a_vo[i] = (a_rr[i] >= 0 && a_rr[i] < lim) ? a_lin[i] : OOB;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) b_vo[i] = b_ok[i] ? ((n0 + (tid / TPR) + RP * i) * (int)p.ldw + a_c4) * 4 : OOB;
        const int tap_bytes = p.dil * (int)p.lda * 4;
        const bool conv = p.taps > 1;
        int f_kt = 0, f_c0 = 0;
        auto load_fast = [&](f32x4(&fa)[A_IT], f32x4(&fb)[B_IT]) {
            const int pe = f_kt < nk ? 0 : OOB;  // the unrolled loop touches up to three tiles past the end: zeros
#pragma unroll
            for (int i = 0; i < A_IT; ++i)
                fa[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_vo[i] | pe, f_c0 * 4, 0));
#pragma unroll
            for (int i = 0; i < B_IT; ++i)
                fb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_vo[i] | pe, f_kt * (BK * 4), 0));
            f_kt += 1;
            f_c0 += BK;
            // next tap (convolutions; a Linear has Cin = K and never wraps inside the loop): rows move by dil, validity is re-derived,
            // branch-free so that the K loop stays one basic block
            const bool wrap = conv && f_c0 >= p.Cin;
            f_c0 = wrap ? 0 : f_c0;
            const int dr = wrap ? p.dil : 0, db = wrap ? tap_bytes : 0;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                a_rr[i] += dr;
                a_lin[i] += db;
                a_vo[i] = (a_rr[i] >= 0 && a_rr[i] < lim) ? a_lin[i] : OOB;
            }
        };
        auto store_fast = [&](int buf, const f32x4(&fa)[A_IT], const f32x4(&fb)[B_IT]) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i)
                if (!A_PART || (tid / TPR) + RP * i < BM)
                    *reinterpret_cast<f32x4*>(&As[buf][((tid / TPR) + RP * i) * LDS_LD + a_c4]) = fa[i];
#pragma unroll
            for (int i = 0; i < B_IT; ++i)
                if (!B_PART || (tid / TPR) + RP * i < BN)
                    *reinterpret_cast<f32x4*>(&Bs[buf][((tid / TPR) + RP * i) * LDS_LD + a_c4]) = fb[i];
        };
        f32x4 fa[2][A_IT], fb[2][B_IT];
        load_fast(fa[0], fb[0]);
        store_fast(0, fa[0], fb[0]);
        load_fast(fa[1], fb[1]);
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            load_fast(fa[0], fb[0]);        // tile kt+2
            compute(0);                     // tile kt
            store_fast(1, fa[1], fb[1]);    // tile kt+1
            __syncthreads();
            load_fast(fa[1], fb[1]);        // tile kt+3
            compute(1);                     // tile kt+1 (all zero when nk is odd and this is past the end)
            store_fast(0, fa[0], fb[0]);    // tile kt+2
            __syncthreads();
        }
    } else {
        load_tiles(0);
        store_tiles(0);
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load_tiles(kt + 1);
            compute(cur);
            if (kt + 1 < nk) store_tiles(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }

    // ---- epilogue.  C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* Cb = p.C + (long)z1 * p.c_s1 + (long)z2 * p.c_s2;
    const float* Rb = p.R ? p.R + (long)z1 * p.r_s1 + (long)z2 * p.r_s2 : nullptr;
    float* C2b = p.C2 ? p.C2 + (long)z1 * p.c2_s1 + (long)z2 * p.c2_s2 : nullptr;

    if (p.swiglu) {
        if constexpr (TN == 2) {
            const int oc = (n0 + wn * WN) / 2 + lr;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m < p.M && oc < p.N / 2) {
                        float g = acc[i][0][r], u = acc[i][1][r];
                        Cb[(long)m * p.ldc + oc] = (g / (1.0f + __expf(-g))) * u;
                    }
                }
        }
        return;
    }

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + lr;
        if (n >= p.N) continue;
        const float bia = p.bias ? p.bias[n] : 0.f;
        const float a1 = p.act1_param ? p.act1_param[n] : 0.f;
        const float a2 = p.act2_param ? p.act2_param[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bia;
                v = cbx_act(v, p.act1, p.act1_slope, a1);
                if (Rb) v += Rb[(long)m * p.ldr + n];
                v *= p.alpha;
                float* dst = Cb + (long)m * p.ldc + n;
                if (p.beta != 0.f) v += p.beta * *dst;
                *dst = v;
                if (C2b) C2b[(long)m * p.ldc2 + n] = cbx_act(v, p.act2, p.act2_slope, a2);
            }
    }
}

template <int BM, int BN, int WARPS_M, int WARPS_N, bool W_KN, int BK = 16, int LD = 0>
int launch(const cbx_gemm_t& p, hipStream_t st) {
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, p.nz1 * p.nz2);
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WARPS_M, WARPS_N, W_KN, BK, LD>), grid, dim3(WARPS_M * WARPS_N * 64), 0, st, p);
    return cbx_check_launch("gemm_f32");
}

// can the buffer-load loader serve this call?  (K tiles that never straddle the end of K or a conv tap, 31-bit byte offsets)
bool fast_loader_ok(const cbx_gemm_t& p, int bk) {
    return !p.w_kn && p.up == 1 && p.K % bk == 0 && p.Cin % bk == 0 && (long)(p.Tin + 1) * p.lda * 4 < 0x7fffffffL &&
           (long)(p.N + 256) * p.ldw * 4 < 0x7fffffffL && (long)p.pad_left * p.lda * 4 < 0x3fffffffL;
}

}  // namespace

int cbx_gemm_split_dispatch(const cbx_gemm_t& p, int planes, hipStream_t st);  // gemm_split.hip

extern "C" int cbx_gemm_f32(const cbx_gemm_t* pp, void* stream) {
    cbx_gemm_t p = *pp;
    hipStream_t st = (hipStream_t)stream;
    if (p.nz1 < 1) p.nz1 = 1;
    if (p.nz2 < 1) p.nz2 = 1;
    if (p.taps < 1) p.taps = 1;
    if (p.stride < 1) p.stride = 1;
    if (p.up < 1) p.up = 1;
    if (p.taps == 1 && p.Cin == 0) p.Cin = p.K;
    if (p.Tin == 0) p.Tin = p.M;
    CBX_REQUIRE(p.A && p.W && p.C, "gemm: null operand");
    CBX_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm: bad shape M=%d N=%d K=%d", p.M, p.N, p.K);
    CBX_REQUIRE(p.K == p.taps * p.Cin, "gemm: K=%d != taps*Cin=%d*%d", p.K, p.taps, p.Cin);
    CBX_REQUIRE(p.taps == 1 || p.Cin % 16 == 0, "gemm: conv needs Cin %% 16 == 0 (Cin=%d)", p.Cin);
    CBX_REQUIRE(p.lda % 4 == 0 && p.a_s1 % 4 == 0 && p.a_s2 % 4 == 0 && ((uintptr_t)p.A & 15) == 0,
                "gemm: A must be 16-byte aligned (lda=%ld)", p.lda);
    CBX_REQUIRE(p.ldw % 4 == 0 && p.w_s1 % 4 == 0 && p.w_s2 % 4 == 0 && ((uintptr_t)p.W & 15) == 0,
                "gemm: W must be 16-byte aligned (ldw=%ld)", p.ldw);
    CBX_REQUIRE((long)p.nz1 * p.nz2 <= 65535, "gemm: too many batches");
    CBX_REQUIRE(!p.swiglu || (p.N % 64 == 0 && !p.bias && !p.R && !p.C2 && !p.w_kn), "gemm: bad swiglu config");
    CBX_REQUIRE(!p.ln_stats || (!p.w_kn && !p.swiglu && p.M > 32 && (p.precision ? p.precision : 0) == 16),
                "gemm: ln_stats (LayerNorm folded into A) is served by the precision-16 split kernel only, M > 32");
    if (p.w_kn) {
        if (p.N <= 64) return launch<128, 64, 2, 2, true>(p, st);
        return launch<128, 128, 2, 2, true>(p, st);
    }
    if (p.swiglu) {
        if (p.M <= 32) return launch<32, 256, 1, 4, false>(p, st);
        if (fast_loader_ok(p, 16)) return launch<128, 128, 2, 2, false, 16, 1>(p, st);
        return launch<128, 128, 2, 2, false>(p, st);
    }
    if (p.M <= 32) return launch<32, 128, 1, 4, false>(p, st);
    {
        // precision 0 = exact; 1 = exact fp32 MFMA; 3 / 6 = fp32 rebuilt from
        // 3 / 6 bf16 plane products on the 16x faster bf16 matrix cores, 16 = from 3 fp16 plane products (gemm_split.hip)
        const int prec = p.precision ? p.precision : 1;
        CBX_REQUIRE(prec == 1 || prec == 3 || prec == 6 || prec == 16, "gemm: precision must be 0, 1, 3, 6 or 16 (got %d)", prec);
        if (prec != 1) {
            int rc = cbx_gemm_split_dispatch(p, prec == 16 ? 16 : prec == 3 ? 2 : 3, st);
            if (rc != -1) return rc;
            CBX_REQUIRE(!p.ln_stats, "gemm: ln_stats on a shape the split kernel does not serve");
        }
    }
    if (p.N <= 64) return launch<128, 64, 2, 2, false>(p, st);
    // measured on the CFM / HiFT shapes (bench.py, MI355X): 64x64 tiles 492 ms per flow pass, 128x64 543, 64x128 536, 128x128 803 -- the
    // single-stage pipeline needs many co-resident waves to hide its load->LDS->barrier latency; 8 waves x (32x32): 81 TF/s on the bench mix
    // vs 77 for 64x64 (4 waves).  (Rounds 1-2 A/B'd six more tile forms through an environment knob; the losers are gone with it.)
    // a handful of workgroups walking a long K (the prefill of the GPT-2 backbones at batch 1: 65 - 441 rows, N = 1024, K = 4096: 16 - 64 workgroups x 256 K tiles of 16)
    // is bound by the latency of a K step, not by the matrix rate: 64-wide K tiles take a quarter of the steps.  Same k order per output element: bit-identical
    // (round 6, profiles/r06_at_*.log)
    const long wgs = (long)((p.M + 127) / 128) * ((p.N + 63) / 64) * p.nz1 * p.nz2;
    if (wgs <= 128 && p.K >= 1024 && fast_loader_ok(p, 64)) return launch<128, 64, 4, 2, false, 64, 1>(p, st);
    if (fast_loader_ok(p, 16)) return launch<128, 64, 4, 2, false, 16, 1>(p, st);
    return launch<128, 64, 4, 2, false>(p, st);
}
