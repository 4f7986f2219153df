// EXPERIMENTAL (opt-in via T3Engine(fused_norm=True) / CBX_T3_FUSED=1; written after the round-1 GPU budget was spent, so not yet
// run on hardware): decode GEMV with the residual add + split-K reduce + RMSNorm of its input folded into its prologue.
//
//   h[m][k]   = res[m][k] + sum_{j < ks_in} part[j][m][k]              (fixed order j = 0, 1, ...: deterministic)
//   out[m][n] = rstd[m] * sum_k (h[m][k] * norm_w[k]) * W[n][k],        rstd[m] = rsqrt(mean_k h[m][k]^2 + eps)
//
// i.e. out = RMSNorm(h) W^T with the per-row scalar rstd factored out of the contraction and applied in the epilogue, so the
// MFMAs do not wait for the row statistic.  Every workgroup of a decode GEMV already reads the whole 16 x K operand; here it reads
// the residual and the producer's (<= 2) split-K partials instead, builds h * norm_w once in LDS (all 8 waves cooperate, 512-B
// coalesced row segments), and feeds the MFMAs from LDS.  Workgroup 0 writes h to the ping-pong residual buffer.  This removes the
// two add_rmsnorm launches of a Llama layer (7 -> 5 dependent launches per layer) without atomics or fences.
//
// Weight streaming is identical to gemv_kernel (gemv_decode.hip): lanes (c = lane&15, q = lane>>4) stream 32 contiguous bytes of
// row n0+c per 32-deep K block; 8 waves split K; fixed-order reduction through LDS.
#include "cbx_common.h"

namespace {

constexpr int GN_NW = 8;

template <int MT, bool SWIGLU, int KSIN>
__global__ __launch_bounds__(GN_NW * 64) void gemv_norm_kernel(const cbx_gemv_norm_t p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = p.K, LDH = K + 4;             // +4 floats: rows 16 B apart modulo 256 B (2-way worst case on the b128 reads)
    float* hs = smem;                           // [MT*16][LDH]   h * norm_w
    float* ssq = hs + MT * 16 * LDH;            // [MT*16] sum_k h^2
    float* red = hs;                            // [(SWIGLU ? 2 : 1)][NW][MT][256] -- reuses the operand image after the MFMAs

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int kper = K / GN_NW;                 // <= 128
    const int kbeg = w * kper;
    const int nit = kper / 32;                  // <= 4

    // ---- 0. this wave's weight stream: every K block is requested before anything else (HBM latency overlaps the prologue)
    long wrow;
    bool wok;
    if constexpr (SWIGLU) {
        const int f = n0 + c;
        wok = f < p.N;
        wrow = (long)(f >> 5) * 64 + (f & 31);
    } else {
        wok = (n0 + c) < p.N;
        wrow = n0 + c;
    }
    const float* wp = p.W + (wok ? wrow : 0) * p.ldw + kbeg + 8 * q;
    const float* wp2 = wp + 32 * p.ldw;
    f32x4 wv[4][2], uv[4][2];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int off = d < nit ? d * 32 : 0;
        wv[d][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp + off));
        wv[d][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp + off + 4));
        if constexpr (SWIGLU) {
            uv[d][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp2 + off));
            uv[d][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp2 + off + 4));
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- 1. prologue: 32 threads per row (thread j covers float4 columns j, j+32, ...), rows tid>>5 (+16 for MT = 2)
    const int prow = tid >> 5, pj = tid & 31;
    const int nc4 = K >> 7;                     // float4 per thread per row = K / (32*4)  (<= 8)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = t * 16 + prow;
        const bool live = m < p.M;
        const float* rp = p.res + (long)(live ? m : 0) * p.ldr;
        const float* pp = KSIN > 0 ? p.part + (long)(live ? m : 0) * p.ldp : nullptr;
        f32x4 hv[8];
        // all loads of the row slice first (unconditional, clamped), then the adds in fixed order
        f32x4 p0[8], p1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c4 = (i < nc4 ? i : 0) * 32 + pj;
            hv[i] = *reinterpret_cast<const f32x4*>(rp + c4 * 4);
            if constexpr (KSIN > 0) p0[i] = *reinterpret_cast<const f32x4*>(pp + c4 * 4);
            if constexpr (KSIN > 1) p1[i] = *reinterpret_cast<const f32x4*>(pp + p.part_stride + c4 * 4);
        }
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < nc4) {
                const int c4 = i * 32 + pj;
                f32x4 h = hv[i];
                if constexpr (KSIN > 0) h += p0[i];
                if constexpr (KSIN > 1) h += p1[i];
                if (!live) h = zero4;
                ss += h[0] * h[0] + h[1] * h[1] + h[2] * h[2] + h[3] * h[3];
                const f32x4 nw4 = *reinterpret_cast<const f32x4*>(p.norm_w + c4 * 4);
                *reinterpret_cast<f32x4*>(&hs[(t * 16 + prow) * LDH + c4 * 4]) = h * nw4;
                if (p.res_out && blockIdx.x == 0 && live) *reinterpret_cast<f32x4*>(p.res_out + (long)m * p.ldro + c4 * 4) = h;
            }
        }
        // row statistic: the 32 threads of a row are one half-wave
        ss += __shfl_xor(ss, 16);
        ss += __shfl_xor(ss, 8);
        ss += __shfl_xor(ss, 4);
        ss += __shfl_xor(ss, 2);
        ss += __shfl_xor(ss, 1);
        if (pj == 0) ssq[t * 16 + prow] = ss;
    }
    __syncthreads();

    // ---- 2. MFMAs: x operand from LDS (same (c, q) k-permutation as the streamed operand), counted waits on the weights
    f32x4 acc[MT], acc2[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        acc[t] = zero4;
        acc2[t] = zero4;
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const bool won = (d < nit) && wok;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 wq = won ? wv[d][h] : zero4;
            f32x4 uq = zero4;
            if constexpr (SWIGLU) uq = won ? uv[d][h] : zero4;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int koff = kbeg + (d < nit ? d * 32 : 0) + 8 * q + 4 * h;
                const f32x4 xq = *reinterpret_cast<const f32x4*>(&hs[(t * 16 + c) * LDH + koff]);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xq[s], wq[s], acc[t], 0, 0, 0);
                    if constexpr (SWIGLU) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xq[s], uq[s], acc2[t], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- 3. fixed-order reduction over the 8 K-slices, rstd, epilogue.  D map: row = q*4 + r, col = c.
    __syncthreads();  // every wave is done reading the operand image that `red` overlays
    float* r1 = red;
    float* r2 = red + GN_NW * MT * 256;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            r1[(w * MT + t) * 256 + (q * 4 + r) * 16 + c] = acc[t][r];
            if constexpr (SWIGLU) r2[(w * MT + t) * 256 + (q * 4 + r) * 16 + c] = acc2[t][r];
        }
    __syncthreads();
    for (int e = tid; e < MT * 256; e += GN_NW * 64) {
        const int t = e >> 8, rc = e & 255, row = rc >> 4, col = rc & 15;
        const int m = t * 16 + row, n = n0 + col;
        if (m >= p.M || n >= p.N) continue;
        float v = 0.f, v2 = 0.f;
#pragma unroll
        for (int ww = 0; ww < GN_NW; ++ww) {
            v += r1[(ww * MT + t) * 256 + rc];
            if constexpr (SWIGLU) v2 += r2[(ww * MT + t) * 256 + rc];
        }
        const float rstd = rsqrtf(ssq[m] / (float)K + p.eps);
        v *= rstd;
        if constexpr (SWIGLU) {
            v2 *= rstd;
            v = (v / (1.0f + __expf(-v))) * v2;
        }
        p.out[(long)m * p.ldo + n] = v;
    }
}

template <int MT, bool SWIGLU, int KSIN>
int launch_gn(const cbx_gemv_norm_t& p, hipStream_t st) {
    static_assert((SWIGLU ? 2 : 1) * GN_NW * 256 <= 16 * (256 + 4), "the reduction buffer must fit inside the operand image");
    const size_t lds = ((size_t)MT * 16 * (p.K + 4) + MT * 16) * sizeof(float);
    auto kern = gemv_norm_kernel<MT, SWIGLU, KSIN>;
    static size_t configured = 0;  // > 64 KiB of dynamic LDS has to be opted into (once per kernel and size)
    if (configured < lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return cbx_set_error((int)e, "gemv_norm: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
        configured = lds;
    }
    hipLaunchKernelGGL(kern, dim3((p.N + 15) / 16), dim3(GN_NW * 64), lds, st, p);
    return cbx_check_launch("gemv_norm");
}

template <int MT, bool SWIGLU>
int launch_gn_ks(const cbx_gemv_norm_t& p, hipStream_t st) {
    switch (p.ks_in) {
        case 0: return launch_gn<MT, SWIGLU, 0>(p, st);
        case 1: return launch_gn<MT, SWIGLU, 1>(p, st);
        default: return launch_gn<MT, SWIGLU, 2>(p, st);
    }
}

}  // namespace

extern "C" int cbx_gemv_norm_f32(const cbx_gemv_norm_t* pp, void* stream) {
    const cbx_gemv_norm_t p = *pp;
    hipStream_t st = (hipStream_t)stream;
    CBX_REQUIRE(p.res && p.norm_w && p.W && p.out, "gemv_norm: null operand");
    CBX_REQUIRE(p.M > 0 && p.M <= 32 && p.N > 0, "gemv_norm: M=%d must be in 1..32", p.M);
    CBX_REQUIRE(p.K >= 256 && p.K <= 1024 && p.K % 256 == 0, "gemv_norm: K=%d must be 256, 512, 768 or 1024", p.K);
    CBX_REQUIRE(p.ks_in >= 0 && p.ks_in <= 2 && (p.ks_in == 0 || p.part), "gemv_norm: ks_in=%d must be 0..2 (with part)", p.ks_in);
    CBX_REQUIRE(p.ldr % 4 == 0 && p.ldw % 4 == 0 && (p.ks_in == 0 || (p.ldp % 4 == 0 && p.part_stride % 4 == 0)) &&
                    (!p.res_out || p.ldro % 4 == 0), "gemv_norm: strides must be multiples of 4 floats");
    CBX_REQUIRE(p.res_out != p.res, "gemv_norm: res_out must not alias res (other workgroups still read it)");
    CBX_REQUIRE(!p.swiglu || p.N % 32 == 0, "gemv_norm: swiglu needs N %% 32 == 0");
    if (p.swiglu) return p.M <= 16 ? launch_gn_ks<1, true>(p, st) : launch_gn_ks<2, true>(p, st);
    return p.M <= 16 ? launch_gn_ks<1, false>(p, st) : launch_gn_ks<2, false>(p, st);
}
