// Skinny-M weight-streaming GEMM for autoregressive decode (M = 2*B rows <= 64): out[M][N] = x[M][K] W[N][K]^T.
//
// Roofline: HBM.  Arithmetic intensity is ~M/2 FLOP per weight byte (fp32), far below the MFMA ridge, so the kernel is
// organised around streaming W exactly once at full width and keeping every CU busy:
//   * a wave owns 16 output columns and a K slice; lanes (c = lane&15, q = lane>>4) read 32 contiguous bytes of row
//     n0+c per 32-deep K block -> each W row is consumed in full 128-B lines, 2 KiB per wave-iteration, several
//     iterations in flight (register prefetch), no LDS round trip for the streamed operand (it is never reused);
//   * x (<= 64 rows, L2-resident) is read with the same (c, q) pattern, which is exactly the A-operand layout of
//     v_mfma_f32_16x16x4_f32 under a k-permutation shared with B, so the 16x16 tile of dot products costs 8 MFMAs
//     per 32-deep block instead of 512 lane-FMAs + a cross-lane reduction;
//   * K is split over the waves of a workgroup (reduced through LDS in fixed order) and optionally over workgroups
//     (partials reduced by the consumer kernel, cbx_add_rmsnorm_f32) -- deterministic, no atomics.
#include <stdlib.h>
#include "cbx_common.h"

namespace {
CBX_TRC_TU

// PK: W is the lane-ordered packed image (cbx.h "packed GEMV weight layout"): the 2 KiB of a (16-row tile, 32-deep K block) are stored
// as [h][lane][4 floats], so every wave-level load instruction reads 1 KiB of CONTIGUOUS memory (8 full 128-B lines) instead of
// 16-B pieces of 16 different rows.  XPK: the same layout for the x operand (written that way by the producing kernel).
// RMS: LlamaRMSNorm of the x operand folded in: the lanes multiply their x values by norm_w[k] on the way to the MFMA, accumulate
// sum_k x^2 per row as a by-product (every workgroup reads whole rows), and the per-row rstd -- a scalar that factors out of the
// contraction -- is applied in the epilogue:  out[m][n] = rstd[m] * sum_k (x[m][k] * norm_w[k]) * W[n][k].  (ksplit must be 1.)
// NP > 0 (RMS variants, <= 16 rows): the x operand is x + sum_{j < NP} xpart[j] -- the split-K partial images of the PRODUCING
// projection are reduced (fixed order) on the way to the MFMA instead of by a separate kernel; workgroup 0 also writes the sum
// (the new residual stream) to x_out.  Every workgroup re-reads the NP + 1 images from L2: NP * 64 KiB extra per workgroup,
// which is cheaper than a dependent launch (~1.8 us boundary + a cross-XCD round trip) only for small NP.
// WB: the packed weight image holds bf16 (cbx_pack_gemv_weight_bf16): half the streamed bytes; a lane's 8 k of a block are ONE 16-byte
// load, widened to fp32 by a shift (exact), the products stay bf16-weight x fp32-activation in the exact fp32 MFMA.  Opt-in numerics
// (the weights are rounded): not the parity path.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 bf16x4_widen(const u32x4 u, int h) {
    const unsigned a = h ? u[2] : u[0], b = h ? u[3] : u[1];
    return f32x4{__uint_as_float(a << 16), __uint_as_float(a & 0xffff0000u), __uint_as_float(b << 16), __uint_as_float(b & 0xffff0000u)};
}

// cbx_gemv_t.half_tile -> output columns per workgroup (0: 16; 1 or 8: 8; 12; 4)
__host__ __device__ __forceinline__ int gemv_tile_cols(int half_tile) { return half_tile == 0 ? 16 : half_tile == 1 ? 8 : half_tile; }

// D8: eight K blocks (instead of four) requested before the first MFMA -- for a wave whose K slice is >= 256 deep (the down projection
// without split-K partials: K = 4096 over 16 waves) the whole slice is then ONE batch of loads instead of two dependent ones.
// D2 (round 5, cbx_gemv_t.flags & CBX_GEMV_SHALLOW): TWO K blocks per batch -- the SwiGLU launch then needs <= 128 VGPRs instead of 162, so its workgroups (2 waves per
// SIMD) fit on a CU beside a workgroup of another stream that leaves half of the register file free (the throughput schedule: profiles/r05_overlap_*).
// Same loads, same MFMA order: bit-identical results.
template <int MT, int NW, bool SWIGLU, bool PK, bool XPK, bool RMS, int NP, bool WB = false, bool D8 = false, bool D2 = false>
__global__ __launch_bounds__(NW * 64) void gemv_kernel(const cbx_gemv_t p) {
    __builtin_amdgcn_s_setprio(3);  // decode-step kernels are latency-bound and issue little: beside a co-resident workgroup of another stream (the throughput schedule, profiles/r05_overlap_*) their waves go first at the SIMD's issue arbiter; alone on the CU it changes nothing
    static_assert(NP == 0 || (RMS && MT == 1), "partial-sum operand: RMS variant, one row tile");
    static_assert(!WB || (PK && XPK), "bf16 weights: packed operands only");
    __shared__ __attribute__((aligned(16))) float red[(SWIGLU ? 2 : 1) * NW * MT * 256];
    __shared__ float ssq[RMS ? NW * MT * 16 : 1];
    __shared__ float ssx[RMS ? NW * MT * 16 : 1];  // row sums (LayerNorm form only)
    CBX_TRC_DECL;
    CBX_TRC_STAMP(0);  // entry
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    // tc = output columns per workgroup: 16, or the narrow tiles 12 / 8 / 4 (cbx_gemv_t.half_tile) that give a projection 4/3, 2 or 4
    // times the workgroups (lanes c >= tc idle in the B operand): 8 for the two N = 1024 projections, 12 puts q/k/v (N = 3072) and 4 the
    // o / down projections on exactly 256 workgroups
    const int tc = PK ? gemv_tile_cols(p.half_tile) : 16;
    const int n0 = blockIdx.x * tc, ks = blockIdx.y;
    const int kper = p.K / (p.ksplit * NW);
    const int kbeg = (ks * NW + w) * kper;
    const int nit = kper / 32;

    // row of W streamed by this lane (swiglu: gate row, the matching up row is 32 rows further in the packed image)
    long wrow;
    bool wok;
    const float *wp, *wp2;
    // floats between consecutive 32-deep K blocks of this lane's stream / between the two 16-B halves of a block; the bf16 image is half
    // as large (one 16-byte load per block: 8 bf16 per lane)
    const int WBLK = PK ? (32 * tc) / (WB ? 2 : 1) : 32;
    const int WHALF = PK ? 16 * tc : 4;
    if constexpr (PK) {
        // packed image: tile-major [tile][K/32][2][64 lanes][4]; swiglu: feature tile f -> tiles 2f (gate), 2f+1 (up); N is padded
        // to whole tiles by the packer, so every load is in range
        const long kb = p.K >> 5;
        const long tile = SWIGLU ? 2L * blockIdx.x : (long)blockIdx.x;
        wok = c < tc;
        const int cl = wok ? c : tc - 1;  // idle lanes re-read the tile's last row (loads are unconditional)
        if constexpr (WB) {  // [tile][K/32][lanes][8 bf16] = 4 floats per lane per block
            wp = p.W + (tile * kb + (kbeg >> 5)) * (16 * tc) + (q * tc + cl) * 4;
            wp2 = wp + kb * 256;
        } else {
            wp = p.W + (tile * kb + (kbeg >> 5)) * (32 * tc) + (q * tc + cl) * 4;
            wp2 = wp + kb * 512;
        }
    } else {
        if constexpr (SWIGLU) {
            const int f = n0 + c;  // feature index
            wok = f < p.N;
            wrow = (long)(f >> 5) * 64 + (f & 31);
        } else {
            wok = (n0 + c) < p.N;
            wrow = n0 + c;
        }
        wp = p.W + (wok ? wrow : 0) * p.ldw + kbeg + 8 * q;
        wp2 = wp + 32 * p.ldw;
    }
    const float* xp[MT];
    bool xok[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int m = t * 16 + c;
        xok[t] = m < p.M;
        if constexpr (XPK)  // packed x: [row tile][K/32][2][64][4], rows padded to whole tiles (pad rows hold finite values).
            // A lane whose row is past M reads ROW 0's 16 bytes of its k group instead of its own pad row (its value is zeroed below either way):
            // a wave-level load then touches 64 M bytes instead of 1 KiB -- at M = 1 (Turbo / Nano at batch 1: profiles/r04_decode_launch_timeline_turbo.txt)
            // the x operand and the partial images cost 1/16 of the lines, and the packed images of small batches stop being "16 rows wide".
            xp[t] = p.x + ((long)t * (p.K >> 5) + (kbeg >> 5)) * 512 + (xok[t] ? lane : (lane & 48)) * 4;
        else
            xp[t] = p.x + (long)(xok[t] ? m : 0) * p.ldx + kbeg + 8 * q;
    }

    const float* nwp = RMS ? p.norm_w + kbeg + 8 * q : nullptr;  // this lane's k indices: kbeg + 32*blk + 8*q + 4*h + s
    float ss[MT], sx[MT];
    f32x4 acc[MT], acc2[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        ss[t] = 0.f;
        sx[t] = 0.f;
    }
    // ---- epilogue operands requested FIRST (opt-in: cbx_gemv_t.flags & CBX_GEMV_PRE_EPI).  The element(s) a thread finishes after the reduction are known now; its
    // residual, bias and LayerNorm-fold constants do not depend on the contraction, so their loads go out with the first weight batch instead
    // of after the LDS reduction -- where each is a dependent global round trip (~1 us) on the critical path of a launch that lasts 5-9 us.
    // Same values, same order of the additions: results unchanged bit for bit.  (res may alias out: a thread reads exactly the element it
    // writes.)
    const bool PRE = (p.flags & CBX_GEMV_PRE_EPI) != 0;  // uniform (kernel argument)
    constexpr int EIT = (MT * 256 + NW * 64 - 1) / (NW * 64);
    float e_res[EIT], e_bias[EIT], e_cw[EIT], e_cb[EIT];
    long e_o[EIT];
    int e_n[EIT];
#pragma unroll
    for (int j = 0; j < EIT; ++j) {
        const int e = tid + j * NW * 64;
        const int t = e >> 8, rc = e & 255, row = rc >> 4, col = rc & 15;
        // threads without an element (idle waves, rows >= M, columns past the tile / N) address this workgroup's element (0, n0): the loads
        // below stay unconditional per lane (a lane-predicated load would put a wait in front of the weight stream)
        const bool ok = e < MT * 256 && (t * 16 + row) < p.M && (n0 + col) < p.N && col < tc;
        const int tt = ok ? t : 0, rr = ok ? row : 0, n = ok ? n0 + col : n0;
        if (p.out_packed)  // the consumer's lane-ordered operand layout (its K = this N): see cbx.h
            e_o[j] = (long)ks * p.part_stride + (((long)tt * (p.N >> 5) + (n >> 5)) * 2 + ((n >> 2) & 1)) * 256 + ((((n >> 3) & 3) << 4) + rr) * 4 + (n & 3);
        else
            e_o[j] = (long)ks * p.part_stride + (long)(tt * 16 + rr) * p.ldo + n;
        e_n[j] = n;
        e_res[j] = e_bias[j] = e_cw[j] = e_cb[j] = 0.f;
    }
    if (PRE && p.res) {  // uniform branches (kernel arguments)
#pragma unroll
        for (int j = 0; j < EIT; ++j) e_res[j] = p.res[e_o[j]];
    }
    if constexpr (!SWIGLU) {
        if (PRE && p.bias && ks == 0) {
#pragma unroll
            for (int j = 0; j < EIT; ++j) e_bias[j] = p.bias[e_n[j]];
        }
    }
    if constexpr (RMS) {
        if (PRE && p.ln_cw) {
#pragma unroll
            for (int j = 0; j < EIT; ++j) e_cw[j] = p.ln_cw[e_n[j]], e_cb[j] = p.ln_cb[e_n[j]];
        }
    }
    constexpr int DEPTH = D8 ? 8 : (MT == 1 && !D2) ? 4 : 2;  // K blocks (2 KiB of W per wave each) issued before the first MFMA
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // Loads are UNCONDITIONAL (out-of-range lanes / K blocks re-read a valid address and are zeroed by a select): a predicated
    // load makes hipcc join all of them behind one vmcnt(0); unconditional ones get counted waits, so the MFMAs of K block d
    // start while blocks d+1.. are still in flight.
    for (int it0 = 0; it0 < nit; it0 += DEPTH) {
        f32x4 wv[DEPTH][2], uv[DEPTH][2], xv[DEPTH][MT][2], nv[DEPTH][2], pv[DEPTH][NP > 0 ? NP : 1][2];
        bool on[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            on[d] = (it0 + d) < nit;
            const int blk = on[d] ? (it0 + d) : 0;
            const int off = blk * WBLK;
            wv[d][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp + off));
            if constexpr (!WB) wv[d][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp + off + WHALF));
            if constexpr (SWIGLU) {
                uv[d][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp2 + off));
                if constexpr (!WB) uv[d][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp2 + off + WHALF));
            }
            const int xoff = blk * (XPK ? 512 : 32);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                xv[d][t][0] = *reinterpret_cast<const f32x4*>(xp[t] + xoff);
                xv[d][t][1] = *reinterpret_cast<const f32x4*>(xp[t] + xoff + (XPK ? 256 : 4));
            }
            if constexpr (RMS) {
                nv[d][0] = *reinterpret_cast<const f32x4*>(nwp + blk * 32);
                nv[d][1] = *reinterpret_cast<const f32x4*>(nwp + blk * 32 + 4);
            }
            if constexpr (NP > 0) {
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const float* pp = p.xpart + (long)j * p.xpart_stride + (xp[0] - p.x);
                    pv[d][j][0] = *reinterpret_cast<const f32x4*>(pp + xoff);
                    pv[d][j][1] = *reinterpret_cast<const f32x4*>(pp + xoff + 256);
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the issue order block by block, so block d's wait is vmcnt(later blocks)
        }
#ifdef CBX_TRACE
        if (it0 == 0) CBX_TRC_STAMP(1);  // first load batch issued
#endif
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const bool won = on[d] && wok;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 wq, uq = zero4;
                if constexpr (WB) {
                    wq = won ? bf16x4_widen(__builtin_bit_cast(u32x4, wv[d][0]), h) : zero4;
                    if constexpr (SWIGLU) uq = won ? bf16x4_widen(__builtin_bit_cast(u32x4, uv[d][0]), h) : zero4;
                } else {
                    wq = won ? wv[d][h] : zero4;
                    if constexpr (SWIGLU) uq = won ? uv[d][h] : zero4;
                }
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    f32x4 xq = xv[d][t][h];
                    if constexpr (NP > 0) {
#pragma unroll
                        for (int j = 0; j < NP; ++j) xq += pv[d][j][h];
                        if (p.x_out && blockIdx.x == 0 && on[d] && xok[0])  // the reduced residual stream, same packed address as x (pad rows stay as allocated: zero)
                            cbx_store_out4(p.x_out + (xp[0] - p.x) + (it0 + d) * 512 + h * 256, xq);
                    }
                    xq = (on[d] && xok[t]) ? xq : zero4;
                    if constexpr (RMS) {
                        ss[t] += (xq[0] * xq[0] + xq[1] * xq[1]) + (xq[2] * xq[2] + xq[3] * xq[3]);
                        sx[t] += (xq[0] + xq[1]) + (xq[2] + xq[3]);
                        xq *= nv[d][h];
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xq[s], wq[s], acc[t], 0, 0, 0);
                        if constexpr (SWIGLU) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xq[s], uq[s], acc2[t], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef CBX_TRACE
            if (it0 == 0 && d == 0) {  // first K block multiplied: its loads (weights, x, norm weights, partial images) have landed
                asm volatile("s_nop 0" ::"v"(acc[0][0]));
                CBX_TRC_STAMP(2);
            }
#endif
        }
    }
#ifdef CBX_TRACE
    asm volatile("s_nop 0" ::"v"(acc[0][0]));
    CBX_TRC_STAMP(3);  // K loop done (wave 0)
#endif

    // ---- fixed-order reduction over the NW K-slices of this workgroup.  D map: row = q*4 + r, col = c.
    float* r1 = red;
    float* r2 = red + NW * MT * 256;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            r1[(w * MT + t) * 256 + (q * 4 + r) * 16 + c] = acc[t][r];
            if constexpr (SWIGLU) r2[(w * MT + t) * 256 + (q * 4 + r) * 16 + c] = acc2[t][r];
        }
    if constexpr (RMS) {  // lanes (c, q = 0..3) hold the four q-parts of row c's sum over this wave's K slice
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            float v = ss[t];
            v += cbx_xor_lane<16>(v);
            v += cbx_xor_lane<32>(v);
            float u = sx[t];
            u += cbx_xor_lane<16>(u);
            u += cbx_xor_lane<32>(u);
            if (q == 0) {
                ssq[(w * MT + t) * 16 + c] = v;
                ssx[(w * MT + t) * 16 + c] = u;
            }
        }
    }
    __syncthreads();
    CBX_TRC_STAMP(4);  // every wave's partial tile is in LDS
#pragma unroll
    for (int j = 0; j < EIT; ++j) {
        const int e = tid + j * NW * 64;
        if (e >= MT * 256) break;
        const int t = e >> 8, rc = e & 255, row = rc >> 4, col = rc & 15;
        const int m = t * 16 + row, n = n0 + col;
        if (m >= p.M || n >= p.N || col >= tc) continue;
        float v = 0.f, v2 = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
            v += r1[(ww * MT + t) * 256 + rc];
            if constexpr (SWIGLU) v2 += r2[(ww * MT + t) * 256 + rc];
        }
        if constexpr (RMS) {
            float sq = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) sq += ssq[(ww * MT + t) * 16 + row];
            if (p.ln_cw) {  // LayerNorm form (GPT-2): y = (x - mean) rstd w + b  =>  out = rstd (acc - mean cw[n]) + cb[n]
                float su = 0.f;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) su += ssx[(ww * MT + t) * 16 + row];
                const float mean = su / (float)p.K;
                const float rstd = rsqrtf(fmaxf(sq / (float)p.K - mean * mean, 0.f) + p.eps);
                v = PRE ? rstd * (v - mean * e_cw[j]) + e_cb[j] : rstd * (v - mean * p.ln_cw[n]) + p.ln_cb[n];
            } else {
                const float rstd = rsqrtf(sq / (float)p.K + p.eps);
                v *= rstd;
                v2 *= rstd;
            }
        }
        if constexpr (SWIGLU) {
            v = (v / (1.0f + __expf(-v))) * v2;
        } else {
            if (p.bias && ks == 0) v += PRE ? e_bias[j] : p.bias[n];
            if (p.act) v = cbx_act(v, p.act, 0.f, 0.f);  // only meaningful with ksplit == 1
        }
        if (p.res) v += PRE ? e_res[j] : p.res[e_o[j]];  // residual stream in the same layout as out (in place is fine: one thread per element, read before written)
        cbx_store_out(p.out + e_o[j], v);
    }
#ifdef CBX_TRACE
    CBX_TRC_STAMP(5);  // epilogue stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CBX_TRC_STAMP(6);  // ... and acknowledged
    CBX_TRC_FLUSH(0x10000000u | (unsigned)(SWIGLU ? 0x1000000 : 0) | (unsigned)(p.K > p.N ? 0x8000000 : 0) | (unsigned)(NP << 20) | (unsigned)(p.N & 0xfffff));
#endif
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Column-tile / split-K form of the RMSNorm-folded packed GEMV (ABI v11, cbx_gemv_t.col_tiles >= 1): the q/k/v projection and the speech head
// of the decode step.  Why: the launch timeline of the decode step (profiles/r04_decode_launch_timeline.txt) shows these kernels bound by the
// bytes a CU moves through its own vector-memory path, and in the one-tile form above HALF or more of those bytes are ACTIVATIONS: a
// workgroup that owns 16 output columns reads the whole 16-row x (64 KiB, K = 1024) for 64 KiB of weights -- and with the two split-K partial
// images of the producing down projection folded in (NP = 2) 192 KiB for 64.  Here a workgroup owns CT column tiles that share every x
// register (x : W = 1 : CT) and, with ksplit > 1, only a 1/ksplit slice of K (x and the partial images shrink by ksplit as well):
//   q/k/v, CT = 3, ksplit = 4:  48 KiB of weights + 3 x 16 KiB of x / partial images per workgroup (was 64 + 192), 64 x 4 = 256 workgroups;
//   head,  CT = 2, ksplit = 1:  257 workgroups (one round of the chip instead of two) at 128 + 192 KiB (was 64 + 192 each, 513 of them).
// ksplit > 1 leaves UN-normalised partial sums out[ks][m][n] = sum_{k in slice} (x[m][k] norm_w[k]) W[n][k] plus the slice's sum of squares
// ssq_out[ks][m] (written by column group 0): rstd factors out of the contraction, so the CONSUMER -- the decode attention, which reads 192
// values per workgroup -- adds the ksplit partials in fixed order and applies rstd = rsqrt(sum_ks ssq / K + eps).  ksplit == 1: rstd in the
// epilogue as above.  Same MFMA, same fixed-order LDS reduction over the 8 waves; deterministic.
template <int CT, int NP, int DEPTH>
__global__ __launch_bounds__(512) void gemv_ct_kernel(const cbx_gemv_t p) {
    __builtin_amdgcn_s_setprio(3);  // decode-step kernels are latency-bound and issue little: beside a co-resident workgroup of another stream (the throughput schedule, profiles/r05_overlap_*) their waves go first at the SIMD's issue arbiter; alone on the CU it changes nothing
    constexpr int NW = 8;
    __shared__ __attribute__((aligned(16))) float red[NW * CT * 256];
    __shared__ float ssq[NW * 16];
    __shared__ float ssx[NW * 16];  // row sums (LayerNorm form: ln_cw / ln_cb, ksplit == 1)
    CBX_TRC_DECL;
    CBX_TRC_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ks = blockIdx.y;
    const long KB = p.K >> 5;                       // 32-deep K blocks of a row
    const int nit = (int)(KB / (p.ksplit * NW));    // blocks per wave
    const long kb0 = ((long)ks * NW + w) * nit;     // first block of this wave
    const int ntiles = (p.N + 15) >> 4;
    const int tile0 = blockIdx.x * CT;
    const float* wp[CT];
    bool wok[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        wok[c] = tile0 + c < ntiles;
        wp[c] = p.W + ((long)(wok[c] ? tile0 + c : tile0) * KB + kb0) * 512 + lane * 4;  // [tile][K/32][2][64][4]; a tile past N re-reads tile0
    }
    const bool xok = (lane & 15) < p.M;
    const long xo = kb0 * 512 + (xok ? lane : (lane & 48)) * 4;  // packed x: [K/32][2][64][4] (one 16-row tile); rows past M read row 0's bytes (gemv_kernel)
    const float* nwp = p.norm_w + kb0 * 32 + 8 * (lane >> 4);
    f32x4 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ss = 0.f, sx = 0.f;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // DEPTH = K blocks requested per batch (picked by the host so that it divides the wave's block count: an idle slot would re-read a block,
    // i.e. spend the very per-CU bytes this form saves; registers: DEPTH * 2 * (CT + NP + 2) float4)
    for (int it0 = 0; it0 < nit; it0 += DEPTH) {
        f32x4 wv[DEPTH][CT][2], xv[DEPTH][2], nv[DEPTH][2], pv[DEPTH][NP > 0 ? NP : 1][2];
        bool on[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            on[d] = it0 + d < nit;
            const int blk = on[d] ? it0 + d : 0;      // loads stay unconditional (counted waits): an idle slot re-reads block 0
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                wv[d][c][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp[c] + blk * 512));
                wv[d][c][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp[c] + blk * 512 + 256));
            }
            xv[d][0] = *reinterpret_cast<const f32x4*>(p.x + xo + blk * 512);
            xv[d][1] = *reinterpret_cast<const f32x4*>(p.x + xo + blk * 512 + 256);
            nv[d][0] = *reinterpret_cast<const f32x4*>(nwp + blk * 32);
            nv[d][1] = *reinterpret_cast<const f32x4*>(nwp + blk * 32 + 4);
            if constexpr (NP > 0) {
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const float* pp = p.xpart + (long)j * p.xpart_stride + xo + blk * 512;
                    pv[d][j][0] = *reinterpret_cast<const f32x4*>(pp);
                    pv[d][j][1] = *reinterpret_cast<const f32x4*>(pp + 256);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef CBX_TRACE
        if (it0 == 0) CBX_TRC_STAMP(1);
#endif
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 xq = xv[d][h];
                if constexpr (NP > 0) {
#pragma unroll
                    for (int j = 0; j < NP; ++j) xq += pv[d][j][h];
                    if (p.x_out && blockIdx.x == 0 && on[d] && xok)  // the reduced residual stream (this wave's K slice), same packed address as x; pad rows stay zero
                        cbx_store_out4(p.x_out + xo + (it0 + d) * 512 + h * 256, xq);
                }
                xq = (on[d] && xok) ? xq : zero4;
                ss += (xq[0] * xq[0] + xq[1] * xq[1]) + (xq[2] * xq[2] + xq[3] * xq[3]);
                sx += (xq[0] + xq[1]) + (xq[2] + xq[3]);
                xq *= nv[d][h];
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const f32x4 wq = (on[d] && wok[c]) ? wv[d][c][h] : zero4;
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(xq[s], wq[s], acc[c], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef CBX_TRACE
            if (it0 == 0 && d == 0) {
                asm volatile("s_nop 0" ::"v"(acc[0][0]));
                CBX_TRC_STAMP(2);
            }
#endif
        }
    }
#ifdef CBX_TRACE
    asm volatile("s_nop 0" ::"v"(acc[0][0]));
    CBX_TRC_STAMP(3);
#endif
    // fixed-order reduction over the 8 K slices of this workgroup.  D map of the MFMA: row = q*4 + r, col = c16
    const int c16 = lane & 15, q = lane >> 4;
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(w * CT + c) * 256 + (q * 4 + r) * 16 + c16] = acc[c][r];
    {
        float v = ss, u = sx;
        v += cbx_xor_lane<16>(v);
        v += cbx_xor_lane<32>(v);
        u += cbx_xor_lane<16>(u);
        u += cbx_xor_lane<32>(u);
        if (q == 0) ssq[w * 16 + c16] = v, ssx[w * 16 + c16] = u;
    }
    __syncthreads();
    CBX_TRC_STAMP(4);
    for (int e = tid; e < CT * 256; e += NW * 64) {
        const int c = e >> 8, rc = e & 255, row = rc >> 4, col = rc & 15;
        const int n = (tile0 + c) * 16 + col;
        if (row >= p.M || n >= p.N) continue;
        float v = 0.f, sq = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) v += red[(ww * CT + c) * 256 + rc];
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) sq += ssq[ww * 16 + row];
        if (p.ln_cw) {  // LayerNorm form (GPT-2 ln_f + head; ksplit == 1): the expression of gemv_kernel
            float su = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) su += ssx[ww * 16 + row];
            const float mean = su / (float)p.K;
            const float rstd = rsqrtf(fmaxf(sq / (float)p.K - mean * mean, 0.f) + p.eps);
            v = rstd * (v - mean * p.ln_cw[n]) + p.ln_cb[n];
        } else if (p.ksplit == 1) {
            v *= rsqrtf(sq / (float)p.K + p.eps);
        } else if (blockIdx.x == 0 && c == 0 && col == 0) {
            cbx_store_out(p.ssq_out + ks * 16 + row, sq);  // this K slice's sum of squares of row `row` (identical in every column group: group 0 writes it)
        }
        cbx_store_out(p.out + (long)ks * p.part_stride + (long)row * p.ldo + n, v);
    }
#ifdef CBX_TRACE
    CBX_TRC_STAMP(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CBX_TRC_STAMP(6);
    CBX_TRC_FLUSH(0x10000000u | 0x2000000u | (unsigned)(NP << 20) | (unsigned)(p.N & 0xfffff));
#endif
}

template <int CT, int NP>
int launch_ct_np(const cbx_gemv_t& p, hipStream_t st) {
    const int ntiles = (p.N + 15) / 16;
    dim3 grid((ntiles + CT - 1) / CT, p.ksplit);
    const int nit = p.K / (32 * p.ksplit * 8);  // K blocks per wave
    if (nit % 2 == 0) hipLaunchKernelGGL((gemv_ct_kernel<CT, NP, 2>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((gemv_ct_kernel<CT, NP, 1>), grid, dim3(512), 0, st, p);
    return cbx_check_launch("gemv (column tiles)");
}

template <int CT>
int launch_ct(const cbx_gemv_t& p, hipStream_t st) {
    if (p.n_xpart == 2) return launch_ct_np<CT, 2>(p, st);
    if (p.n_xpart == 4) return launch_ct_np<CT, 4>(p, st);
    return launch_ct_np<CT, 0>(p, st);
}

// process-wide TEST HOOKS: ORed into cbx_gemv_t.flags of every cbx_gemv_f32 launch (the engines set the flags per launch instead)
int g_gemv_deep = 0;     // cbx_set_gemv_deep_batches
int g_gemv_pre_epi = 0;  // cbx_set_gemv_epilogue_prefetch

template <int MT, bool SWIGLU, bool PK, bool XPK, bool RMS, int NP = 0, bool WB = false>
int launch_nw(const cbx_gemv_t& p, hipStream_t st) {
    const int tc = PK ? gemv_tile_cols(p.half_tile) : 16;
    dim3 grid((p.N + tc - 1) / tc, p.ksplit);
    if constexpr (MT == 1 && !SWIGLU && !RMS && (PK == XPK)) {
        if constexpr (PK) {  // a K slice of >= 256 per wave (ABI v9: the down projection with ksplit = 1 on 8 waves): batches of 8 K blocks in flight.
            // 8-wave workgroups only: with 16 waves (128 VGPRs per wave) the 8-deep form spills
            if ((p.flags & CBX_GEMV_DEEP) && p.nw >= 8 && p.nw != 16 && p.K / (p.ksplit * 8) >= 256) {
                hipLaunchKernelGGL((gemv_kernel<1, 8, false, true, true, false, 0, WB, true>), grid, dim3(512), 0, st, p);
                return cbx_check_launch("gemv");
            }
        }
        if (p.nw == 16) {  // 16 K-slices per workgroup: projections whose output tile count (N / 16) is small
            hipLaunchKernelGGL((gemv_kernel<1, 16, false, PK, XPK, false, 0, WB>), grid, dim3(1024), 0, st, p);
            return cbx_check_launch("gemv");
        }
    }
    if constexpr (MT == 1 && SWIGLU && PK && XPK && RMS && NP == 0 && !WB) {
        if ((p.flags & CBX_GEMV_SHALLOW) && p.nw >= 8) {
            hipLaunchKernelGGL((gemv_kernel<1, 8, true, true, true, true, 0, false, false, true>), grid, dim3(512), 0, st, p);
            return cbx_check_launch("gemv");
        }
    }
    if (p.nw >= 8 || NP > 0) {
        hipLaunchKernelGGL((gemv_kernel<MT, 8, SWIGLU, PK, XPK, RMS, NP, WB>), grid, dim3(512), 0, st, p);
    } else {
        hipLaunchKernelGGL((gemv_kernel<MT, 4, SWIGLU, PK, XPK, RMS, 0, WB>), grid, dim3(256), 0, st, p);
    }
    return cbx_check_launch("gemv");
}

template <int MT, bool SWIGLU>
int launch_pk(const cbx_gemv_t& p, hipStream_t st) {
    if (p.w_bf16) {  // checked: w_packed && x_packed, rows <= 16
        if constexpr (MT == 1) {
            if (p.norm_w) {
                if (p.n_xpart == 2) return launch_nw<1, SWIGLU, true, true, true, 2, true>(p, st);
                if constexpr (!SWIGLU)
                    if (p.n_xpart == 4) return launch_nw<1, false, true, true, true, 4, true>(p, st);
                return launch_nw<1, SWIGLU, true, true, true, 0, true>(p, st);
            }
            return launch_nw<1, SWIGLU, true, true, false, 0, true>(p, st);
        }
        return cbx_set_error(CBX_EINVAL, "gemv: bf16 weights serve M <= 16");
    }
    if (p.norm_w) {  // checked: w_packed && x_packed && ksplit == 1
        if constexpr (MT == 1) {
            if (p.n_xpart == 2) return launch_nw<1, SWIGLU, true, true, true, 2>(p, st);
            if constexpr (!SWIGLU)  // (the swiglu form would need > 256 VGPRs with 4 partial images in flight)
                if (p.n_xpart == 4) return launch_nw<1, false, true, true, true, 4>(p, st);
        }
        return launch_nw<MT, SWIGLU, true, true, true>(p, st);
    }
    if (p.w_packed && p.x_packed) return launch_nw<MT, SWIGLU, true, true, false>(p, st);
    if (p.w_packed) return launch_nw<MT, SWIGLU, true, false, false>(p, st);
    return launch_nw<MT, SWIGLU, false, false, false>(p, st);
}

template <bool SWIGLU>
int launch_mt(const cbx_gemv_t& p, hipStream_t st) {
    switch ((p.M + 15) / 16) {
        case 1: return launch_pk<1, SWIGLU>(p, st);
        case 2: return launch_pk<2, SWIGLU>(p, st);
        case 3:
        case 4: return launch_pk<4, SWIGLU>(p, st);
    }
    return cbx_set_error(CBX_EINVAL, "gemv: M=%d > 64", p.M);
}

// x[row] += sum_ks partial[ks][row]  (fixed order);  h[row] = rmsnorm(x[row]) * w.
// One 256-thread workgroup per row, one float4 (x up to 4) per thread, all partial loads issued before the first add:
// the op is pure latency (16..64 rows), so the only lever is memory-level parallelism.
constexpr int AR_MAXKS = 8;
__global__ __launch_bounds__(256) void add_rmsnorm_kernel(float* x, const float* __restrict__ part, int ksplit, long part_stride,
                                                          long ldp, const float* __restrict__ w, const float* __restrict__ b,
                                                          float* __restrict__ h, int rows, int C, long ldx, long ldh, float eps,
                                                          int rms) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int nv = C >> 2;
    float* xr = x + (long)row * ldx;
    f32x4 v[4];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c4 = i * 256 + tid;
        if (c4 < nv) {
            f32x4 t = *reinterpret_cast<const f32x4*>(xr + c4 * 4);
            if (ksplit > 0) {  // unconditional loads from a clamped slice index: every load is in flight before the first add
                f32x4 pk[AR_MAXKS];
#pragma unroll
                for (int k = 0; k < AR_MAXKS; ++k) {
                    const int kk = k < ksplit ? k : ksplit - 1;
                    pk[k] = *reinterpret_cast<const f32x4*>(part + kk * part_stride + (long)row * ldp + c4 * 4);
                }
#pragma unroll
                for (int k = 0; k < AR_MAXKS; ++k) {
                    const float on = k < ksplit ? 1.0f : 0.0f;
                    t += pk[k] * on;
                }
            }
            v[i] = t;
            if (ksplit > 0) *reinterpret_cast<f32x4*>(xr + c4 * 4) = t;
            ss += t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3];
        }
    }
    float mean = 0.f;
    if (!rms) {  // LayerNorm (GPT-2): two-pass mean / variance over the row held in registers
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i * 256 + tid < nv) sm += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        sm = wave_sum(sm);
        if ((tid & 63) == 0) red[tid >> 6] = sm;
        __syncthreads();
        mean = ((red[0] + red[1]) + (red[2] + red[3])) / C;
        __syncthreads();
        ss = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i * 256 + tid < nv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[i][e] - mean;
                    ss += d * d;
                }
            }
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    ss = (red[0] + red[1]) + (red[2] + red[3]);
    const float rstd = rsqrtf(ss / C + eps);
    float* hr = h + (long)row * ldh;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c4 = i * 256 + tid;
        if (c4 < nv) {
            f32x4 wv = *reinterpret_cast<const f32x4*>(w + c4 * 4);
            f32x4 bv = b ? *reinterpret_cast<const f32x4*>(b + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * wv[e] + bv[e];
            *reinterpret_cast<f32x4*>(hr + c4 * 4) = o;
        }
    }
}

// one thread per float4 of the packed image (layout: include/cbx.h "Packed GEMV weight layout")
__global__ __launch_bounds__(256) void pack_gemv_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int K,
                                                               long ld, int swiglu, long n4, int half_tile) {
    const long i4 = (long)blockIdx.x * 256 + threadIdx.x;
    if (i4 >= n4) return;
    if (half_tile) {  // narrow tiles of tr = 8 / 12 / 4 rows: [tile][K/32][2][4 * tr lanes = q * tr + c][4]
        const int tr = half_tile, nl = 4 * tr;
        const int l32 = (int)(i4 % nl), h = (int)((i4 / nl) & 1);
        const long tb = i4 / (2 * nl);
        const int KB = K >> 5;
        const long tile = tb / KB;
        const int kb = (int)(tb - tile * KB);
        const long r = tile * tr + (l32 % tr);
        const int k = kb * 32 + (l32 / tr) * 8 + h * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < N) v = *reinterpret_cast<const f32x4*>(src + r * ld + k);
        *reinterpret_cast<f32x4*>(dst + i4 * 4) = v;
        return;
    }
    const int lane = i4 & 63, h = (i4 >> 6) & 1;
    const long tb = i4 >> 7;  // tile * KB + kb
    const int KB = K >> 5;
    const long tile = tb / KB;
    const int kb = (int)(tb - tile * KB);
    long r;
    bool ok;
    if (swiglu) {
        const long f = (tile >> 1) * 16 + (lane & 15);
        ok = f < N;
        r = (tile & 1) * (long)N + f;
    } else {
        r = tile * 16 + (lane & 15);
        ok = r < N;
    }
    const int k = kb * 32 + (lane >> 4) * 8 + h * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok) v = *reinterpret_cast<const f32x4*>(src + r * ld + k);
    *reinterpret_cast<f32x4*>(dst + i4 * 4) = v;
}

// bf16 image: [tile][K/32][lanes][8 bf16], lane (c, q) holds k = 8q .. 8q+7 of row c; RNE rounding (__float2bfloat16 semantics)
__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__global__ __launch_bounds__(256) void pack_gemv_weight_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int N,
                                                                    int K, long ld, int swiglu, long nl, int half_tile) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // one thread per (tile, kb, lane): 8 bf16
    if (i >= nl) return;
    const int rows_t = half_tile ? half_tile : 16, lanes = 4 * rows_t;
    const int l = (int)(i % lanes);
    const long tb = i / lanes;
    const int KB = K >> 5;
    const long tile = tb / KB;
    const int kb = (int)(tb - tile * KB);
    const int c = l % rows_t, q = l / rows_t;
    long r;
    bool ok;
    if (swiglu) {
        const long f = (tile >> 1) * 16 + c;
        ok = f < N;
        r = (tile & 1) * (long)N + f;
    } else {
        r = tile * rows_t + c;
        ok = r < N;
    }
    const float* sp = src + r * ld + kb * 32 + q * 8;
    unsigned short o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = ok ? f32_to_bf16_rne(sp[e]) : (unsigned short)0;
    uint4 pk;
    pk.x = o[0] | ((unsigned)o[1] << 16), pk.y = o[2] | ((unsigned)o[3] << 16), pk.z = o[4] | ((unsigned)o[5] << 16), pk.w = o[6] | ((unsigned)o[7] << 16);
    *reinterpret_cast<uint4*>(dst + i * 8) = pk;
}

}  // namespace
CBX_TRC_SETTER(cbx_trace_set_gemv)

extern "C" int cbx_pack_gemv_weight_bf16(const float* src, void* dst, int N, int K, long ld_src, int swiglu, void* stream) {
    CBX_REQUIRE(src && dst && N > 0 && K > 0 && K % 32 == 0 && ld_src % 4 == 0, "pack_gemv_weight_bf16: bad args (K %% 32, ld %% 4)");
    const int half_tile = (swiglu == 8 || swiglu == 12 || swiglu == 4) ? swiglu : 0;  // rows per narrow tile
    if (half_tile) swiglu = 0;
    const long tiles = half_tile ? (N + half_tile - 1) / half_tile : (long)((N + 15) / 16) * (swiglu ? 2 : 1);
    const long nl = tiles * (K / 32) * (half_tile ? 4 * half_tile : 64);
    hipLaunchKernelGGL(pack_gemv_weight_bf16_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                       (unsigned short*)dst, N, K, ld_src, swiglu, nl, half_tile);
    return cbx_check_launch("pack_gemv_weight_bf16");
}

extern "C" int cbx_pack_gemv_weight_f32(const float* src, float* dst, int N, int K, long ld_src, int swiglu, void* stream) {
    CBX_REQUIRE(src && dst && N > 0 && K > 0 && K % 32 == 0 && ld_src % 4 == 0, "pack_gemv_weight: bad args (K %% 32, ld %% 4)");
    CBX_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "pack_gemv_weight: alignment");
    const int half_tile = (swiglu == 8 || swiglu == 12 || swiglu == 4) ? swiglu : 0;  // swiglu = 8 / 12 / 4 selects the narrow-tile image (cbx_gemv_t.half_tile) of a plain weight
    if (half_tile) swiglu = 0;
    const long tiles = half_tile ? (N + half_tile - 1) / half_tile : (long)((N + 15) / 16) * (swiglu ? 2 : 1);
    const long n4 = tiles * (K / 32) * (half_tile ? 8 * half_tile : 128);
    hipLaunchKernelGGL(pack_gemv_weight_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, N, K,
                       ld_src, swiglu, n4, half_tile);
    return cbx_check_launch("pack_gemv_weight");
}

extern "C" int cbx_set_gemv_deep_batches(int on) {
    g_gemv_deep = on != 0;
    return 0;
}

extern "C" int cbx_set_gemv_epilogue_prefetch(int on) {
    g_gemv_pre_epi = on != 0;
    return 0;
}

extern "C" int cbx_gemv_f32(const cbx_gemv_t* pp, void* stream) {
    cbx_gemv_t p = *pp;
    if (p.ksplit < 1) p.ksplit = 1;
    if (p.nw != 8 && p.nw != 16) p.nw = 4;
    p.flags = (p.flags & (CBX_GEMV_PRE_EPI | CBX_GEMV_DEEP | CBX_GEMV_SHALLOW)) | (g_gemv_pre_epi ? CBX_GEMV_PRE_EPI : 0) | (g_gemv_deep ? CBX_GEMV_DEEP : 0);
    CBX_REQUIRE(p.x && p.W && p.out, "gemv: null operand");
    CBX_REQUIRE(p.M >= 1 && p.M <= 64 && p.N > 0 && p.K > 0, "gemv: bad shape M=%d N=%d K=%d", p.M, p.N, p.K);
    CBX_REQUIRE(p.col_tiles > 0 || p.K % (32 * p.ksplit * p.nw) == 0, "gemv: K=%d must be a multiple of 32*ksplit*nw=%d", p.K, 32 * p.ksplit * p.nw);
    CBX_REQUIRE(p.ldx % 4 == 0 && p.ldw % 4 == 0 && (((uintptr_t)p.x | (uintptr_t)p.W) & 15) == 0, "gemv: alignment");
    CBX_REQUIRE(!p.swiglu || (p.ksplit == 1 && p.N % 32 == 0), "gemv: swiglu needs ksplit == 1 and N %% 32 == 0");
    CBX_REQUIRE(p.half_tile == 0 || p.half_tile == 1 || p.half_tile == 8 || p.half_tile == 12 || p.half_tile == 4, "gemv: half_tile must be 0, 1 (= 8), 8, 12 or 4");
    CBX_REQUIRE(p.half_tile == 0 || (p.w_packed && !p.swiglu), "gemv: narrow tiles need the matching packed image and no swiglu");
    CBX_REQUIRE(!p.act || p.ksplit == 1, "gemv: an activation epilogue needs ksplit == 1");
    CBX_REQUIRE(!p.x_packed || p.w_packed, "gemv: x_packed needs w_packed");
    CBX_REQUIRE(!p.w_bf16 || (p.w_packed && p.x_packed && p.M <= 16), "gemv: w_bf16 needs w_packed, x_packed and M <= 16");
    CBX_REQUIRE(!p.norm_w || (p.w_packed && p.x_packed && (p.ksplit == 1 || p.col_tiles > 0)), "gemv: norm_w needs w_packed, x_packed and ksplit == 1 (or col_tiles)");
    CBX_REQUIRE(!p.res || p.ksplit == 1, "gemv: res needs ksplit == 1");
    CBX_REQUIRE(!p.ln_cw || (p.norm_w && p.ln_cb && !p.swiglu && !p.bias), "gemv: the LayerNorm form needs norm_w, ln_cb, no swiglu, bias folded into ln_cb");
    CBX_REQUIRE(p.n_xpart == 0 || p.col_tiles > 0 || (p.norm_w && p.xpart && p.M <= 16 && (p.n_xpart == 2 || (p.n_xpart == 4 && !p.swiglu)) && p.nw == 8 && p.x_out != p.x),
                "gemv: xpart needs norm_w, M <= 16, n_xpart in {2, 4}, nw == 8 and x_out != x");
    CBX_REQUIRE(!p.out_packed || p.N % 32 == 0, "gemv: out_packed needs N %% 32 == 0");
    CBX_REQUIRE(!(p.w_packed || p.x_packed) || p.K % 32 == 0, "gemv: packed operands need K %% 32 == 0");
    if (p.col_tiles > 0) {  // column-tile / split-K form of the RMSNorm-folded packed GEMV (gemv_ct_kernel)
        CBX_REQUIRE(p.col_tiles <= 4 && p.norm_w && p.w_packed && p.x_packed && !p.w_bf16 && !p.swiglu && !p.bias && !p.res && !p.act && !p.out_packed &&
                        p.half_tile == 0 && p.M <= 16 && (!p.ln_cw || (p.ln_cb && p.ksplit == 1)),
                    "gemv: col_tiles serves the plain RMSNorm- / LayerNorm-folded packed fp32 form (M <= 16, 16-column image, no bias / residual / activation; LayerNorm form: ksplit == 1)");
        CBX_REQUIRE(p.K % (256 * p.ksplit) == 0 && (p.ksplit == 1 || p.ssq_out), "gemv: col_tiles needs K %% (256 * ksplit) == 0, and ssq_out with ksplit > 1");
        CBX_REQUIRE(p.n_xpart == 0 || ((p.n_xpart == 2 || p.n_xpart == 4) && p.xpart && p.x_out != p.x), "gemv: col_tiles with xpart: 2 or 4 images, x_out != x");
        hipStream_t st = (hipStream_t)stream;
        switch (p.col_tiles) {
            case 1: return launch_ct<1>(p, st);
            case 2: return launch_ct<2>(p, st);
            case 3: return launch_ct<3>(p, st);
            default: return launch_ct<4>(p, st);
        }
    }
    return p.swiglu ? launch_mt<true>(p, (hipStream_t)stream) : launch_mt<false>(p, (hipStream_t)stream);
}

extern "C" int cbx_add_norm_f32(float* x, const float* part, int ksplit, long part_stride, long ldp, const float* w, const float* b,
                                float* h, int rows, int C, long ldx, long ldh, float eps, int rms, void* stream);

extern "C" int cbx_add_rmsnorm_f32(float* x, const float* part, int ksplit, long part_stride, long ldp, const float* w, float* h,
                                   int rows, int C, long ldx, long ldh, float eps, void* stream) {
    return cbx_add_norm_f32(x, part, ksplit, part_stride, ldp, w, nullptr, h, rows, C, ldx, ldh, eps, 1, stream);
}

extern "C" int cbx_add_norm_f32(float* x, const float* part, int ksplit, long part_stride, long ldp, const float* w, const float* b,
                                float* h, int rows, int C, long ldx, long ldh, float eps, int rms, void* stream) {
    CBX_REQUIRE(x && w && h && (ksplit == 0 || part), "add_rmsnorm: null operand");
    CBX_REQUIRE(C % 4 == 0 && C <= 4096 && ldx % 4 == 0 && ldh % 4 == 0 && ldp % 4 == 0 && part_stride % 4 == 0, "add_rmsnorm: alignment");
    CBX_REQUIRE(ksplit >= 0 && ksplit <= AR_MAXKS, "add_rmsnorm: ksplit=%d > %d", ksplit, AR_MAXKS);
    hipLaunchKernelGGL(add_rmsnorm_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, part, ksplit, part_stride, ldp,
                       w, b, h, rows, C, ldx, ldh, eps, rms);
    return cbx_check_launch("add_rmsnorm");
}
