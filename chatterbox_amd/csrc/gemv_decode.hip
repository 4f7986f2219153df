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
#include "cbx_common.h"

namespace {

template <int MT, int NW, bool SWIGLU>
__global__ __launch_bounds__(NW * 64) void gemv_kernel(const cbx_gemv_t p) {
    __shared__ __attribute__((aligned(16))) float red[(SWIGLU ? 2 : 1) * NW * MT * 256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.x * 16, ks = blockIdx.y;
    const int kper = p.K / (p.ksplit * NW);
    const int kbeg = (ks * NW + w) * kper;
    const int nit = kper / 32;

    // row of W streamed by this lane (swiglu: gate row, the matching up row is 32 rows further in the packed image)
    long wrow;
    bool wok;
    if constexpr (SWIGLU) {
        const int f = n0 + c;  // feature index
        wok = f < p.N;
        wrow = (long)(f >> 5) * 64 + (f & 31);
    } else {
        wok = (n0 + c) < p.N;
        wrow = n0 + c;
    }
    const float* wp = p.W + (wok ? wrow : 0) * p.ldw + kbeg + 8 * q;
    const float* wp2 = wp + 32 * p.ldw;
    const float* xp[MT];
    bool xok[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int m = t * 16 + c;
        xok[t] = m < p.M;
        xp[t] = p.x + (long)(xok[t] ? m : 0) * p.ldx + kbeg + 8 * q;
    }

    f32x4 acc[MT], acc2[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    constexpr int DEPTH = (MT == 1) ? 4 : 2;  // K blocks (2 KiB of W per wave each) issued before the first MFMA
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // Loads are UNCONDITIONAL (out-of-range lanes / K blocks re-read a valid address and are zeroed by a select): a predicated
    // load makes hipcc join all of them behind one vmcnt(0); unconditional ones get counted waits, so the MFMAs of K block d
    // start while blocks d+1.. are still in flight.
    for (int it0 = 0; it0 < nit; it0 += DEPTH) {
        f32x4 wv[DEPTH][2], uv[DEPTH][2], xv[DEPTH][MT][2];
        bool on[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            on[d] = (it0 + d) < nit;
            const int off = on[d] ? (it0 + d) * 32 : 0;
            wv[d][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp + off));
            wv[d][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp + off + 4));
            if constexpr (SWIGLU) {
                uv[d][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp2 + off));
                uv[d][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp2 + off + 4));
            }
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                xv[d][t][0] = *reinterpret_cast<const f32x4*>(xp[t] + off);
                xv[d][t][1] = *reinterpret_cast<const f32x4*>(xp[t] + off + 4);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the issue order block by block, so block d's wait is vmcnt(later blocks)
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const bool won = on[d] && wok;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 wq = won ? wv[d][h] : zero4;
                f32x4 uq = zero4;
                if constexpr (SWIGLU) uq = won ? uv[d][h] : zero4;
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const f32x4 xq = (on[d] && xok[t]) ? xv[d][t][h] : zero4;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xq[s], wq[s], acc[t], 0, 0, 0);
                        if constexpr (SWIGLU) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xq[s], uq[s], acc2[t], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

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
    __syncthreads();
    for (int e = tid; e < MT * 256; e += NW * 64) {
        const int t = e >> 8, rc = e & 255, row = rc >> 4, col = rc & 15;
        const int m = t * 16 + row, n = n0 + col;
        if (m >= p.M || n >= p.N) continue;
        float v = 0.f, v2 = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
            v += r1[(ww * MT + t) * 256 + rc];
            if constexpr (SWIGLU) v2 += r2[(ww * MT + t) * 256 + rc];
        }
        if constexpr (SWIGLU) {
            v = (v / (1.0f + __expf(-v))) * v2;
        } else {
            if (p.bias && ks == 0) v += p.bias[n];
            if (p.act) v = cbx_act(v, p.act, 0.f, 0.f);  // only meaningful with ksplit == 1
        }
        p.out[(long)ks * p.part_stride + (long)m * p.ldo + n] = v;
    }
}

template <int MT, bool SWIGLU>
int launch_nw(const cbx_gemv_t& p, hipStream_t st) {
    dim3 grid((p.N + 15) / 16, p.ksplit);
    if constexpr (MT == 1 && !SWIGLU) {
        if (p.nw == 16) {  // 16 K-slices per workgroup: the ksplit <= 2 form of the K = 4096 down-projection (experimental fused-norm path)
            hipLaunchKernelGGL((gemv_kernel<1, 16, false>), grid, dim3(1024), 0, st, p);
            return cbx_check_launch("gemv");
        }
    }
    if (p.nw >= 8) {
        hipLaunchKernelGGL((gemv_kernel<MT, 8, SWIGLU>), grid, dim3(512), 0, st, p);
    } else {
        hipLaunchKernelGGL((gemv_kernel<MT, 4, SWIGLU>), grid, dim3(256), 0, st, p);
    }
    return cbx_check_launch("gemv");
}

template <bool SWIGLU>
int launch_mt(const cbx_gemv_t& p, hipStream_t st) {
    switch ((p.M + 15) / 16) {
        case 1: return launch_nw<1, SWIGLU>(p, st);
        case 2: return launch_nw<2, SWIGLU>(p, st);
        case 3:
        case 4: return launch_nw<4, SWIGLU>(p, st);
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

}  // namespace

extern "C" int cbx_gemv_f32(const cbx_gemv_t* pp, void* stream) {
    cbx_gemv_t p = *pp;
    if (p.ksplit < 1) p.ksplit = 1;
    if (p.nw != 8 && p.nw != 16) p.nw = 4;
    CBX_REQUIRE(p.x && p.W && p.out, "gemv: null operand");
    CBX_REQUIRE(p.M >= 1 && p.M <= 64 && p.N > 0 && p.K > 0, "gemv: bad shape M=%d N=%d K=%d", p.M, p.N, p.K);
    CBX_REQUIRE(p.K % (32 * p.ksplit * p.nw) == 0, "gemv: K=%d must be a multiple of 32*ksplit*nw=%d", p.K, 32 * p.ksplit * p.nw);
    CBX_REQUIRE(p.ldx % 4 == 0 && p.ldw % 4 == 0 && (((uintptr_t)p.x | (uintptr_t)p.W) & 15) == 0, "gemv: alignment");
    CBX_REQUIRE(!p.swiglu || (p.ksplit == 1 && p.N % 32 == 0), "gemv: swiglu needs ksplit == 1 and N %% 32 == 0");
    CBX_REQUIRE(!p.act || p.ksplit == 1, "gemv: an activation epilogue needs ksplit == 1");
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
