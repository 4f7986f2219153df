// LayerNorm / RMSNorm with fused activation: one 64-lane wavefront per row, row held in registers,
// two-pass statistics by wave shuffles (no LDS, no atomics).  HBM-bound: one read + one write per element.
#include <stdlib.h>
#include "cbx_common.h"

namespace {

constexpr int MAXV = 16;  // float4 per lane -> C <= 4096

__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        const float* __restrict__ post_add, long rows, int C,
                                                        long ldx, long ldy, float eps, int rms, int act,
                                                        float out_scale) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = C >> 2;  // float4 count per row
    const float* xr = x + row * ldx;
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int c4 = i * 64 + lane;
        if (c4 < nv) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + c4 * 4);
            s += rms ? (v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3])
                     : (v[i][0] + v[i][1] + v[i][2] + v[i][3]);
        }
    }
    s = wave_sum(s);
    float mean = 0.f, rstd;
    if (rms) {
        rstd = rsqrtf(s / C + eps);
    } else {
        mean = s / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int c4 = i * 64 + lane;
            if (c4 < nv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float d = v[i][e] - mean;
                    q += d * d;
                }
            }
        }
        q = wave_sum(q);
        rstd = rsqrtf(q / C + eps);
    }
    float* yr = y + row * ldy;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int c4 = i * 64 + lane;
        if (c4 < nv) {
            f32x4 wv = *reinterpret_cast<const f32x4*>(w + c4 * 4);
            f32x4 bv = {0.f, 0.f, 0.f, 0.f}, pv = {0.f, 0.f, 0.f, 0.f};
            if (b) bv = *reinterpret_cast<const f32x4*>(b + c4 * 4);
            if (post_add) pv = *reinterpret_cast<const f32x4*>(post_add + c4 * 4);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = (v[i][e] - mean) * rstd * wv[e] + bv[e];
                t = cbx_act(t, act, 0.f, 0.f);
                o[e] = t * out_scale + pv[e];
            }
            *reinterpret_cast<f32x4*>(yr + c4 * 4) = o;
        }
    }
}

// Narrow rows (C = 64*NV floats, NV = 4: the 256-channel LayerNorms of the CFM estimator -- 1500 launches per utterance
// batch): 16 lanes per row, 4 rows per wave, NV float4 per lane all in flight at once (a wave streams 4 KiB instead of 1 KiB
// per round trip), statistics by 4 xor-shuffles inside the 16-lane group.
// STATS: write {mean, rstd} per row to y (2 floats per row) instead of the normalised row (cbx_row_stats_f32): same loads, same reductions.
// PL: the normalised row is written in PLANE format (two fp16 planes h, l = 2048 (y - h), see gemm_planes.hip) for a consuming
// cbx_gemm_planes: y is the fp16 base, ldy the row stride in halves, p_lo the offset of the l plane; same bytes as the fp32 row.
template <int NV, int RPT, bool STATS = false, bool PL = false>
__global__ __launch_bounds__(256) void layernorm_narrow_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               const float* __restrict__ w, const float* __restrict__ b,
                                                               const float* __restrict__ post_add, long rows, long ldx, long ldy,
                                                               float eps, int rms, int act, float out_scale, long p_lo = 0,
                                                               int* range_flag = nullptr) {
    constexpr int C = 64 * NV;
    // RPT = rows per 16-lane group: RPT x NV float4 loads in flight per lane (RPT x 4 KiB per wave per round trip)
    const int l16 = threadIdx.x & 15;
    const long row0 = (long)blockIdx.x * (16 * RPT) + (threadIdx.x >> 4);
    f32x4 v[RPT][NV];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const long row = row0 + 16 * r;
        const float* xr = x + (row < rows ? row : rows - 1) * ldx;  // unconditional loads on a clamped row
#pragma unroll
        for (int i = 0; i < NV; ++i) v[r][i] = *reinterpret_cast<const f32x4*>(xr + (i * 16 + l16) * 4);
    }
    // per-channel parameters: loaded once, shared by both rows
    f32x4 wv[NV], bv[NV], pv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if constexpr (STATS) continue;
        const int c = (i * 16 + l16) * 4;
        wv[i] = *reinterpret_cast<const f32x4*>(w + c);
        bv[i] = b ? *reinterpret_cast<const f32x4*>(b + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        pv[i] = post_add ? *reinterpret_cast<const f32x4*>(post_add + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto group_sum = [](float t) {
        t += cbx_xor_lane<8>(t);
        t += cbx_xor_lane<4>(t);
        t += cbx_xor_lane<2>(t);
        t += cbx_xor_lane<1>(t);
        return t;
    };
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const long row = row0 + 16 * r;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            s += rms ? (v[r][i][0] * v[r][i][0] + v[r][i][1] * v[r][i][1] + v[r][i][2] * v[r][i][2] + v[r][i][3] * v[r][i][3])
                     : (v[r][i][0] + v[r][i][1] + v[r][i][2] + v[r][i][3]);
        s = group_sum(s);
        float mean = 0.f, rstd;
        if (rms) {
            rstd = rsqrtf(s / C + eps);
        } else {
            mean = s / C;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[r][i][e] - mean;
                    q += d * d;
                }
            q = group_sum(q);
            rstd = rsqrtf(q / C + eps);
        }
        if (row >= rows) continue;
        if constexpr (STATS) {
            if (l16 == 0) {
                y[2 * row] = mean;
                y[2 * row + 1] = rstd;
            }
            continue;
        }
        float* yr = y + row * ldy;
        [[maybe_unused]] float amax = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 16 + l16) * 4;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = (v[r][i][e] - mean) * rstd * wv[i][e] + bv[i][e];
                t = cbx_act(t, act, 0.f, 0.f);
                o[e] = t * out_scale + pv[i][e];
            }
            if constexpr (PL) {
                typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                _Float16* pr = reinterpret_cast<_Float16*>(y) + row * ldy + c;
                const f16x4 h = __builtin_convertvector(o, f16x4);
                const f32x4 t2 = o * CBX_F16_LO_SCALE;
                f32x4 d;
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = __builtin_fmaf((float)h[e], -CBX_F16_LO_SCALE, t2[e]);
                *reinterpret_cast<f16x4*>(pr) = h;
                *reinterpret_cast<f16x4*>(pr + p_lo) = __builtin_convertvector(d, f16x4);
                cbx_amax4(amax, o);
            } else {
                *reinterpret_cast<f32x4*>(yr + c) = o;
            }
        }
        if constexpr (PL) {
            if (amax > 65504.f && range_flag) atomicOr(range_flag, 1);
        }
    }
}

}  // namespace

extern "C" int cbx_layernorm_f32(const float* x, float* y, const float* w, const float* b, const float* post_add,
                                 long rows, int C, long ldx, long ldy, float eps, int rms, int act, float out_scale,
                                 void* stream) {
    CBX_REQUIRE(x && y && w, "layernorm: null operand");
    CBX_REQUIRE(C % 4 == 0 && C <= 4096 && ldx % 4 == 0 && ldy % 4 == 0, "layernorm: C=%d ldx=%ld ldy=%ld", C, ldx, ldy);
    if (rows <= 0) return 0;
    constexpr int narrow = 1;  // (2: the two-rows-per-wave form, 0: the generic kernel -- measured slower at C = 256: profiles/r02_layernorm_variants.log)
    if (narrow && C == 256 && rows >= 64) {
        if (narrow == 2)
            hipLaunchKernelGGL((layernorm_narrow_kernel<4, 2>), dim3((unsigned)((rows + 31) / 32)), dim3(256), cbx_coresident_lds((hipStream_t)stream, 0, 2), (hipStream_t)stream, x, y, w,
                               b, post_add, rows, ldx, ldy, eps, rms, act, out_scale);
        else
            hipLaunchKernelGGL((layernorm_narrow_kernel<4, 1>), dim3((unsigned)((rows + 15) / 16)), dim3(256), cbx_coresident_lds((hipStream_t)stream, 0, 2), (hipStream_t)stream, x, y, w,
                               b, post_add, rows, ldx, ldy, eps, rms, act, out_scale);
        return cbx_check_launch("layernorm");
    }
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), cbx_coresident_lds((hipStream_t)stream, 0, 2), (hipStream_t)stream, x, y, w, b,
                       post_add, rows, C, ldx, ldy, eps, rms, act, out_scale);
    return cbx_check_launch("layernorm");
}

extern "C" int cbx_layernorm_planes_f32(const float* x, void* planes, const float* w, const float* b, const float* post_add, long rows,
                                        int C, long ldx, long ldp, long p_lo, float eps, int act, float out_scale, void* stream) {
    CBX_REQUIRE(x && planes && w, "layernorm_planes: null operand");
    CBX_REQUIRE(C == 256 && ldx % 4 == 0 && ldp % 4 == 0 && p_lo % 4 == 0, "layernorm_planes: C must be 256 (got %d); strides multiples of 4", C);
    if (rows <= 0) return 0;
    hipLaunchKernelGGL((layernorm_narrow_kernel<4, 1, false, true>), dim3((unsigned)((rows + 15) / 16)), dim3(256), cbx_coresident_lds((hipStream_t)stream, 0, 2), (hipStream_t)stream, x,
                       reinterpret_cast<float*>(planes), w, b, post_add, rows, ldx, ldp, eps, 0, act, out_scale, p_lo, cbx_range_flag());
    return cbx_check_launch("layernorm_planes");
}

extern "C" int cbx_row_stats_f32(const float* x, float* stats, long rows, int C, long ldx, float eps, void* stream) {
    CBX_REQUIRE(x && stats && C == 256 && ldx % 4 == 0, "row_stats: C must be 256 (got %d), ldx %% 4 == 0", C);
    if (rows <= 0) return 0;
    hipLaunchKernelGGL((layernorm_narrow_kernel<4, 1, true>), dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, (hipStream_t)stream, x, stats,
                       nullptr, nullptr, nullptr, rows, ldx, 0L, eps, 0, 0, 1.0f);
    return cbx_check_launch("row_stats");
}
