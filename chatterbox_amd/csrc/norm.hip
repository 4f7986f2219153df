// LayerNorm / RMSNorm with fused activation: one 64-lane wavefront per row, row held in registers,
// two-pass statistics by wave shuffles (no LDS, no atomics).  HBM-bound: one read + one write per element.
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

}  // namespace

extern "C" int cbx_layernorm_f32(const float* x, float* y, const float* w, const float* b, const float* post_add,
                                 long rows, int C, long ldx, long ldy, float eps, int rms, int act, float out_scale,
                                 void* stream) {
    CBX_REQUIRE(x && y && w, "layernorm: null operand");
    CBX_REQUIRE(C % 4 == 0 && C <= 4096 && ldx % 4 == 0 && ldy % 4 == 0, "layernorm: C=%d ldx=%ld ldy=%ld", C, ldx, ldy);
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, y, w, b,
                       post_add, rows, C, ldx, ldy, eps, rms, act, out_scale);
    return cbx_check_launch("layernorm");
}
