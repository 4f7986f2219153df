// Memory-bound glue kernels: strided activation / axpby / embedding gather / RoPE + KV-cache append / CFM Euler.
// All are grid-stride over float4 where the layout allows (16 B per lane, coalesced).
#include <stdarg.h>
#include "cbx_common.h"

thread_local char cbx_err_buf[512] = {0};

int cbx_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(cbx_err_buf, sizeof(cbx_err_buf), fmt, ap);
    va_end(ap);
    return code;
}

int cbx_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cbx_set_error((int)e, "%s: launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

extern "C" int cbx_abi_version(void) { return CBX_ABI_VERSION; }

// co-resident streams (cbx_common.h): a handful of registered stream handles; registration happens once, before the stream is used
static void* volatile g_cores_streams[16];
extern "C" int cbx_set_stream_coresident(void* stream, int on) {
    CBX_REQUIRE(stream != nullptr, "set_stream_coresident: the NULL stream cannot be marked (create a stream for the throughput schedule)");
    int free_slot = -1;
    for (int i = 0; i < 16; ++i) {
        if (g_cores_streams[i] == stream) {
            if (!on) g_cores_streams[i] = nullptr;
            return 0;
        }
        if (free_slot < 0 && g_cores_streams[i] == nullptr) free_slot = i;
    }
    if (!on) return 0;
    CBX_REQUIRE(free_slot >= 0, "set_stream_coresident: more than 16 co-resident streams");
    g_cores_streams[free_slot] = stream;
    return 0;
}
int cbx_stream_coresident(hipStream_t st) {
    if (!st) return 0;
    for (int i = 0; i < 16; ++i)
        if (g_cores_streams[i] == (void*)st) return 1;
    return 0;
}
extern "C" const char* cbx_last_error(void) { return cbx_err_buf; }

namespace {
CBX_TRC_TU

inline unsigned grid_for(long n, int per_block = 256) {
    long g = (n + per_block - 1) / per_block;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (unsigned)g;
}

__global__ void act_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ param, long rows,
                           int C, long ldx, long ldy, int act, float slope) {
    const int c4n = C >> 2;
    const long total = rows * c4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i / c4n;
        int c = (int)(i - r * c4n) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
        f32x4 pv = {0.f, 0.f, 0.f, 0.f};
        if (param) pv = *reinterpret_cast<const f32x4*>(param + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = cbx_act(v[e], act, slope, pv[e]);
        *reinterpret_cast<f32x4*>(y + r * ldy + c) = o;
    }
}

__global__ void axpby_kernel(const float* __restrict__ x, float* y, long rows, int C, long ldx, long ldy, float a, float b) {
    const long total = rows * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i / C;
        int c = (int)(i - r * C);
        float v = a * x[r * ldx + c];
        if (b != 0.f) v += b * y[r * ldy + c];
        y[r * ldy + c] = v;
    }
}

__global__ void embed_kernel(const long long* __restrict__ ids, const float* __restrict__ table,
                             const float* __restrict__ table2, const int* __restrict__ ids2, float* __restrict__ out,
                             long rows, int C, long ld_out, float scale, int flags) {
    CBX_TRC_DECL;
    CBX_TRC_STAMP(0);
    const int c4n = C >> 2;
    const long total = rows * c4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i / c4n;
        int c = (int)(i - r * c4n) * 4;
        long long id = ids[r];
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (id >= 0) v = *reinterpret_cast<const f32x4*>(table + id * C + c) * scale;
        if (table2 && ids2 && ids2[r] >= 0) v += *reinterpret_cast<const f32x4*>(table2 + (long)ids2[r] * C + c);
        long o = r * ld_out + c;
        if (flags & 2)  // packed GEMV operand layout (cbx.h): one float4 of the row = one float4 of the image
            o = ((((r >> 4) * (C >> 5) + (c >> 5)) * 2 + ((c >> 2) & 1)) * 64 + (((c >> 3) & 3) << 4) + (r & 15)) * 4;
        *reinterpret_cast<f32x4*>(out + o) = v;
    }
#ifdef CBX_TRACE
    CBX_TRC_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CBX_TRC_STAMP(2);
    CBX_TRC_FLUSH(0x40000000u);
#endif
}

// one 64-lane wave per (row, head): lane d<32 pairs with d+32 (rotate_half form)
__global__ __launch_bounds__(256) void rope_kv_kernel(float* qkv, const int* __restrict__ positions,
                                                      const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                      float* __restrict__ kc, float* __restrict__ vc,
                                                      const int* __restrict__ cache_rows, long n_rows, int n_heads,
                                                      long ld_qkv, long row_stride, long head_stride) {
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_rows * n_heads) return;
    const long r = w / n_heads;
    const int h = (int)(w - r * n_heads), d = threadIdx.x & 63;
    const int pos = positions[r];
    const float c = cos_t ? cos_t[(long)pos * 64 + d] : 1.f, s = cos_t ? sin_t[(long)pos * 64 + d] : 0.f;  // NULL tables: GPT-2, no RoPE
    float* qp = qkv + r * ld_qkv + h * 64;
    float* kp = qp + (long)n_heads * 64;
    const float* vp = kp + (long)n_heads * 64;
    const float qv = qp[d], kv = kp[d];
    const float qo = cbx_xor_lane<32>(qv), ko = cbx_xor_lane<32>(kv);
    const float sgn = d < 32 ? -1.f : 1.f;
    const float qn = qv * c + sgn * qo * s;
    const float kn = kv * c + sgn * ko * s;
    qp[d] = qn;
    kp[d] = kn;
    if (kc) {
        const long crow = cache_rows ? cache_rows[r] : r;
        const long off = crow * row_stride + (long)h * head_stride + (long)pos * 64 + d;
        kc[off] = kn;
        vc[off] = vp[d];
    }
}

__global__ void cfm_euler_kernel(float* xin, const float* __restrict__ v, int B, long T, int C, long ld_x, long ld_v,
                                 long xs_b, long vs_b, float dt, float w, int cfg) {
    const long total = (long)B * T * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long t = (i / C) % T;
        int b = (int)(i / ((long)C * T));
        float vc = v[b * vs_b + t * ld_v + c];
        float dx = cfg ? ((1.0f + w) * vc - w * v[(long)(B + b) * vs_b + t * ld_v + c]) : vc;
        float* xp = xin + b * xs_b + t * ld_x + c;
        float xn = *xp + dt * dx;
        *xp = xn;
        if (cfg) xin[(long)(B + b) * xs_b + t * ld_x + c] = xn;
    }
}

}  // namespace
CBX_TRC_SETTER(cbx_trace_set_elementwise)

extern "C" int cbx_act_f32(const float* x, float* y, const float* param, long rows, int C, long ldx, long ldy, int act,
                           float slope, void* stream) {
    CBX_REQUIRE(x && y && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "act: bad args C=%d", C);
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(act_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, y, param, rows, C,
                       ldx, ldy, act, slope);
    return cbx_check_launch("act");
}

extern "C" int cbx_axpby_f32(const float* x, float* y, long rows, int C, long ldx, long ldy, float a, float b, void* stream) {
    CBX_REQUIRE(x && y, "axpby: null");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, x, y, rows, C, ldx, ldy, a, b);
    return cbx_check_launch("axpby");
}

extern "C" int cbx_embed_f32(const long long* ids, const float* table, const float* table2, const int* ids2, float* out,
                             long rows, int C, long ld_out, float scale, int flags, void* stream) {
    CBX_REQUIRE(ids && table && out && C % 4 == 0 && ld_out % 4 == 0 && (!(flags & 2) || C % 32 == 0), "embed: bad args");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(embed_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, ids, table, table2,
                       ids2, out, rows, C, ld_out, scale, flags);
    return cbx_check_launch("embed");
}

extern "C" int cbx_rope_kv_f32(float* qkv, const int* positions, const float* cos_t, const float* sin_t, float* kc,
                               float* vc, const int* cache_rows, long n_rows, int n_heads, long ld_qkv,
                               long cache_row_stride, long cache_head_stride, void* stream) {
    CBX_REQUIRE(qkv && positions && (!cos_t == !sin_t), "rope_kv: null");
    if (n_rows <= 0) return 0;
    long waves = n_rows * n_heads;
    hipLaunchKernelGGL(rope_kv_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, qkv, positions,
                       cos_t, sin_t, kc, vc, cache_rows, n_rows, n_heads, ld_qkv, cache_row_stride, cache_head_stride);
    return cbx_check_launch("rope_kv");
}

extern "C" int cbx_cfm_euler_f32(float* xin, const float* v, int B, long T, int C, long ld_x, long ld_v, long xs_b,
                                 long vs_b, float dt, float w, int cfg, void* stream) {
    CBX_REQUIRE(xin && v && B > 0 && T > 0, "cfm_euler: bad args");
    hipLaunchKernelGGL(cfm_euler_kernel, dim3(grid_for((long)B * T * C)), dim3(256), 0, (hipStream_t)stream, xin, v, B, T, C,
                       ld_x, ld_v, xs_b, vs_b, dt, w, cfg);
    return cbx_check_launch("cfm_euler");
}
