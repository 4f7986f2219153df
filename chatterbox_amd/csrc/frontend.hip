// Small kernels of the voice-prompt / voice-conversion front-end (S3 tokenizer, CAMPPlus x-vector, voice encoder, mel
// extractors).  The contractions of these stages (framed DFT as a GEMM over overlapping rows, mel filterbanks, convolutions,
// attention, LSTM input / recurrent projections) run on the shared MFMA kernels (gemm_f32.hip, attention.hip, gemv_decode.hip);
// what is left are HBM-bound element-wise / reduction passes over channel-last (rows = time, channels) activations.
#include "cbx_common.h"

namespace {

__device__ __forceinline__ float block_reduce_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += red[i];
    return r;
}
__device__ __forceinline__ float block_reduce_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = fmaxf(r, red[i]);
    return r;
}

// y[b][t][c] = sum_k w[c][k] * x[b][t + k - pad_left][c]  (+ x[b][t][c]);  rows t >= lens[b] (or outside [0, T)) read as zero and are
// written as zero.  One thread per (t, 4 channels).
__global__ void dwconv1d_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                const int* __restrict__ lens, int B, int T, int C, int taps, int pad_left, long ldx, long ldy, long xsb,
                                long ysb, int add_input) {
    const int c4n = C >> 2;
    const long total = (long)B * T * c4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const long bt = i / c4n;
        const int t = (int)(bt % T), b = (int)(bt / T);
        const int lim = lens ? min(T, lens[b]) : T;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (t < lim) {
            const float* xb = x + (long)b * xsb + c;
            for (int k = 0; k < taps; ++k) {
                const int tt = t + k - pad_left;
                if (tt < 0 || tt >= lim) continue;
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xb + (long)tt * ldx);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += w[(long)(c + e) * taps + k] * xv[e];
            }
            if (add_input) acc += *reinterpret_cast<const f32x4*>(xb + (long)t * ldx);
        }
        *reinterpret_cast<f32x4*>(y + (long)b * ysb + (long)t * ldy + c) = acc;
    }
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// torch.nn.LSTM cell, gate order (i, f, g, o):  gates = pre + hh;  c = sig(f) c + sig(i) tanh(g);  h = sig(o) tanh(c)
__global__ void lstm_cell_kernel(const float* __restrict__ pre, const float* __restrict__ hh, float* __restrict__ c, float* __restrict__ h,
                                 int B, int H, long ld_pre, long ld_hh, long ldc, long ldh) {
    const long total = (long)B * H;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i % H), b = (int)(i / H);
        const float* p = pre + (long)b * ld_pre;
        const float* q = hh + (long)b * ld_hh;
        const float gi = p[j] + q[j], gf = p[H + j] + q[H + j], gg = p[2 * H + j] + q[2 * H + j], go = p[3 * H + j] + q[3 * H + j];
        const float cn = sigmoidf_(gf) * c[(long)b * ldc + j] + sigmoidf_(gi) * tanhf(gg);
        c[(long)b * ldc + j] = cn;
        h[(long)b * ldh + j] = sigmoidf_(go) * tanhf(cn);
    }
}

// y = act(x * scale[c] + shift[c])  (eval-mode BatchNorm folded to an affine map, followed by ReLU in CAMPPlus)
__global__ void affine_act_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ scale,
                                  const float* __restrict__ shift, long rows, int C, long ldx, long ldy, int act) {
    const int c4n = C >> 2;
    const long total = rows * c4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c4n;
        const int c = (int)(i - r * c4n) * 4;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = cbx_act(xv[e] * sc[e] + sh[e], act, 0.f, 0.f);
        *reinterpret_cast<f32x4*>(y + r * ldy + c) = o;
    }
}

// spec row = [re(0..F-1) | im(0..F-1)]  ->  out[f] = re^2 + im^2 (mode 0) | sqrt(re^2 + im^2 + eps) (mode 1)
__global__ void cplx_power_kernel(const float* __restrict__ spec, float* __restrict__ out, long rows, int F, long ld_spec, long ld_out,
                                  int mode, float eps) {
    const long total = rows * F;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / F;
        const int f = (int)(i - r * F);
        const float re = spec[r * ld_spec + f], im = spec[r * ld_spec + F + f];
        const float p = re * re + im * im;
        out[r * ld_out + f] = mode ? sqrtf(p + eps) : p;
    }
}

// element-wise maps of the log-mel front-ends
__global__ void unary_kernel(const float* __restrict__ x, float* __restrict__ y, long rows, int C, long ldx, long ldy, int op, float a,
                             float b, const float* __restrict__ dev_scalar) {
    const long total = rows * C;
    const float g = dev_scalar ? *dev_scalar : 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        const float v = x[r * ldx + c];
        float o;
        switch (op) {
            case CBX_UN_LOG_CLAMP: o = logf(fmaxf(v, a)); break;                    // log(clamp(x, min = a))
            case CBX_UN_LOG10_CLAMP: o = log10f(fmaxf(v, a)); break;                // log10(clamp(x, min = a))
            case CBX_UN_FLOOR_AFFINE: o = (fmaxf(v, g - a) + b) / b; break;         // (max(x, gmax - a) + b) / b   (s3tokenizer.py:164-166)
            case CBX_UN_AFFINE: o = v * a + b; break;
            default: o = v;
        }
        y[r * ldy + c] = o;
    }
}

__global__ __launch_bounds__(1024) void reduce_max_kernel(const float* __restrict__ x, float* __restrict__ out, long rows, int C, long ldx) {
    __shared__ float red[16];
    float m = -INFINITY;
    const long total = rows * C;
    for (long i = threadIdx.x; i < total; i += 1024) {
        const long r = i / C;
        m = fmaxf(m, x[r * ldx + (i - r * C)]);
    }
    m = block_reduce_max(m, red);
    if (threadIdx.x == 0) *out = m;
}

// CAMPPlus context: ctx[s][c] = mean_t x[t][c] + mean_{t in segment s} x[t][c]   (xvector.py:215-231; avg_pool1d with ceil_mode:
// the last window averages over the frames it actually contains).  One workgroup per 4 channels, 256 threads over time.
__global__ __launch_bounds__(256) void seg_context_kernel(const float* __restrict__ x, float* __restrict__ ctx, int T, int C, int seg_len,
                                                          long ldx, long ldc) {
    extern __shared__ float seg_sum[];  // [n_seg][4]
    __shared__ float red[4];
    const int c0 = blockIdx.x * 4, n_seg = (T + seg_len - 1) / seg_len;
    for (int i = threadIdx.x; i < n_seg * 4; i += 256) seg_sum[i] = 0.f;
    __syncthreads();
    // every wave owns whole segments so that each segment sum is accumulated in a fixed order
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int s = wv; s < n_seg; s += 4) {
        const int t0 = s * seg_len, t1 = min(T, t0 + seg_len);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int t = t0 + lane; t < t1; t += 64) acc += *reinterpret_cast<const f32x4*>(x + (long)t * ldx + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = wave_sum(acc[e]);
            if (lane == 0) seg_sum[s * 4 + e] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        float tot = 0.f;
        for (int s = 0; s < n_seg; ++s) tot += seg_sum[s * 4 + threadIdx.x];
        red[threadIdx.x] = tot / (float)T;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_seg * 4; i += 256) {
        const int s = i >> 2, e = i & 3;
        const int cnt = min(T, (s + 1) * seg_len) - s * seg_len;
        ctx[(long)s * ldc + c0 + e] = red[e] + seg_sum[i] / (float)cnt;
    }
}

// y[t][c] *= sigmoid(m[t / seg_len][c])   (CAMLayer gating, xvector.py:209-213)
__global__ void seg_gate_mul_kernel(float* __restrict__ y, const float* __restrict__ m, int T, int C, int seg_len, long ldy, long ldm) {
    const int c4n = C >> 2;
    const long total = (long)T * c4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)(i / c4n), c = (int)(i % c4n) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(y + (long)t * ldy + c);
        const f32x4 g = *reinterpret_cast<const f32x4*>(m + (long)(t / seg_len) * ldm + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= sigmoidf_(g[e]);
        *reinterpret_cast<f32x4*>(y + (long)t * ldy + c) = v;
    }
}

// StatsPool: out[c] = mean_t x[t][c], out[C + c] = unbiased std_t x[t][c]  (two-pass).  One workgroup per channel.
__global__ __launch_bounds__(256) void stats_pool_kernel(const float* __restrict__ x, float* __restrict__ out, int T, int C, long ldx) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    float s = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) s += x[(long)t * ldx + c];
    const float mean = block_reduce_sum(s, red) / (float)T;
    float v = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) {
        const float d = x[(long)t * ldx + c] - mean;
        v += d * d;
    }
    v = block_reduce_sum(v, red);
    if (threadIdx.x == 0) {
        out[c] = mean;
        out[C + c] = sqrtf(v / (float)(T - 1));
    }
}

// FSQ codebook index (S3TokenizerV2): h[t][0..7] -> sum_d (round(tanh(h_d) * 0.999) + 1) * 3^d
__global__ void fsq_index_kernel(const float* __restrict__ h, long long* __restrict__ idx, long rows, long ldh) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    long long code = 0, p = 1;
    for (int d = 0; d < 8; ++d) {
        const float q = rintf(tanhf(h[r * ldh + d]) * 0.9990000128746033f) + 1.0f;  // torch.round = round-half-to-even
        code += (long long)q * p;
        p *= 3;
    }
    idx[r] = code;
}

inline unsigned grid_for(long n) {
    long g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 65535 ? 65535 : g));
}

}  // namespace

extern "C" int cbx_dwconv1d_f32(const float* x, const float* w, float* y, const int* lens, int B, int T, int C, int taps, int pad_left,
                                long ldx, long ldy, long x_sb, long y_sb, int add_input, void* stream) {
    CBX_REQUIRE(x && w && y && B > 0 && T > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && taps > 0, "dwconv1d: bad args");
    hipLaunchKernelGGL(dwconv1d_kernel, dim3(grid_for((long)B * T * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, w, y, lens, B, T, C,
                       taps, pad_left, ldx, ldy, x_sb, y_sb, add_input);
    return cbx_check_launch("dwconv1d");
}

extern "C" int cbx_lstm_cell_f32(const float* pre, const float* hh, float* c, float* h, int B, int H, long ld_pre, long ld_hh, long ldc,
                                 long ldh, void* stream) {
    CBX_REQUIRE(pre && hh && c && h && B > 0 && H > 0, "lstm_cell: bad args");
    hipLaunchKernelGGL(lstm_cell_kernel, dim3(grid_for((long)B * H)), dim3(256), 0, (hipStream_t)stream, pre, hh, c, h, B, H, ld_pre,
                       ld_hh, ldc, ldh);
    return cbx_check_launch("lstm_cell");
}

extern "C" int cbx_affine_act_f32(const float* x, float* y, const float* scale, const float* shift, long rows, int C, long ldx, long ldy,
                                  int act, void* stream) {
    CBX_REQUIRE(x && y && scale && shift && rows > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "affine_act: bad args");
    hipLaunchKernelGGL(affine_act_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, y, scale, shift, rows, C,
                       ldx, ldy, act);
    return cbx_check_launch("affine_act");
}

extern "C" int cbx_cplx_power_f32(const float* spec, float* out, long rows, int F, long ld_spec, long ld_out, int mode, float eps,
                                  void* stream) {
    CBX_REQUIRE(spec && out && rows > 0 && F > 0, "cplx_power: bad args");
    hipLaunchKernelGGL(cplx_power_kernel, dim3(grid_for(rows * F)), dim3(256), 0, (hipStream_t)stream, spec, out, rows, F, ld_spec,
                       ld_out, mode, eps);
    return cbx_check_launch("cplx_power");
}

extern "C" int cbx_unary_f32(const float* x, float* y, long rows, int C, long ldx, long ldy, int op, float a, float b,
                             const float* dev_scalar, void* stream) {
    CBX_REQUIRE(x && y && rows > 0 && C > 0 && (op != CBX_UN_FLOOR_AFFINE || dev_scalar), "unary: bad args");
    hipLaunchKernelGGL(unary_kernel, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, x, y, rows, C, ldx, ldy, op, a, b,
                       dev_scalar);
    return cbx_check_launch("unary");
}

extern "C" int cbx_reduce_max_f32(const float* x, float* out, long rows, int C, long ldx, void* stream) {
    CBX_REQUIRE(x && out && rows > 0 && C > 0, "reduce_max: bad args");
    hipLaunchKernelGGL(reduce_max_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, out, rows, C, ldx);
    return cbx_check_launch("reduce_max");
}

extern "C" int cbx_seg_context_f32(const float* x, float* ctx, int T, int C, int seg_len, long ldx, long ldc, void* stream) {
    CBX_REQUIRE(x && ctx && T > 0 && C % 4 == 0 && seg_len > 0 && ldx % 4 == 0, "seg_context: bad args");
    const int n_seg = (T + seg_len - 1) / seg_len;
    CBX_REQUIRE((size_t)n_seg * 16 <= 48 * 1024, "seg_context: too many segments");
    hipLaunchKernelGGL(seg_context_kernel, dim3(C / 4), dim3(256), (size_t)n_seg * 16, (hipStream_t)stream, x, ctx, T, C, seg_len, ldx,
                       ldc);
    return cbx_check_launch("seg_context");
}

extern "C" int cbx_seg_gate_mul_f32(float* y, const float* m, int T, int C, int seg_len, long ldy, long ldm, void* stream) {
    CBX_REQUIRE(y && m && T > 0 && C % 4 == 0 && seg_len > 0 && ldy % 4 == 0 && ldm % 4 == 0, "seg_gate_mul: bad args");
    hipLaunchKernelGGL(seg_gate_mul_kernel, dim3(grid_for((long)T * (C / 4))), dim3(256), 0, (hipStream_t)stream, y, m, T, C, seg_len, ldy,
                       ldm);
    return cbx_check_launch("seg_gate_mul");
}

extern "C" int cbx_stats_pool_f32(const float* x, float* out, int T, int C, long ldx, void* stream) {
    CBX_REQUIRE(x && out && T > 1 && C > 0, "stats_pool: bad args (T >= 2)");
    hipLaunchKernelGGL(stats_pool_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, out, T, C, ldx);
    return cbx_check_launch("stats_pool");
}

extern "C" int cbx_fsq_index(const float* h, long long* idx, long rows, long ldh, void* stream) {
    CBX_REQUIRE(h && idx && rows > 0 && ldh >= 8, "fsq_index: bad args");
    hipLaunchKernelGGL(fsq_index_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h, idx, rows, ldh);
    return cbx_check_launch("fsq_index");
}
