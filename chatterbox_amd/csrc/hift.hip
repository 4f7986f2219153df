// HiFT source and (i)STFT kernels (reference: models/s3gen/hifigan.py:201-231, 267-283, 396-410).
//
//  * source: nearest-upsampled F0 -> 9 harmonics -> tanh(Linear) excitation.  The reference's
//    `cumsum(f0*(h+1)/sr) % 1` runs on CPU with a double accumulator rounded to fp32 per sample; F0 is piecewise
//    constant per mel frame, so the same value is reproduced with a tiny fp64 scan over FRAMES plus a closed form
//    inside the frame -- exact parity without a 240k-long serial fp32 chain.
//  * STFT(16, hop 4, hann, center/reflect) of the excitation as a direct 16-point DFT per frame.
//  * conv_post head -> magnitude/phase -> inverse DFT + windowed overlap-add + envelope normalisation + clamp,
//    fused; spectra of the ~68 frames a 256-sample block touches are converted once into LDS.
#include "cbx_common.h"

namespace {

constexpr float TWO_PI_F = 6.2831855f;  // fl32(2*pi), the scalar torch multiplies with

__device__ __forceinline__ float harmonic_inc(float f0, int h, float sr) {
    float t = f0 * (float)(h + 1);  // F_mat = f0 * (i+1) / sr, both steps rounded to fp32 (hifigan.py:209)
    return t / sr;
}

// exclusive scan over frames of up * F[b][h][frame], in fp64: one lane per (b, h)
__global__ void source_scan_kernel(const float* __restrict__ f0, double* __restrict__ frame_cum, int B, int T, int up,
                                   float sr) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 9) return;
    int b = i / 9, h = i - b * 9;
    const float* fr = f0 + (long)b * T;
    double* out = frame_cum + (long)i * T;
    double acc = 0.0;
    for (int t = 0; t < T; ++t) {
        out[t] = acc;
        acc += (double)up * (double)harmonic_inc(fr[t], h, sr);
    }
}

__global__ void source_kernel(const float* __restrict__ f0, const float* __restrict__ phase, const float* __restrict__ noise,
                              const float* __restrict__ lin_w, float lin_b, float* __restrict__ s,
                              const double* __restrict__ frame_cum, int B, int T, int up, float sr) {
    const long L = (long)T * up;
    const long total = (long)B * L;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int b = (int)(i / L);
        long n = i - (long)b * L;
        int fr = (int)(n / up), k = (int)(n - (long)fr * up);
        float f = f0[(long)b * T + fr];
        float uv = f > 10.0f ? 1.0f : 0.0f;
        float amp = uv * 0.003f + (1.0f - uv) * 0.1f / 3.0f;
        float acc = lin_b;
#pragma unroll
        for (int h = 0; h < 9; ++h) {
            float F = harmonic_inc(f, h, sr);
            double c = frame_cum[((long)b * 9 + h) * T + fr] + (double)(k + 1) * (double)F;
            float cf = (float)c;
            float frac = cf - floorf(cf);
            float theta = TWO_PI_F * frac;
            float sw = 0.1f * sinf(theta + phase[b * 9 + h]);
            sw = sw * uv + amp * noise[((long)b * 9 + h) * L + n];
            acc += lin_w[h] * sw;
        }
        s[i] = tanhf(acc);
    }
}

__constant__ float c_cos16[16] = {1.0f, 0.92387953f, 0.70710678f, 0.38268343f, 0.0f, -0.38268343f, -0.70710678f, -0.92387953f,
                                  -1.0f, -0.92387953f, -0.70710678f, -0.38268343f, 0.0f, 0.38268343f, 0.70710678f, 0.92387953f};
__constant__ float c_sin16[16] = {0.0f, 0.38268343f, 0.70710678f, 0.92387953f, 1.0f, 0.92387953f, 0.70710678f, 0.38268343f,
                                  0.0f, -0.38268343f, -0.70710678f, -0.92387953f, -1.0f, -0.92387953f, -0.70710678f, -0.38268343f};
// periodic hann(16): 0.5 - 0.5 cos(2 pi n / 16)
__constant__ float c_hann16[16] = {0.0f, 0.03806023f, 0.14644661f, 0.30865828f, 0.5f, 0.69134172f, 0.85355339f, 0.96193977f,
                                   1.0f, 0.96193977f, 0.85355339f, 0.69134172f, 0.5f, 0.30865828f, 0.14644661f, 0.03806023f};

// spec[b][t][0..8] = Re, [9..17] = Im, [18..ld) = 0
__global__ void stft_kernel(const float* __restrict__ s, float* __restrict__ spec, const int* __restrict__ sample_lens, int B,
                            long L, long frames, long ld) {
    const long total = (long)B * frames;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int b = (int)(i / frames);
        long t = i - (long)b * frames;
        const float* x = s + (long)b * L;
        const long Lb = sample_lens ? min((long)sample_lens[b], L) : L;  // ragged batch: reflect at the row's own end
        float xw[16];
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            long q = 4 * t + n - 8;  // reflect padding (center=True)
            if (q < 0) q = -q;
            if (q >= Lb) q = 2 * (Lb - 1) - q;
            if (q < 0) q = 0;
            xw[n] = x[q] * c_hann16[n];
        }
        float* o = spec + i * ld;
#pragma unroll
        for (int f = 0; f < 9; ++f) {
            float re = 0.f, im = 0.f;
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                re += xw[n] * c_cos16[(f * n) & 15];
                im -= xw[n] * c_sin16[(f * n) & 15];
            }
            o[f] = re;
            o[9 + f] = im;
        }
        for (int c = 18; c < ld; ++c) o[c] = 0.f;
    }
}

constexpr int IS_BLK = 256;
constexpr int IS_FR = IS_BLK / 4 + 5;

__global__ __launch_bounds__(IS_BLK) void istft_kernel(const float* __restrict__ x, float* __restrict__ wav, long frames,
                                                      long ldx, float clampv, int fade_n) {
    __shared__ float re[IS_FR][9];
    __shared__ float im[IS_FR][9];
    const int b = blockIdx.y;
    const long out_len = 4 * (frames - 1);
    const long m0 = (long)blockIdx.x * IS_BLK;
    // frames touched by padded positions q in [m0+8, m0+8+255]: t in [ceil((q-15)/4), floor(q/4)]
    long t_lo = (m0 + 8 - 15 + 3) / 4;
    if (m0 + 8 - 15 < 0) t_lo = 0;
    const int tid = threadIdx.x;
    if (tid < IS_FR) {
        long t = t_lo + tid;
        if (t < frames) {
            const float* xr = x + ((long)b * frames + t) * ldx;
#pragma unroll
            for (int f = 0; f < 9; ++f) {
                float mag = fminf(expf(xr[f]), 100.0f);
                float ph = sinf(xr[9 + f]);
                re[tid][f] = mag * cosf(ph);
                im[tid][f] = mag * sinf(ph);
            }
        }
    }
    __syncthreads();
    const long m = m0 + tid;
    if (m >= out_len) return;
    const long q = m + 8;
    float acc = 0.f, env = 0.f;
    long t_hi = q / 4;
    if (t_hi > frames - 1) t_hi = frames - 1;
    long t_first = (q - 15 + 3) / 4;
    if (q - 15 < 0) t_first = 0;
    for (long t = t_first; t <= t_hi; ++t) {
        int n = (int)(q - 4 * t);
        int lt = (int)(t - t_lo);
        float y = re[lt][0] + ((n & 1) ? -re[lt][8] : re[lt][8]);
#pragma unroll
        for (int f = 1; f < 8; ++f) y += 2.0f * (re[lt][f] * c_cos16[(f * n) & 15] - im[lt][f] * c_sin16[(f * n) & 15]);
        y *= (1.0f / 16.0f);
        float w = c_hann16[n];
        acc += y * w;
        env += w * w;
    }
    float v = acc / env;
    v = fminf(fmaxf(v, -clampv), clampv);
    if (fade_n > 0 && m < 2 * fade_n) {  // S3Token2Wav.trim_fade (s3gen.py:255-258,360)
        float g = 0.f;
        if (m >= fade_n) {
            float ang = 3.14159265358979f * (1.0f - (float)(m - fade_n) / (float)(fade_n - 1));
            g = (cosf(ang) + 1.0f) * 0.5f;
        }
        v *= g;
    }
    wav[(long)b * out_len + m] = v;
}

}  // namespace

extern "C" int cbx_hift_source_f32(const float* f0, const float* phase, const float* noise, const float* lin_w, float lin_b,
                                   float* s, double* frame_cum, int B, int T, int up, float sr, void* stream) {
    CBX_REQUIRE(f0 && phase && noise && lin_w && s && frame_cum && B > 0 && T > 0, "hift_source: bad args");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(source_scan_kernel, dim3((B * 9 + 63) / 64), dim3(64), 0, st, f0, frame_cum, B, T, up, sr);
    long total = (long)B * T * up;
    unsigned grid = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(source_kernel, dim3(grid), dim3(256), 0, st, f0, phase, noise, lin_w, lin_b, s, frame_cum, B, T, up, sr);
    return cbx_check_launch("hift_source");
}

extern "C" int cbx_hift_stft_f32(const float* s, float* spec, const int* sample_lens, int B, long L, long ld_spec,
                                 void* stream) {
    CBX_REQUIRE(s && spec && B > 0 && L >= 16 && L % 4 == 0 && ld_spec >= 18, "hift_stft: bad args");
    long frames = L / 4 + 1, total = (long)B * frames;
    unsigned grid = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(stft_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, s, spec, sample_lens, B, L, frames, ld_spec);
    return cbx_check_launch("hift_stft");
}

extern "C" int cbx_hift_istft_f32(const float* x, float* wav, int B, long frames, long ldx, float clampv, int fade_n,
                                  void* stream) {
    CBX_REQUIRE(x && wav && B > 0 && frames >= 2 && ldx >= 18, "hift_istft: bad args");
    long out_len = 4 * (frames - 1);
    dim3 grid((unsigned)((out_len + IS_BLK - 1) / IS_BLK), B);
    hipLaunchKernelGGL(istft_kernel, grid, dim3(IS_BLK), 0, (hipStream_t)stream, x, wav, frames, ldx, clampv, fade_n);
    return cbx_check_launch("hift_istft");
}
