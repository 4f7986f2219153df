// Stand-alone HIP program (hipcc --offload-arch=gfx950 -O2 mall_prefetch.hip -o mall_prefetch): does warming the 256 MiB Infinity Cache from a PARALLEL hipGraph
// branch shorten a chain of dependent weight-streaming launches?  (the T3 decode step is 152 dependent launches of ~4.2 us fixed cost + bytes / 5.5 TB/s;
// HBM idles during every launch boundary and ramp.  A side branch that touches the NEXT launches' weights while the current one runs would let the chain read them from the
// Infinity Cache -- if that is faster than HBM, and if the side branch does not slow the chain down.)
//
// chain:    LAYERS x 5 dependent launches shaped like the decode layer (q|k|v 12.6 MB, KV 29.7 MB, o 4.2 MB, gate|up 33.5 MB, down 16.8 MB), each 256 workgroups of 512
//           threads streaming its slice with non-temporal 16-byte loads (what gemv_decode.hip does), DISTINCT buffers per layer (2.9 GB: nothing is cache-resident by accident);
// "hot":    the same chain cycling over ONE layer's buffers (97 MB: beyond the 32 MiB of L2, inside the Infinity Cache) = what a perfect warm-up could give;
// prefetch: a second captured stream; launch P(j + D) is forked when chain launch j - 1 has finished (it runs beside launch j) and touches one dword per 128-byte line of the
//           buffer launch j + D will stream (default cache policy), on PG workgroups; the chain never waits for it (joined at the end of the graph only).
// Prints one line per variant: us per chain launch, ms per 150-launch "token".
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int T, int U>
__global__ __launch_bounds__(T) void stream_kernel_t(const f32x4* __restrict__ w, long n4_per_wg, float* __restrict__ out, const float* __restrict__ dep) {
    const f32x4* p = w + (long)blockIdx.x * n4_per_wg;
    f32x4 acc[U] = {};
    const float d = dep[threadIdx.x & 255];  // the previous launch's output: a true data dependency, like the activations
    long i = threadIdx.x;
    for (; i + (U - 1) * T < n4_per_wg; i += U * T) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + i + u * T);
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] += v[u] * d;
    }
    for (; i < n4_per_wg; i += T) acc[0] += __builtin_nontemporal_load(p + i) * d;
    f32x4 s = acc[0];
#pragma unroll
    for (int u = 1; u < U; ++u) s += acc[u];
    float r = s.x + s.y + s.z + s.w;
    for (int o = 32; o; o >>= 1) r += __shfl_xor(r, o);
    __shared__ float part[T / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = r;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < T / 64; ++k) t += part[k];
        out[blockIdx.x & 255] = t * 1e-9f;
    }
}
#define stream_kernel stream_kernel_t<512, 4>

// one dword per 128-byte line, default policy: the line is fetched into (the prefetcher's XCD's) L2 and the Infinity Cache
__global__ __launch_bounds__(256) void touch_kernel(const float* __restrict__ w, long lines, float* __restrict__ sink) {
    float a = 0.f;
    for (long l = (long)blockIdx.x * 256 + threadIdx.x; l < lines; l += (long)gridDim.x * 256) a += w[l * 32];
    if (a == 123.456f) sink[0] = a;
}

// XCD-matched form: 256 workgroups, workgroup b touches the first `lines_per_wg` lines of the slice chain workgroup b will stream (same blockIdx -> same XCD -> same L2)
__global__ __launch_bounds__(256) void touch_matched_kernel(const float* __restrict__ w, long slice_lines, long lines_per_wg, float* __restrict__ sink) {
    const float* p = w + (long)blockIdx.x * slice_lines * 32;
    float a = 0.f;
    for (long l = threadIdx.x; l < lines_per_wg; l += 256) a += p[l * 32];
    if (a == 123.456f) sink[0] = a;
}

struct Op { const char* name; long bytes; };
static const Op OPS[5] = {{"qkv", 12582912}, {"attn-kv", 29753344}, {"o", 4194304}, {"gate|up", 33554432}, {"down", 16777216}};

int main(int argc, char** argv) {
    const int LAYERS = 30, REPLAYS = argc > 1 ? atoi(argv[1]) : 20;
    hipStream_t s0, s1;
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&s0, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, lo));
    std::vector<float*> buf(LAYERS * 5);
    for (int l = 0; l < LAYERS; ++l)
        for (int o = 0; o < 5; ++o) {
            CK(hipMalloc(&buf[l * 5 + o], OPS[o].bytes));
            CK(hipMemsetAsync(buf[l * 5 + o], 0, OPS[o].bytes, s0));
        }
    float *act[2], *sink;
    CK(hipMalloc(&act[0], 4096)); CK(hipMalloc(&act[1], 4096)); CK(hipMalloc(&sink, 4096));
    CK(hipMemsetAsync(act[0], 0, 4096, s0)); CK(hipMemsetAsync(act[1], 0, 4096, s0));
    CK(hipStreamSynchronize(s0));
    const int N = LAYERS * 5;
    std::vector<hipEvent_t> ev(N + 2);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));

    // variant: hot = cycle over layer 0's buffers; D = prefetch distance in launches (0: no side branch); PG = prefetch workgroups (-1: the XCD-matched form); frac256 = part of the buffer touched (/256)
    auto run = [&](const char* name, bool hot, int D, int PG, int frac256) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
        bool forked = false;
        for (int j = 0; j < N; ++j) {
            const int o = j % 5, b = hot ? o : j;
            if (D > 0 && j + D < N) {  // fork P(j + D): after launch j - 1 (the capture point reached so far), beside launch j
                CK(hipEventRecord(ev[j], s0));
                CK(hipStreamWaitEvent(s1, ev[j], 0));
                const int oo = (j + D) % 5, bb = hot ? oo : j + D;
                const long lines = OPS[oo].bytes / 128 * frac256 / 256;
                if (PG < 0) hipLaunchKernelGGL(touch_matched_kernel, dim3(256), dim3(256), 0, s1, buf[bb], OPS[oo].bytes / 128 / 256, OPS[oo].bytes / 128 / 256 * frac256 / 256, sink);
                else hipLaunchKernelGGL(touch_kernel, dim3(PG), dim3(256), 0, s1, buf[bb], lines, sink);
                forked = true;
            }
            hipLaunchKernelGGL(stream_kernel, dim3(256), dim3(512), 0, s0, reinterpret_cast<const f32x4*>(buf[b]), OPS[o].bytes / 16 / 256, act[(j + 1) & 1], act[j & 1]);
        }
        if (forked) { CK(hipEventRecord(ev[N], s1)); CK(hipStreamWaitEvent(s0, ev[N], 0)); }
        CK(hipStreamEndCapture(s0, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s0));
        CK(hipStreamSynchronize(s0));
        float best = 1e30f, sum = 0.f;
        for (int k = 0; k < 3; ++k) {
            CK(hipEventRecord(t0, s0));
            for (int r = 0; r < REPLAYS; ++r) CK(hipGraphLaunch(ge, s0));
            CK(hipEventRecord(t1, s0));
            CK(hipStreamSynchronize(s0));
            float ms; CK(hipEventElapsedTime(&ms, t0, t1));
            sum += ms; if (ms < best) best = ms;
        }
        printf("%-64s %6.2f us / launch   %.3f ms / 150-launch token   (best of 3: %.3f)\n", name, sum / 3 / REPLAYS / N * 1e3, sum / 3 / REPLAYS, best / REPLAYS);
        fflush(stdout);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    };
    // per-op floors: a chain of N launches of ONE op's size over distinct buffers (every buffer at least that large; the cycle is far beyond the Infinity Cache), per geometry
    auto run_op = [&](int o, int geom) {
        std::vector<float*> pool;
        for (int j = 0; j < N; ++j) if (OPS[j % 5].bytes >= OPS[o].bytes) pool.push_back(buf[j]);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
        for (int j = 0; j < N; ++j) {
            const f32x4* w = reinterpret_cast<const f32x4*>(pool[j % pool.size()]);
            const long n4 = OPS[o].bytes / 16;
            float *dst = act[(j + 1) & 1], *src = act[j & 1];
            switch (geom) {
                case 0: hipLaunchKernelGGL((stream_kernel_t<512, 4>), dim3(256), dim3(512), 0, s0, w, n4 / 256, dst, src); break;
                case 1: hipLaunchKernelGGL((stream_kernel_t<1024, 4>), dim3(256), dim3(1024), 0, s0, w, n4 / 256, dst, src); break;
                case 2: hipLaunchKernelGGL((stream_kernel_t<512, 4>), dim3(512), dim3(512), 0, s0, w, n4 / 512, dst, src); break;
                case 3: hipLaunchKernelGGL((stream_kernel_t<512, 8>), dim3(256), dim3(512), 0, s0, w, n4 / 256, dst, src); break;
                case 4: hipLaunchKernelGGL((stream_kernel_t<256, 8>), dim3(1024), dim3(256), 0, s0, w, n4 / 1024, dst, src); break;
                default: hipLaunchKernelGGL((stream_kernel_t<256, 4>), dim3(2048), dim3(256), 0, s0, w, n4 / 2048, dst, src); break;
            }
        }
        CK(hipStreamEndCapture(s0, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s0));
        CK(hipStreamSynchronize(s0));
        CK(hipEventRecord(t0, s0));
        for (int r = 0; r < REPLAYS; ++r) CK(hipGraphLaunch(ge, s0));
        CK(hipEventRecord(t1, s0));
        CK(hipStreamSynchronize(s0));
        float ms; CK(hipEventElapsedTime(&ms, t0, t1));
        const double us = ms * 1e3 / REPLAYS / N;
        static const char* GN[] = {"256 wg x 512 thr, 4 loads in flight", "256 wg x 1024 thr, 4 loads", "512 wg x 512 thr, 4 loads", "256 wg x 512 thr, 8 loads", "1024 wg x 256 thr, 8 loads", "2048 wg x 256 thr, 4 loads"};
        printf("plain stream of %-8s %5.1f MB  %-36s %6.2f us / launch  %5.2f TB/s\n", OPS[o].name, OPS[o].bytes * 1e-6, GN[geom], us, OPS[o].bytes / us * 1e-6);
        fflush(stdout);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    };
    if (argc > 2) {
        for (int round = 0; round < 2; ++round)
            for (int o = 0; o < 5; ++o)
                for (int geom = 0; geom < 6; ++geom) run_op(o, geom);
        return 0;
    }
    for (int round = 0; round < 2; ++round) {  // interleaved rounds: position effects show as differences between the two
        run("chain, distinct buffers (HBM)", false, 0, 0, 0);
        run("chain, one layer's buffers (Infinity-Cache resident)", true, 0, 0, 0);
        run("chain + touch next launch (D 1), 64 wg, whole buffer", false, 1, 64, 256);
        run("chain + touch next launch (D 1), 256 wg, whole buffer", false, 1, 256, 256);
        run("chain + touch D 2, 64 wg, whole buffer", false, 2, 64, 256);
        run("chain + touch D 2, 256 wg, whole buffer", false, 2, 256, 256);
        run("chain + touch D 5 (a layer ahead), 64 wg, whole buffer", false, 5, 64, 256);
        run("chain + touch D 5 (a layer ahead), 256 wg, whole buffer", false, 5, 256, 256);
        run("chain + touch D 1, 64 wg, first quarter of the buffer", false, 1, 64, 64);
        run("chain + touch D 2, 256 wg, first quarter of the buffer", false, 2, 256, 64);
        run("chain + touch D 5, 256 wg, first half of the buffer", false, 5, 256, 128);
        run("chain + XCD-matched touch D 1, whole slice", false, 1, -1, 256);
        run("chain + XCD-matched touch D 1, first quarter of each slice", false, 1, -1, 64);
        run("chain + XCD-matched touch D 2, first quarter of each slice", false, 2, -1, 64);
    }
    return 0;
}
