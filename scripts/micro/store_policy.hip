// Stand-alone HIP program (hipcc --offload-arch=gfx950 -O2 store_policy.hip -o store_policy): what do the bytes a kernel WRITES cost the next launch of a dependent chain,
// and does the store's cache policy change it?  (Round 6, call az: write-through stores of the decode kernels' outputs made the T3 token step 2.1 % shorter.)
//
// chain: 150 dependent launches in one hipGraph, 256 workgroups x 512 threads.  Every launch streams 8 MB of its own weights (non-temporal loads: ~4 us, like a small decode
// GEMV), then its workgroups write OUT bytes in total (each its 1 / 256 share, 16 bytes per lane per store) with the policy under test: plain, sc1 (agent-scope write-through),
// sc0 sc1, nt.  READ = 1: every workgroup of the NEXT launch first reads the whole OUT bytes its predecessor wrote (the all-to-all hand-off of the decode step: each workgroup
// reads the full activation image).  Prints us per launch for every (OUT, policy, READ).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int POLICY>
__device__ __forceinline__ void store16(float* p, f32x4 v) {
    if constexpr (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 3) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
    else *reinterpret_cast<f32x4*>(p) = v;
}

template <int POLICY, bool READ>
__global__ __launch_bounds__(512) void step_kernel(const f32x4* __restrict__ w, long n4_per_wg, const float* __restrict__ in, float* __restrict__ out, long out4_total) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (READ) {  // the predecessor's whole output, by every workgroup
        const f32x4* i4 = reinterpret_cast<const f32x4*>(in);
        for (long i = threadIdx.x; i < out4_total; i += 512) acc += i4[i];
    } else {
        acc.x = in[threadIdx.x & 63];
    }
    const f32x4* p = w + (long)blockIdx.x * n4_per_wg;
    f32x4 a2[4] = {};
    long i = threadIdx.x;
    for (; i + 3 * 512 < n4_per_wg; i += 4 * 512) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(p + i + u * 512);
#pragma unroll
        for (int u = 0; u < 4; ++u) a2[u] += v[u];
    }
    for (; i < n4_per_wg; i += 512) a2[0] += __builtin_nontemporal_load(p + i);
    acc = (acc + a2[0] + a2[1] + a2[2] + a2[3]) * 1e-9f;
    const long per_wg = out4_total / gridDim.x;  // 16-byte pieces per workgroup (>= 1)
    float* o = out + (long)blockIdx.x * per_wg * 4;
    for (long j = threadIdx.x; j < per_wg; j += 512) store16<POLICY>(o + j * 4, acc);
}

// The hand-off as a tuned kernel would do it: the whole 64 KB image (8 loads of 16 bytes per thread) is REQUESTED before the weight stream and consumed after it.
template <int POLICY, int NX>
__global__ __launch_bounds__(512) void hidden_kernel(const f32x4* __restrict__ w, long n4_per_wg, const float* __restrict__ in, float* __restrict__ out, long out4_total) {
    const f32x4* i4 = reinterpret_cast<const f32x4*>(in);
    f32x4 xv[NX > 0 ? NX : 1];
#pragma unroll
    for (int u = 0; u < NX; ++u) xv[u] = i4[threadIdx.x + u * 512];
    const f32x4* p = w + (long)blockIdx.x * n4_per_wg;
    f32x4 a2[4] = {};
    long i = threadIdx.x;
    for (; i + 3 * 512 < n4_per_wg; i += 4 * 512) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(p + i + u * 512);
#pragma unroll
        for (int u = 0; u < 4; ++u) a2[u] += v[u];
    }
    for (; i < n4_per_wg; i += 512) a2[0] += __builtin_nontemporal_load(p + i);
    f32x4 acc = a2[0] + a2[1] + a2[2] + a2[3];
#pragma unroll
    for (int u = 0; u < NX; ++u) acc += xv[u];
    if (NX == 0) acc.x += in[threadIdx.x & 63];
    acc *= 1e-9f;
    const long per_wg = out4_total / gridDim.x;
    float* o = out + (long)blockIdx.x * per_wg * 4;
    for (long j = threadIdx.x; j < per_wg; j += 512) store16<POLICY>(o + j * 4, acc);
}

typedef void (*kern_t)(const f32x4*, long, const float*, float*, long);
static const kern_t K[4][2] = {{step_kernel<0, false>, step_kernel<0, true>}, {step_kernel<1, false>, step_kernel<1, true>},
                               {step_kernel<2, false>, step_kernel<2, true>}, {step_kernel<3, false>, step_kernel<3, true>}};
static const char* PN[4] = {"plain", "sc1", "sc0 sc1", "nt"};

int main(int argc, char** argv) {
    const int N = 150, REPLAYS = argc > 1 ? atoi(argv[1]) : 20;
    const long WBYTES = 8 << 20, OUTMAX = 4 << 20, WMAX = argc > 2 ? (32 << 20) : WBYTES;
    hipStream_t s0;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    std::vector<float*> w(N);
    for (int j = 0; j < N; ++j) { CK(hipMalloc(&w[j], WMAX)); CK(hipMemsetAsync(w[j], 0, WMAX, s0)); }
    float* act[2];
    for (int k = 0; k < 2; ++k) { CK(hipMalloc(&act[k], OUTMAX)); CK(hipMemsetAsync(act[k], 0, OUTMAX, s0)); }
    CK(hipStreamSynchronize(s0));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    if (argc > 2) {  // hand-off mode: a 64 KB image written by every launch (sc1) and read WHOLE by every workgroup of the next one, requested before the weight stream
        const long wb[] = {4 << 20, 12 << 20, 16 << 20, 32 << 20};  // the decode projections: o, q | k | v, down, gate|up
        for (int round = 0; round < 2; ++round)
            for (long wbytes : wb) {
                double us[3];
                for (int m = 0; m < 3; ++m) {  // 0: no hand-off (256 B read), 1: 64 KB requested up front, 2: 64 KB read in a serial loop first
                    kern_t k = m == 0 ? (kern_t)hidden_kernel<1, 0> : m == 1 ? (kern_t)hidden_kernel<1, 8> : (kern_t)step_kernel<1, true>;
                    hipGraph_t g; hipGraphExec_t ge;
                    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
                    for (int j = 0; j < N; ++j)
                        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, s0, reinterpret_cast<const f32x4*>(w[j]), wbytes / 16 / 256, act[j & 1], act[(j + 1) & 1], (long)(64 << 10) / 16);
                    CK(hipStreamEndCapture(s0, &g));
                    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s0));
                    CK(hipStreamSynchronize(s0));
                    CK(hipEventRecord(t0, s0));
                    for (int r = 0; r < REPLAYS; ++r) CK(hipGraphLaunch(ge, s0));
                    CK(hipEventRecord(t1, s0));
                    CK(hipStreamSynchronize(s0));
                    float ms; CK(hipEventElapsedTime(&ms, t0, t1));
                    us[m] = ms * 1e3 / REPLAYS / N;
                    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
                }
                printf("weights %2ld MB per launch, 64 KB image out (sc1): no hand-off %.2f us | image requested before the stream %.2f us | image read first, serially %.2f us\n",
                       wbytes >> 20, us[0], us[1], us[2]);
                fflush(stdout);
            }
        return 0;
    }
    const long outs[] = {4 << 10, 64 << 10, 256 << 10, 512 << 10, 2 << 20};
    for (int round = 0; round < 2; ++round)
        for (int rd = 0; rd < 2; ++rd)
            for (long ob : outs) {
                if (rd && ob > (512 << 10)) continue;  // 256 workgroups x 2 MB of reads is not a hand-off any more
                double us[4];
                for (int pol = 0; pol < 4; ++pol) {
                    hipGraph_t g; hipGraphExec_t ge;
                    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
                    for (int j = 0; j < N; ++j)
                        hipLaunchKernelGGL(K[pol][rd], dim3(256), dim3(512), 0, s0, reinterpret_cast<const f32x4*>(w[j]), WBYTES / 16 / 256, act[j & 1], act[(j + 1) & 1], ob / 16);
                    CK(hipStreamEndCapture(s0, &g));
                    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s0));
                    CK(hipStreamSynchronize(s0));
                    CK(hipEventRecord(t0, s0));
                    for (int r = 0; r < REPLAYS; ++r) CK(hipGraphLaunch(ge, s0));
                    CK(hipEventRecord(t1, s0));
                    CK(hipStreamSynchronize(s0));
                    float ms; CK(hipEventElapsedTime(&ms, t0, t1));
                    us[pol] = ms * 1e3 / REPLAYS / N;
                    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
                }
                printf("out %5ld KB per launch, next launch %s: ", ob >> 10, rd ? "reads ALL of it in every workgroup" : "reads 256 B of it              ");
                for (int pol = 0; pol < 4; ++pol) printf(" %s %.2f us", PN[pol], us[pol]);
                printf("\n");
                fflush(stdout);
            }
    return 0;
}
