// Stand-alone HIP program (hipcc --offload-arch=gfx950 -O2 gemv_anatomy.hip -o gemv_anatomy): which PART of a decode GEMV costs the 1.3 - 1.9 us per launch that a bare
// stream of the same bytes with the same hand-off does not (profiles/r06_be_handoff_micro.log)?  Anatomy by construction: start from the bare stream and
// add the GEMV's parts one at a time.
//
// chain: 150 dependent launches in one hipGraph, 256 workgroups x 512 threads (8 waves), distinct weights per launch (4 / 16 / 32 MB: o, down, gate|up), a 64 KB image written
// by every launch (sc1 stores) and read WHOLE by every workgroup of the next one, requested before the weight stream.
//   WAVE:   each wave streams its own contiguous K slice, 1 KB per instruction (the packed GEMV image: [K block][2][64 lanes][4 floats]), 8 instructions in flight,
//           instead of the workgroup-wide 8 KB per instruction of the bare stream;
//   MFMA:   8 v_mfma_f32_16x16x4_f32 per 32-deep K block on the loaded registers (the 16 x 16 tile of dot products), the image as the A operand.  As hipcc schedules this
//           source the MFMAs of a batch of 8 loads start when (nearly) the whole batch and the image have arrived (s_waitcnt vmcnt(1) before the first one): the matrix-pipe
//           time of the launch -- 64 MFMAs x 32 cycles x 2 waves per SIMD = 1.95 us at 32 MB, 0.97 us at 16 MB -- is then EXPOSED after the stream, which is the upper
//           bound of what the exact-fp32 MFMA form can cost a 16-row GEMV;
//   REDUCE: the waves' 16 x 16 partial tiles go through LDS (8 x 256 floats), one barrier, 256 threads add them in fixed order and store one float each.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store_wt(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

template <bool WAVE, bool MFMA, bool REDUCE>
__global__ __launch_bounds__(512) void gemv_like(const f32x4* __restrict__ w, long n4_per_wg, const float* __restrict__ in, float* __restrict__ out) {
    __shared__ float red[8 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const f32x4* i4 = reinterpret_cast<const f32x4*>(in);
    f32x4 xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xv[u] = i4[tid + u * 512];  // the 64 KB image, requested first
    __builtin_amdgcn_sched_barrier(0);
    const f32x4* p = w + (long)blockIdx.x * n4_per_wg;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    // 8 loads of 16 bytes per lane in flight either way
    const long per_wave = n4_per_wg / 8;
    const f32x4* q = WAVE ? p + wv * per_wave + lane : p + tid;
    const long step = WAVE ? 64 : 512, n = WAVE ? per_wave : n4_per_wg;
    for (long i = 0; i + 7 * step < n; i += 8 * step) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(q + i + u * step);
        __builtin_amdgcn_sched_barrier(0);  // all 8 loads are requested before the first MFMA (without it hipcc re-uses two registers quads: 2 loads in flight, see the ISA)
        if constexpr (MFMA) {
#pragma unroll
            for (int u = 0; u < 8; u += 2) {  // a 32-deep K block = two 16-byte halves per lane: 8 MFMAs
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u][e], v[u][e], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u + 1][e], v[u + 1][e], acc2, 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; u += 2) acc += v[u], acc2 += v[u + 1];
        }
    }
    for (long i = n - n % (8 * step); i < n; i += step) acc += __builtin_nontemporal_load(q + i);  // tail (the 4 MB case: two loads per lane)
    acc += acc2;
    if constexpr (!MFMA) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += xv[u];
    }
    if constexpr (REDUCE) {
        *reinterpret_cast<f32x4*>(&red[wv * 256 + lane * 4]) = acc;
        __syncthreads();
        if (tid < 256) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += red[k * 256 + tid];
            if (blockIdx.x < 64) store_wt(out + blockIdx.x * 256 + tid, s * 1e-9f);  // 64 KB in total
        }
    } else {
        const float s = (acc.x + acc.y + acc.z + acc.w) * 1e-9f;
        if (blockIdx.x < 32) store_wt(out + blockIdx.x * 512 + tid, s);  // 64 KB in total
    }
}

typedef void (*kern_t)(const f32x4*, long, const float*, float*);

int main(int argc, char** argv) {
    const int N = 150, REPLAYS = argc > 1 ? atoi(argv[1]) : 20;
    const long WMAX = 32 << 20;
    hipStream_t s0;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    std::vector<float*> w(N);
    for (int j = 0; j < N; ++j) { CK(hipMalloc(&w[j], WMAX)); CK(hipMemsetAsync(w[j], 0, WMAX, s0)); }
    float* act[2];
    for (int k = 0; k < 2; ++k) { CK(hipMalloc(&act[k], 1 << 20)); CK(hipMemsetAsync(act[k], 0, 1 << 20, s0)); }
    CK(hipStreamSynchronize(s0));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    struct V { const char* name; kern_t k; };
    const V vs[] = {{"bare stream + hand-off", gemv_like<false, false, false>}, {"+ per-wave 1 KB pattern", gemv_like<true, false, false>},
                    {"+ MFMA (workgroup-wide pattern)", gemv_like<false, true, false>}, {"+ LDS reduce (workgroup-wide pattern)", gemv_like<false, false, true>},
                    {"+ per-wave pattern + MFMA", gemv_like<true, true, false>}, {"+ per-wave pattern + MFMA + LDS reduce", gemv_like<true, true, true>}};
    const long wb[] = {4 << 20, 16 << 20, 32 << 20};
    for (int round = 0; round < 2; ++round)
        for (long wbytes : wb) {
            printf("weights %2ld MB:", wbytes >> 20);
            for (const V& v : vs) {
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
                for (int j = 0; j < N; ++j)
                    hipLaunchKernelGGL(v.k, dim3(256), dim3(512), 0, s0, reinterpret_cast<const f32x4*>(w[j]), wbytes / 16 / 256, act[j & 1], act[(j + 1) & 1]);
                CK(hipStreamEndCapture(s0, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s0));
                CK(hipStreamSynchronize(s0));
                CK(hipEventRecord(t0, s0));
                for (int r = 0; r < REPLAYS; ++r) CK(hipGraphLaunch(ge, s0));
                CK(hipEventRecord(t1, s0));
                CK(hipStreamSynchronize(s0));
                float ms; CK(hipEventElapsedTime(&ms, t0, t1));
                printf("  %s %.2f us |", v.name, ms * 1e3 / REPLAYS / N);
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            }
            printf("\n");
            fflush(stdout);
        }
    return 0;
}
