// Stand-alone HIP program (hipcc --offload-arch=gfx950 -O2 concur.hip -o concur): can a chain of small dependent kernels on one HIP stream keep its pace
// beside a stream of chip-filling kernels -- and what decides it?  (the throughput schedule "T3 of batch k + 1 beside flow + vocoder
// of batch k" overlaps only 7 %.)
//
// foreground ("T3 decode"): a hipGraph of CHAIN dependent launches, each 256 workgroups of FT threads holding FV VGPRs and FL bytes of LDS, busy for
//                           ~FG_US us (spin on the 100 MHz wall clock: latency-bound work that needs no issue slots to speak of);
// background ("flow"):      BGN back-to-back launches on a second (lower-priority) stream, each BW workgroups per CU x 256 CUs of BT threads with BV VGPRs and
//                           BL bytes of LDS, busy for ~BG_US us.
// For every (foreground, background) pair: foreground alone, background alone, both at once.  If workgroups of the two streams can be CO-RESIDENT on a
// CU (resources left over by the background), the chain keeps its per-launch time; if not, every foreground launch waits for background workgroups to
// retire.  Prints one JSON line per pair.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int V, int MAXT>
__global__ __launch_bounds__(MAXT) void busy_kernel(int ticks, float* sink) {
    extern __shared__ float lds[];
    float r[V];
#pragma unroll
    for (int i = 0; i < V; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"((float)(threadIdx.x + i)));
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(r[i]));
    if (acc == 123.456f) sink[0] = acc + lds[threadIdx.x & 63];
}

typedef void (*kern_t)(int, float*);
static const kern_t ALL[] = {busy_kernel<16, 1024>, busy_kernel<48, 1024>, busy_kernel<80, 576>, busy_kernel<112, 576>, busy_kernel<152, 576>, busy_kernel<224, 512>};
static kern_t pick(int v) { return ALL[v <= 24 ? 0 : v <= 56 ? 1 : v <= 88 ? 2 : v <= 120 ? 3 : v <= 160 ? 4 : 5]; }

struct Shape { const char* name; int threads, vgpr, lds, wg_per_cu; float us; };

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int CHAIN = 100, REPLAYS = argc > 1 ? atoi(argv[1]) : 20;
    int dev = 0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    const int CUS = prop.multiProcessorCount;
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t sf, sb;
    CK(hipStreamCreateWithPriority(&sf, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, lo));
    float* sink;
    CK(hipMalloc(&sink, 4096));
    // foregrounds: the decode GEMVs are 512-thread workgroups with 68 .. 198 VGPRs and 8-16 KiB of LDS
    const Shape fgs[] = {{"fg 256thr 24v 0K", 256, 24, 0, 1, 2.5f},    {"fg 512thr 56v 0K", 512, 56, 0, 1, 2.5f},   {"fg 512thr 88v 16K", 512, 88, 16384, 1, 2.5f},
                         {"fg 512thr 120v 16K", 512, 120, 16384, 1, 2.5f}, {"fg 512thr 160v 16K", 512, 160, 16384, 1, 2.5f}, {"fg 1024thr 56v 16K", 1024, 56, 16384, 1, 2.5f}};
    // backgrounds: the plane GEMM's loader form (576 threads, 128 KiB, <= 168 VGPRs, 1 per CU), its symmetric form (512 threads, 64 KiB, 128 VGPRs, 2 per CU),
    // the plane attention (512 threads, ~100 KiB, ~200 VGPRs), a "polite" GEMM (512 threads, 96 KiB, 120 VGPRs, 1 per CU), and short-lived workgroups
    const Shape bgs[] = {{"bg loader-form 576thr 160v 128K x1 40us", 576, 160, 131072, 1, 40.f},
                         {"bg symmetric 512thr 120v 64K x2 40us", 512, 120, 65536, 2, 40.f},
                         {"bg attention 512thr 232v 96K x1 100us", 512, 232, 98304, 1, 100.f},
                         {"bg polite 512thr 120v 96K x1 40us", 512, 120, 98304, 1, 40.f},
                         {"bg polite 512thr 120v 96K x1 8us x5 waves of workgroups", 512, 120, 98304, 5, 8.f}};
    for (auto k : ALL)
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    auto launch = [&](const Shape& s, hipStream_t st, int grid) {
        hipLaunchKernelGGL(pick(s.vgpr), dim3(grid), dim3(s.threads), s.lds, st, (int)(s.us * 100.f), sink);
    };
    for (const Shape& fg : fgs) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(sf, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < CHAIN; ++i) launch(fg, sf, CUS);
        CK(hipStreamEndCapture(sf, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        auto run_fg = [&]() { for (int r = 0; r < REPLAYS; ++r) CK(hipGraphLaunch(ge, sf)); };
        run_fg();
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1, b0, b1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
        CK(hipEventRecord(e0, sf));
        run_fg();
        CK(hipEventRecord(e1, sf));
        CK(hipDeviceSynchronize());
        float fg_alone;
        CK(hipEventElapsedTime(&fg_alone, e0, e1));
        for (const Shape& bg : bgs) {
            const int grid = CUS * bg.wg_per_cu;
            // enough background launches to outlast a 6x slower foreground
            const int bgn = (int)(6.f * fg_alone * 1000.f / (bg.us * (bg.wg_per_cu > 2 ? bg.wg_per_cu : 1))) + 8;
            for (int i = 0; i < 8; ++i) launch(bg, sb, grid);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(b0, sb));
            for (int i = 0; i < bgn; ++i) launch(bg, sb, grid);
            CK(hipEventRecord(b1, sb));
            CK(hipDeviceSynchronize());
            float bg_alone;
            CK(hipEventElapsedTime(&bg_alone, b0, b1));
            const double t0 = now_ms();
            CK(hipEventRecord(b0, sb));
            for (int i = 0; i < bgn; ++i) launch(bg, sb, grid);
            CK(hipEventRecord(b1, sb));
            CK(hipEventRecord(e0, sf));
            run_fg();
            CK(hipEventRecord(e1, sf));
            CK(hipDeviceSynchronize());
            const double wall = now_ms() - t0;
            float fg_both, bg_both;
            CK(hipEventElapsedTime(&fg_both, e0, e1));
            CK(hipEventElapsedTime(&bg_both, b0, b1));
            printf("{\"fg\": \"%s\", \"bg\": \"%s\", \"fg_us_per_launch_alone\": %.2f, \"fg_us_per_launch_beside_bg\": %.2f, \"fg_slowdown\": %.2f, "
                   "\"bg_ms_alone\": %.2f, \"bg_ms_beside_fg\": %.2f, \"wall_ms_both\": %.2f, \"serial_ms\": %.2f}\n",
                   fg.name, bg.name, 1e3f * fg_alone / (CHAIN * REPLAYS), 1e3f * fg_both / (CHAIN * REPLAYS), fg_both / fg_alone, bg_alone, bg_both, wall,
                   fg_alone + bg_alone);
            fflush(stdout);
        }
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
