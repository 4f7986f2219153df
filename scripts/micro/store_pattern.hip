// How fast can a GEMM epilogue write C (M x N fp32, row-major)?  Three store patterns over the same 128 x 64 workgroup tiles
// (8 waves x 32 x 32), no arithmetic:  hipcc -O3 --offload-arch=gfx950 scripts/micro/store_pattern.hip -o /tmp/sp && /tmp/sp
//   0: the MFMA C layout as gemm_split_kernel stores it today -- lane = column, 16 x 4-byte stores (2 rows x 128 B per instruction)
//   1: swapped operands (C^T in registers) -- lane = row, 4 x 16-byte stores (32 rows x 32 B per instruction)
//   2: through an LDS transpose -- each instruction writes 4 rows x 256 B... here simply 16 B per lane, 8 lanes per row (32 B x 8 = 128 B per row, 8 rows)
//   3: ideal streaming: every wave instruction writes 1 KiB contiguous (what a row-major elementwise kernel does)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int P>
__global__ __launch_bounds__(512) void k(float* __restrict__ C, int M, int N) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 31, lh = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    const int n0 = blockIdx.x * 64 + wn * 32, m0 = blockIdx.y * 128 + wm * 32;
    const float v = (float)tid;
    if (P == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m < M) C[(long)m * N + n0 + lr] = v + r;
        }
    } else if (P == 1) {
        const int m = m0 + lr;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (m < M) *reinterpret_cast<f32x4*>(C + (long)m * N + n0 + 8 * g + 4 * lh) = f32x4{v, v + g, v, v};
    } else if (P == 2) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int m = m0 + g * 8 + (lane >> 3);
            if (m < M) *reinterpret_cast<f32x4*>(C + (long)m * N + n0 + (lane & 7) * 4) = f32x4{v, v + g, v, v};
        }
    } else {
        const long tile = (long)blockIdx.y * gridDim.x + blockIdx.x;
        float* base = C + tile * (128 * 64) + wid * 1024;
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(base + g * 256 + lane * 4) = f32x4{v, v + g, v, v};
    }
}
template <int P>
void run(float* C, int M, int N) {
    dim3 grid(N / 64, (M + 127) / 128);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<P>, grid, dim3(512), 0, 0, C, M, N);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k<P>, grid, dim3(512), 0, 0, C, M, N);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("pattern %d  M=%d N=%d  %.2f us  %.2f TB/s\n", P, M, N, ms / 50 * 1e3, (double)M * N * 4 / (ms / 50 * 1e-3) / 1e12);
}
int main() {
    for (int N : {1536, 1024, 256}) {
        const int M = 16000;
        float* C; hipMalloc(&C, (size_t)(M + 128) * N * 4);
        run<0>(C, M, N); run<1>(C, M, N); run<2>(C, M, N); run<3>(C, M, N);
        hipFree(C);
    }
    return 0;
}
