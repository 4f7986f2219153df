// EXPERIMENTAL (opt-in via CBX_SPLIT_AK=1; written after the round-1 GPU budget was spent, so not yet run on hardware):
// A-stationary split-bf16 ("bf16x3") Linear for short contractions, K == 256 -- the q/k/v and first feed-forward projections of
// the 56 CFM transformer blocks (N = 1536 / 1024), where gemm_split_kernel spends its time in prologue / epilogue and operand
// staging (129 TF fp32-equivalent vs 175-185 TF on the K >= 768 shapes).
//
// A workgroup (8 waves, 4 along M x 2 along N) owns 128 rows and a contiguous range of 64-column tiles.  Each wave converts ITS
// 32 x 256 slice of A to bf16 planes once and keeps it in registers as MFMA A-fragments (16 k-chunks x 2 planes x 4 VGPRs = 128
// VGPRs); the loop over column tiles then stages only W (64 rows x 256, split on the fly) through LDS, so per column tile a wave
// issues 32 ds_read_b128 for 48 MFMAs and nothing else competes for LDS.  Same epilogue contract as gemm_split_kernel.
// Known issue to resolve on hardware: at 2 waves per SIMD the 128-VGPR A panel + 32-VGPR W prefetch + the activation epilogue
// exceed 256 registers (hipcc spills 56 dwords around the epilogue); candidates: 4-wave workgroups, or W tiles via global_load_lds.
#include <stdlib.h>
#include "cbx_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

constexpr int AK_K = 256;            // contraction length served
constexpr int AK_NKC = AK_K / 16;    // MFMA k-chunks
constexpr int AK_BM = 128, AK_BN = 64;
constexpr int AK_LD = AK_K + 8;      // LDS row stride in bf16 (528 B = 33 x 16 B: odd -> conflict-free ds_read_b128)
constexpr int AK_PLANE = AK_BN * AK_LD;

__device__ __forceinline__ void ak_split8(const f32x8 v, bf16x8& hi, bf16x8& lo) {
    hi = __builtin_convertvector(v, bf16x8);
    const f32x8 r = v - __builtin_convertvector(hi, f32x8);
    lo = __builtin_convertvector(r, bf16x8);
}

__global__ __launch_bounds__(512) void gemm_split_ak_kernel(const cbx_gemm_t p, int tiles_per_wg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ak_smem[];
    __bf16* Ws = reinterpret_cast<__bf16*>(ak_smem);  // planes hi | lo of one 64 x 256 W tile (67.6 KB: dynamic, > 64 KiB)

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int z = blockIdx.z, z1 = z / p.nz2, z2 = z - z1 * p.nz2;
    const int m0 = blockIdx.y * AK_BM;
    const int n_tiles = (p.N + AK_BN - 1) / AK_BN;
    const int t_beg = blockIdx.x * tiles_per_wg, t_end = min(n_tiles, t_beg + tiles_per_wg);

    const float* __restrict__ Ab = p.A + (long)z1 * p.a_s1 + (long)z2 * p.a_s2;
    const float* __restrict__ Wb = p.W + (long)z1 * p.w_s1 + (long)z2 * p.w_s2;

    // ---- A panel of this wave: rows m0 + 32*wm + lr, fragment kc covers k = 16kc + 8lh .. +8 (unconditional, clamped row)
    bf16x8 ah[AK_NKC], al[AK_NKC];
    {
        const int m = m0 + wm * 32 + lr;
        const bool ok = m < p.M;
        const float* ap = Ab + (long)(ok ? m : 0) * p.lda + 8 * lh;
#pragma unroll
        for (int k0 = 0; k0 < AK_NKC; k0 += 4) {  // 4 k-chunks (8 float4) in flight at a time: keeps the transient registers small
            f32x4 raw[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                raw[j][0] = *reinterpret_cast<const f32x4*>(ap + 16 * (k0 + j));
                raw[j][1] = *reinterpret_cast<const f32x4*>(ap + 16 * (k0 + j) + 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x8 v = {raw[j][0][0], raw[j][0][1], raw[j][0][2], raw[j][0][3], raw[j][1][0], raw[j][1][1], raw[j][1][2], raw[j][1][3]};
                if (!ok) v = f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                ak_split8(v, ah[k0 + j], al[k0 + j]);
            }
        }
    }

    // ---- W staging map: thread -> (row = tid >> 3, 8 float4 columns (tid & 7) + 8*i): 128-B coalesced segments per row
    const int w_row = tid >> 3, w_c = tid & 7;
    f32x4 wreg[8];
    bool wrow_ok = false;
    auto fetch_w = [&](int tile) {
        const int n = tile * AK_BN + w_row;
        wrow_ok = tile < t_end && n < p.N;
        const float* wp = Wb + (long)(wrow_ok ? n : 0) * p.ldw;
#pragma unroll
        for (int i = 0; i < 8; ++i) wreg[i] = *reinterpret_cast<const f32x4*>(wp + (w_c + 8 * i) * 4);
    };
    auto stage_w = [&]() {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) {  // 4 consecutive k per float4 -> one 8-B store per plane
            const f32x4 x = wrow_ok ? wreg[i] : zero;
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            const bf16x4 h = __builtin_convertvector(x, bf16x4);
            const f32x4 r = x - __builtin_convertvector(h, f32x4);
            const bf16x4 l = __builtin_convertvector(r, bf16x4);
            __bf16* dst = &Ws[w_row * AK_LD + (w_c + 8 * i) * 4];
            *reinterpret_cast<bf16x4*>(dst) = h;
            *reinterpret_cast<bf16x4*>(dst + AK_PLANE) = l;
        }
    };

    float* Cb = p.C + (long)z1 * p.c_s1 + (long)z2 * p.c_s2;
    const float* Rb = p.R ? p.R + (long)z1 * p.r_s1 + (long)z2 * p.r_s2 : nullptr;
    float* C2b = p.C2 ? p.C2 + (long)z1 * p.c2_s1 + (long)z2 * p.c2_s2 : nullptr;

    if (t_beg < t_end) fetch_w(t_beg);
    for (int tile = t_beg; tile < t_end; ++tile) {
        stage_w();
        __syncthreads();
        fetch_w(tile + 1);  // unconditional (rows past the range re-read row 0 and are never staged as data)

        // ---- 32 x 32 per wave: columns n0 + 32*wn + (lane & 31), 16 k-chunks x 3 plane products
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const __bf16* bs = &Ws[(wn * 32 + lr) * AK_LD + 8 * lh];
#pragma unroll
        for (int kc = 0; kc < AK_NKC; ++kc) {
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bs + 16 * kc);
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(bs + AK_PLANE + 16 * kc);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kc], bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[kc], bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kc], bh, acc, 0, 0, 0);
        }

        // ---- epilogue of this column tile.  C/D map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        const int n = tile * AK_BN + wn * 32 + lr;
        if (n < p.N) {
            const float bia = p.bias ? p.bias[n] : 0.f;
            const float a1 = p.act1_param ? p.act1_param[n] : 0.f;
            const float a2 = p.act2_param ? p.act2_param[n] : 0.f;
            const int mb = m0 + wm * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                if (m >= p.M) continue;
                float v = acc[r] + bia;
                v = cbx_act(v, p.act1, p.act1_slope, a1);
                if (Rb) v += Rb[(long)m * p.ldr + n];
                v *= p.alpha;
                float* dst = Cb + (long)m * p.ldc + n;
                if (p.beta != 0.f) v += p.beta * *dst;
                *dst = v;
                if (C2b) C2b[(long)m * p.ldc2 + n] = cbx_act(v, p.act2, p.act2_slope, a2);
            }
        }
        __syncthreads();  // every wave is done with this W tile before the next one is staged
    }
}

}  // namespace

// Called from cbx_gemm_split_dispatch.  Returns -1 when the shape is not served (caller continues with gemm_split_kernel).
int cbx_gemm_split_ak_dispatch(const cbx_gemm_t& p, hipStream_t st) {
    static const int enabled = getenv("CBX_SPLIT_AK") ? atoi(getenv("CBX_SPLIT_AK")) : 0;
    if (!enabled) return -1;
    if (p.K != AK_K || p.taps != 1 || p.up != 1 || p.stride != 1 || p.pad_left != 0 || p.lens || p.w_kn || p.swiglu) return -1;
    if (p.N < 512 || p.M < 1024) return -1;  // needs enough column tiles per workgroup to amortise the A panel
    const int n_tiles = (p.N + AK_BN - 1) / AK_BN;
    const int m_tiles = (p.M + AK_BM - 1) / AK_BM;
    // column-tile groups: fill the 256 CUs about once (1 workgroup of 8 waves per CU), at least 4 tiles per workgroup
    int groups = 256 / (m_tiles * p.nz1 * p.nz2);
    groups = max(1, min(groups, n_tiles / 4));
    const int per = (n_tiles + groups - 1) / groups;
    dim3 grid((n_tiles + per - 1) / per, m_tiles, p.nz1 * p.nz2);
    constexpr size_t lds = (size_t)2 * AK_PLANE * sizeof(__bf16);
    static bool configured = false;  // > 64 KiB of dynamic LDS has to be opted into once
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_ak_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return cbx_set_error((int)e, "gemm_split_ak: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
        configured = true;
    }
    hipLaunchKernelGGL(gemm_split_ak_kernel, grid, dim3(512), lds, st, p, per);
    return cbx_check_launch("gemm_split_ak");
}
