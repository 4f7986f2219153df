// Fused feed-forward of the CFM transformer blocks on plane-format operands (round 3):
//     x[m][:] += W2 . GELU(W1 . h[m][:] + b1) + b2          h = LayerNorm(x) in plane format (cbx_layernorm_planes_f32), D = 256, F = 1024
// in ONE launch: the F-wide intermediate never leaves the chip.  Unfused (cbx_gemm_planes twice) it is written and re-read as 2 x 65 MB
// of planes per call at the bench shape, and each of the two GEMMs pays its own prologue / epilogue; this kernel reads h (16 MB) and the
// residual, writes x.
//
// Structure = attention without a softmax (gemm_planes.hip for the arithmetic, attention_planes.hip for the register hand-over):
//   a workgroup (8 waves) owns 64 tokens and walks the hidden dimension in chunks of 128:
//   phase 1  H^T[128 hidden][64 tokens] = W1[chunk] . h^T   (K = 256 in 8 tiles of 32; operands swapped on purpose: the accumulator then has
//            lane = token, registers = hidden, i.e. after bias + GELU + split a lane holds 4 CONSECUTIVE hidden values of its token per
//            register quad -- exactly an 8-byte piece of the A operand of phase 2, written to LDS with ds_write_b64, no transpose);
//   phase 2  O[64 tokens][256] += GELU(H)[64][128 of this chunk] . W2[:, chunk]^T   (K = 128 in 4 tiles of 32), accumulators live across chunks;
//   epilogue O + b2 + residual -> x (fp32) and / or planes.
//   Two DMA rings (global -> LDS, buffer_load ... lds, the address in the vector offset so that the descriptor bounds-checks it): phase-1
//   tiles (h rows + W1 rows, 24 KB) and phase-2 tiles (W2 rows, 32 KB); each phase starts with its first two tiles prefetched during the
//   other phase, every wait is a counted vmcnt that leaves the other ring's prefetch in flight.  144 KB of LDS, one workgroup per CU.
//
// Replaces FeedForward (GELU(proj) -> Linear) + the residual add of BasicTransformerBlock (reference matcha/transformer.py:243-316, diffusers
// GELU: F.gelu(x) exact erf form).
#include <stdlib.h>
#include "cbx_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lptr_t;

struct MlpArgs {
    const _Float16* h; const _Float16* w1; const _Float16* w2; const float* b1; const float* b2;
    float* x; _Float16* outp;
    int M, F;
    long ldh, h_lo, ldw1, w1_lo, ldw2, w2_lo, ldx, ldp, p_lo;
    int write_x;
};

constexpr int MD = 256;                 // model width (K of phase 1, N of phase 2)
constexpr int MBM = 64;                 // tokens per workgroup
constexpr int MCH = 128;                // hidden units per chunk
constexpr int P1_PLANE = (MBM + MCH) * 4 * 16, P1_STAGE = 2 * P1_PLANE;   // 12 KB, 24 KB
constexpr int HC_TILE = MBM * 4 * 16, HC_BYTES = 4 * 2 * HC_TILE;         // one plane of one K tile 4 KB; 32 KB
constexpr int P2_PLANE = MD * 4 * 16, P2_STAGE = 2 * P2_PLANE;            // 16 KB, 32 KB
constexpr int OFF_HC = 2 * P1_STAGE, OFF_P2 = OFF_HC + HC_BYTES, MLP_LDS = OFF_P2 + 2 * P2_STAGE;  // 48 K, 80 K, 144 K

__device__ __forceinline__ void mlp_dma16(const __amdgpu_buffer_rsrc_t rs, unsigned char* lds, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)lds, 16, voff, 0, 0, 0);
}

__device__ __forceinline__ float gelu_erf(float v) { return cbx_gelu_erf(v); }

#define MLP_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

__global__ __launch_bounds__(512, 2) void mlp_pl_kernel(const MlpArgs a, int* range_flag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * MBM;
    const int nchunk = a.F / MCH;

    const __amdgpu_buffer_rsrc_t h_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.h), 0, (int)((long)a.M * a.ldh * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t w1_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.w1), 0, (int)((long)a.F * a.ldw1 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t w2_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.w2), 0, (int)((long)MD * a.ldw2 * 2), 0x00020000);

    // ---- DMA lane offsets.  Phase-1 tile: 1536 slots = 24 wave loads, 3 per wave; phase-2 tile: 2048 slots = 32 loads, 4 per wave.
    int vb1[3], vb2[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int s = (wid * 3 + i) * 64 + lane, q = s / 768, rs = s % 768, R = rs >> 2, pc = rs & 3, c = pc ^ ((R >> 2) & 3);
        vb1[i] = R < MBM ? (int)(((long)(m0 + R) * a.ldh + (q ? a.h_lo : 0) + c * 8) * 2)
                         : (int)(((long)(R - MBM) * a.ldw1 + (q ? a.w1_lo : 0) + c * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = (wid * 4 + i) * 64 + lane, q = s / 1024, rs = s % 1024, R = rs >> 2, pc = rs & 3, c = pc ^ ((R >> 2) & 3);
        vb2[i] = (int)(((long)R * a.ldw2 + (q ? a.w2_lo : 0) + c * 8) * 2);
    }
    const int w1_chunk = (int)(MCH * a.ldw1 * 2);
    auto issue1 = [&](int c, int kt) {  // phase-1 tile kt of chunk c -> ring stage kt & 1
        unsigned char* dst = smem + (kt & 1) * P1_STAGE + wid * 3 * 1024;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const bool isH = ((wid * 3 + i) * 64) % 768 < MBM * 4;  // wave-uniform: rows 0-63 of each plane are tokens
            if (isH) mlp_dma16(h_rs, dst + i * 1024, vb1[i] + kt * 64);
            else mlp_dma16(w1_rs, dst + i * 1024, vb1[i] + c * w1_chunk + kt * 64);
        }
    };
    auto issue2 = [&](int c, int kt) {  // phase-2 tile kt (K = hidden 128 c + 32 kt ..) -> ring stage kt & 1
        unsigned char* dst = smem + OFF_P2 + (kt & 1) * P2_STAGE + wid * 4 * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) mlp_dma16(w2_rs, dst + i * 1024, vb2[i] + (c * MCH + kt * 32) * 2);
    };

    // ---- wave roles
    const int hw = wid & 3, tw = wid >> 2;  // phase 1: hidden tile 32 hw of the chunk, token tile 32 tw; phase 2: n_out 64 hw + {0, 32}, token tile tw
    const int swz = (lr >> 2) & 3;
    const int a1_off = (MBM + 32 * hw + lr) * 64, b1_off = (32 * tw + lr) * 64;  // phase-1 A rows (W1), B rows (h)
    const int a2_off = (32 * tw + lr) * 64;                                      // phase-2 A rows (GELU(H) tile)
    f32x16 o[2], oc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[j][r] = oc[j][r] = 0.f;
    float amax = 0.f;

    issue1(0, 0);
    issue1(0, 1);
    for (int c = 0; c < nchunk; ++c) {
        // ================= phase 1: H^T tile (32 hidden x 32 tokens per wave), K = 256
        f32x16 s1, s1c;
#pragma unroll
        for (int r = 0; r < 16; ++r) s1[r] = s1c[r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            if (kt == 0 && c > 0) MLP_WAIT(3);        // tile 0 of this chunk was prefetched in phase 2 of the previous one; tile 1 may still fly
            else if (kt >= 6) MLP_WAIT(4);            // a phase-2 prefetch (4 loads) was issued after this tile
            else MLP_WAIT(0);
            __builtin_amdgcn_s_barrier();
            if (kt >= 1 && kt < 7) issue1(c, kt + 1);
            if (kt == 5) issue2(c, 0);
            if (kt == 6) issue2(c, 1);
            const unsigned char* st = smem + (kt & 1) * P1_STAGE;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const int co = ((kc * 2 + lh) ^ swz) << 4;
                const f16x8 wh = *reinterpret_cast<const f16x8*>(st + a1_off + co), wl = *reinterpret_cast<const f16x8*>(st + P1_PLANE + a1_off + co);
                const f16x8 xh = *reinterpret_cast<const f16x8*>(st + b1_off + co), xl = *reinterpret_cast<const f16x8*>(st + P1_PLANE + b1_off + co);
                s1c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, s1c, 0, 0, 0);
                s1c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, s1c, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, s1, 0, 0, 0);
            }
        }
        // ---- bias + GELU + split: register r is hidden 128 c + 32 hw + 8 (r >> 2) + 4 lh + (r & 3) of token 32 tw + lr
        {
            const float* bp = a.b1 + c * MCH + 32 * hw + 4 * lh;
            unsigned char* hc = smem + OFF_HC + hw * (2 * HC_TILE) + ((32 * tw + lr) * 4) * 16 + 8 * lh;  // K tile hw of the chunk, row = token
            const int rsw = ((32 * tw + lr) >> 2) & 3;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(bp + 8 * g);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf(__builtin_fmaf(s1c[4 * g + e], 1.0f / CBX_F16_LO_SCALE, s1[4 * g + e]) + bb[e]);
                unsigned h0, l0, h1, l1;
                cbx_split2(v[0], v[1], h0, l0);
                cbx_split2(v[2], v[3], h1, l1);
                asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(v[0]), "v"(v[1]));
                asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(v[2]), "v"(v[3]));
                const int so = ((g ^ rsw) << 4);
                *reinterpret_cast<uint2*>(hc + so) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(hc + HC_TILE + so) = make_uint2(l0, l1);
            }
        }
        // ================= phase 2: O[64 tokens][256] += GELU(H)[:, chunk] . W2[:, chunk]^T, K = 128
        const bool more = c + 1 < nchunk;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt == 0) {
                MLP_WAIT(4);                                 // W2 tile 0 landed; tile 1 (4 loads) may still fly
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's GELU(H) stores: a raw s_barrier does not wait for LDS writes
            }
            else if (kt >= 2 && more) MLP_WAIT(3);           // a phase-1 prefetch of the next chunk (3 loads) was issued after this tile
            else MLP_WAIT(0);
            __builtin_amdgcn_s_barrier();                    // kt = 0: also publishes the GELU(H) tile
            if (kt == 1 || kt == 2) issue2(c, kt + 1);
            if (more && kt == 1) issue1(c + 1, 0);
            if (more && kt == 2) issue1(c + 1, 1);
            const unsigned char* sa = smem + OFF_HC + kt * (2 * HC_TILE);
            const unsigned char* sb = smem + OFF_P2 + (kt & 1) * P2_STAGE;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const int co = ((kc * 2 + lh) ^ swz) << 4;
                const f16x8 ah = *reinterpret_cast<const f16x8*>(sa + a2_off + co), al = *reinterpret_cast<const f16x8*>(sa + HC_TILE + a2_off + co);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int bo = (64 * hw + 32 * j + lr) * 64 + co;
                    const f16x8 bh = *reinterpret_cast<const f16x8*>(sb + bo), bl = *reinterpret_cast<const f16x8*>(sb + P2_PLANE + bo);
                    oc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, oc[j], 0, 0, 0);
                    oc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, oc[j], 0, 0, 0);
                    o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, o[j], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: O + b2 + residual -> x (fp32) and / or planes.  C/D map: col = lane & 31 (n_out), row = (r&3) + 8 (r>>2) + 4 lh (token)
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(a.x, 0, (int)((((long)a.M - 1) * a.ldx + MD) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t p_rs = __builtin_amdgcn_make_buffer_rsrc(a.outp, 0, a.outp ? (int)((((long)a.M - 1) * a.ldp + a.p_lo + MD) * 2) : 0, 0x00020000);
    const int ldx4 = (int)a.ldx * 4, ldp2 = (int)a.ldp * 2, plo2 = (int)a.p_lo * 2;
    const bool odd = lr & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = 64 * hw + 32 * j + lr;
        const float bia = a.b2[n];
        const int mb = m0 + 32 * tw + 4 * lh;
        const int xo = mb * ldx4 + n * 4;
        const int po = mb * ldp2 + (n & ~1) * 2 + (odd ? ldp2 : 0);
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 8) {
            float v[8], res[8];
#pragma unroll
            for (int r = 0; r < 8; ++r)
                res[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rs, xo, (((r0 + r) & 3) + 8 * ((r0 + r) >> 2)) * ldx4, 0));
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = __builtin_fmaf(oc[j][r0 + r], 1.0f / CBX_F16_LO_SCALE, o[j][r0 + r]) + bia + res[r];
            if (a.write_x) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), x_rs, xo, (((r0 + r) & 3) + 8 * ((r0 + r) >> 2)) * ldx4, 0);
            }
            if (a.outp) {  // row-pair exchange: see gemm_planes.hip
#pragma unroll
                for (int r = 0; r < 8; r += 2) {
                    const float give = odd ? v[r] : v[r + 1];
                    const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xF, 0xF, true));
                    const float c0 = odd ? got : v[r], c1 = odd ? v[r + 1] : got;
                    unsigned h2, l2;
                    cbx_split2(c0, c1, h2, l2);
                    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(c0), "v"(c1));
                    const int so = (((r0 + r) & 3) + 8 * ((r0 + r) >> 2)) * ldp2;
                    __builtin_amdgcn_raw_buffer_store_b32(h2, p_rs, po, so, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(l2, p_rs, po + plo2, so, 0);
                }
            }
        }
    }
    if (amax > 65504.f && range_flag) atomicOr(range_flag, 1);
}

}  // namespace

extern "C" int cbx_mlp_planes(const void* h, const void* w1, const void* w2, const float* b1, const float* b2, float* x, void* out_planes,
                              int M, int D, int F, long ldh, long h_lo, long ldw1, long w1_lo, long ldw2, long w2_lo, long ldx, long ldp,
                              long p_lo, int write_x, void* stream) {
    CBX_REQUIRE(h && w1 && w2 && b1 && b2 && x, "mlp_planes: null operand");
    CBX_REQUIRE(D == MD && F > 0 && F % MCH == 0 && M > 0, "mlp_planes: D must be 256 and F a multiple of 128 (got D=%d F=%d)", D, F);
    CBX_REQUIRE((ldh | h_lo | ldw1 | w1_lo | ldw2 | w2_lo) % 8 == 0 && (((uintptr_t)h | (uintptr_t)w1 | (uintptr_t)w2) & 15) == 0,
                "mlp_planes: operand planes must be 16-byte aligned");
    CBX_REQUIRE(ldh >= h_lo + D && ldw1 >= w1_lo + D && ldw2 >= w2_lo + F && h_lo > 0 && w1_lo > 0 && w2_lo > 0, "mlp_planes: operand rows hold [h | l]");
    CBX_REQUIRE((long)(M + MBM) * ldh * 2 < 0x7fffffffL && (long)F * ldw1 * 2 < 0x7fffffffL && (long)D * ldw2 * 2 < 0x7fffffffL &&
                    ((long)M * ldx + D) * 4 < 0x7fffffffL,
                "mlp_planes: operands must span less than 2 GiB (32-bit buffer offsets)");
    CBX_REQUIRE(ldx >= D && (((uintptr_t)b1) & 15) == 0, "mlp_planes: ldx >= D, b1 16-byte aligned");
    CBX_REQUIRE(!out_planes || (ldp >= p_lo + D && p_lo > 0 && (ldp | p_lo) % 2 == 0 && ((long)M * ldp + p_lo + D) * 2 < 0x7fffffffL), "mlp_planes: plane output rows hold [h | l]");
    CBX_REQUIRE(write_x || out_planes, "mlp_planes: nothing to write");
    static unsigned long long configured = 0;  // one bit per device ordinal
    const int dev = cbx_device();
    if (!(configured >> dev & 1)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_pl_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS);
        if (e != hipSuccess) return cbx_set_error((int)e, "mlp_planes: cannot reserve %d B of LDS: %s", MLP_LDS, hipGetErrorString(e));
        configured |= 1ull << dev;
    }
    MlpArgs a{reinterpret_cast<const _Float16*>(h), reinterpret_cast<const _Float16*>(w1), reinterpret_cast<const _Float16*>(w2), b1, b2, x,
              reinterpret_cast<_Float16*>(out_planes), M, F, ldh, h_lo, ldw1, w1_lo, ldw2, w2_lo, ldx, ldp, p_lo, write_x};
    hipLaunchKernelGGL(mlp_pl_kernel, dim3((M + MBM - 1) / MBM), dim3(512), MLP_LDS, (hipStream_t)stream, a, cbx_range_flag());
    return cbx_check_launch("mlp_planes");
}
