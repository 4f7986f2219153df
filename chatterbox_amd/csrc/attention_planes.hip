// Flash attention (head_dim 64) of the CFM transformer blocks on PLANE-FORMAT operands (round 3): the f16x3 arithmetic of
// attention_split.hip without any operand conversion in the KV loop.
//
//   Q, K : plane-format tensors [token][d] as the q/k projection (cbx_gemm_planes, P output) writes them;
//   V^T  : plane-format tensor [d][token] -- the v projection computed with swapped operands (A = W_v, "W" = the activations), so that
//          the transposed operand the second product needs is what the GEMM stores; row tails beyond Tk must hold finite values.
//   O    : plane-format [token][d] (feeds the to_out projection only).
//
//   S^T = K Q^T : A = K tile  from LDS [key slot][d]   (lane: slot = lane&31, d = 16kc + 8*(lane>>5) .. +8), B = Q^T in registers
//   O^T = V^T P^T: A = V^T tile from LDS [d][key]      (lane: d = lane&31, keys 16c + 8*(lane>>5) .. +8),    B = P^T from the S accumulators
//   Register r of S sub-tile t in half-wave lh is A-row (r&3) + 8(r>>2) + 4 lh.  The second product wants the lane's registers 8u..8u+7 to
//   be the 8 CONSECUTIVE keys 16(2t+u) + 8 lh .. +8 of the V^T row; so the K tile is loaded with its rows permuted (row slot i holds key
//   i with bits 2 and 3 swapped) -- a permutation of the DMA source address, free -- and V^T stays in natural key order.
//
//   Both operand tiles go global -> LDS by global_load_lds (16 B per lane, no VGPR, no VALU, no ds_write), two stages, ONE barrier per
//   KV tile; the LDS image is lane-linear with the bank-conflict XOR swizzle applied to the source address and the ds_read_b128 address
//   (chunk c of row r at slot r*8 + (c ^ ((r/2) & 7)), gemm_planes.hip).  Keys >= key_lens[z] are fetched from a zero page / masked in
//   the softmax.  The softmax scale is folded into the exponent (p = exp2(fma(s, scale*log2 e, -m*scale*log2 e))): Q is used as stored.
//
// Replaces diffusers Attention (scaled_dot_product_attention) inside BasicTransformerBlock (reference matcha/transformer.py:243-316).
#include <stdlib.h>
#include "cbx_common.h"

namespace {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __attribute__((aligned(16))) const unsigned cbx_attn_zero_page[4] = {0u, 0u, 0u, 0u};
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct FlashPlArgs {
    const _Float16* q; const _Float16* k; const _Float16* vt; _Float16* o; const int* key_lens;
    int Tq, Tk;
    long q_sb, q_st, q_lo, k_sb, k_st, k_lo, vt_sb, vt_sd, vt_lo, o_sb, o_st, o_lo;  // halves
    float scale;
    int causal;
    int diag;  // -DCBX_DIAG builds only (scripts/diag_planes.sh): 1 no DMA in the loop, 2 no S MFMAs, 4 no softmax, 8 no PV MFMAs, 16 no P split
};

constexpr int PKT = 64;                  // keys per tile
constexpr int PL_TILE = 64 * 8 * 16;     // one plane of one operand tile: 64 rows x 8 chunks x 16 B
constexpr int PL_STAGE = 4 * PL_TILE;    // K h, K l, V^T h, V^T l

__device__ __forceinline__ void mma3(const f16x8 ah, const f16x8 al, const f16x8 bh, const f16x8 bl, f32x16& acc, f32x16& accc) {
    accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accc, 0, 0, 0);
    accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void flash_attn_pl_kernel(const FlashPlArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PL_STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int tile = cbx_xcd_remap((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gridDim.x * gridDim.y * gridDim.z);
    const int qt = tile % gridDim.x, head = (tile / gridDim.x) % gridDim.y, z = tile / (gridDim.x * gridDim.y);
    const int q0 = qt * 128;
    const int qi = q0 + wid * 32 + lr;  // this lane's query
    const _Float16* qb = a.q + (long)z * a.q_sb + head * 64;
    const _Float16* kb = a.k + (long)z * a.k_sb + head * 64;
    const _Float16* vb = a.vt + (long)z * a.vt_sb + (long)head * 64 * a.vt_sd;
    const int klen = a.key_lens ? min(a.Tk, a.key_lens[z]) : a.Tk;
    const int coff = a.Tk - a.Tq;
    int kend = klen;
    if (a.causal) kend = min(kend, q0 + 128 + coff);

    // ---- DMA descriptors.  Wave-level load L = wid*8 + i fills slots [64 L, 64 L + 64) of a stage: L / 8 = {K h, K l, V^T h, V^T l}
    //      (= the wave), 8 rows of 8 chunks per load.
    const _Float16* ptr[8];
    int kv_key[8];  // K: key offset of the lane's row slot inside a tile; V^T: first key of the lane's chunk
    const _Float16* zp = reinterpret_cast<const _Float16*>(cbx_attn_zero_page);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 8 * i + (lane >> 3), pc = lane & 7;
        const int c = pc ^ ((row >> 1) & 7);
        if (wid < 2) {  // K plane wid: row slot `row` holds key row with bits 2 and 3 swapped
            const int key = (row & ~12) | ((row & 4) << 1) | ((row & 8) >> 1);
            kv_key[i] = key;
            ptr[i] = kb + (long)key * a.k_st + (wid ? a.k_lo : 0) + c * 8;
        } else {        // V^T plane wid - 2: row = d, chunk = 8 consecutive keys
            kv_key[i] = c * 8;
            ptr[i] = vb + (long)row * a.vt_sd + (wid == 3 ? a.vt_lo : 0) + c * 8;
        }
    }
    const long step = wid < 2 ? (long)PKT * a.k_st : (long)PKT;
    int ld_j0 = 0;
    auto issue = [&](int stage) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const _Float16* src = ld_j0 + kv_key[i] < klen ? ptr[i] : zp;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + stage * PL_STAGE + (wid * 8 + i) * 1024), 16, 0, 0);
            ptr[i] += step;
        }
        ld_j0 += PKT;
    };
    if (kend > 0) issue(0);

    // ---- Q planes straight from memory: chunk kc covers d = 16kc + 8lh .. +8
    f16x8 qh[4], ql[4];
    {
        const bool ok = qi < a.Tq;
        const _Float16* qp = qb + (long)(ok ? qi : 0) * a.q_st + 8 * lh;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            qh[kc] = *reinterpret_cast<const f16x8*>(qp + 16 * kc);
            ql[kc] = *reinterpret_cast<const f16x8*>(qp + a.q_lo + 16 * kc);
        }
    }
    const float sc = a.scale * 1.4426950408889634f;  // scores enter the exponent in units of log2

    f32x16 ot[2], otc[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[d][r] = otc[d][r] = 0.f;
    float m_run = -INFINITY, mc_run = -INFINITY, l_run = 0.f;  // running maximum (raw score), the same times sc, running sum

    const int q_lo = q0 + wid * 32 + coff;  // causal: the wave's first query sees keys <= q_lo
    const int jmax = a.causal ? min(klen - 1, qi + coff) : klen - 1;  // last key this lane's query sees
    const int swz = (lr >> 1) & 7;
    const int row_off = lr * 128;

#ifdef CBX_DIAG
    const int dg = a.diag;
#define DG(bit) (dg & (bit))
#else
#define DG(bit) 0
#endif
    for (int j0 = 0, t = 0; j0 < kend; j0 += PKT, ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (j0 + PKT < kend && !DG(1)) issue((t + 1) & 1);
        const unsigned char* st_k = smem + (t & 1) * PL_STAGE;
        const unsigned char* st_v = st_k + 2 * PL_TILE;

        // ---- S^T = K Q^T  (2 sub-tiles of 32 key slots, 4 d-chunks of 16)
        f32x16 st[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x16 stc;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[s][r] = stc[r] = 0.f;
            if (DG(2)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) st[s][r] = (float)(r + lane) * 0.01f;
            } else
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const int off = s * 32 * 128 + row_off + (((2 * kc + lh) ^ swz) << 4);
                const f16x8 kh = *reinterpret_cast<const f16x8*>(st_k + off);
                const f16x8 kl = *reinterpret_cast<const f16x8*>(st_k + PL_TILE + off);
                mma3(kh, kl, qh[kc], ql[kc], st[s], stc);
            }
            st[s] += stc * (1.0f / CBX_F16_LO_SCALE);
        }

        // ---- online softmax: register r of sub-tile s is key j0 + 32 s + 16 (r >> 3) + 8 lh + (r & 7)
        const bool full = j0 + PKT <= klen && (!a.causal || j0 + PKT - 1 <= q_lo);  // wave-uniform: nothing to mask in this tile
        float mt = -INFINITY;
        if (!DG(4)) {
        if (full) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int r = 0; r < 16; r += 2) mt = fmaxf(fmaxf(mt, st[s][r]), st[s][r + 1]);
        } else {
            const int jl = jmax - j0 - 8 * lh;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sv = s * 32 + 16 * (r >> 3) + (r & 7) <= jl ? st[s][r] : -INFINITY;
                    st[s][r] = sv;
                    mt = fmaxf(mt, sv);
                }
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float mc_new = m_new > -INFINITY ? m_new * sc : 0.f;  // a row that has seen no key yet: exp2(-inf - 0) = 0
        const float alpha = __builtin_amdgcn_exp2f(mc_run - mc_new);
        float ls = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(st[s][r], sc, -mc_new));
                st[s][r] = pv;
                ls += pv;
            }
        l_run = l_run * alpha + ls;
        m_run = m_new;
        mc_run = m_new > -INFINITY ? mc_new : -INFINITY;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                ot[d] *= alpha;
                otc[d] *= alpha;
            }
        }
        }  // !DG(4)

        // ---- O^T += V^T P^T : chunk c = 2s + u contracts the 16 keys held in registers 8u..8u+7 of both half-waves
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                u32x4 ph, pl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned h2, l2;
                    if (DG(16)) {
                        h2 = __builtin_bit_cast(unsigned, st[s][8 * u + 2 * e]);
                        l2 = __builtin_bit_cast(unsigned, st[s][8 * u + 2 * e + 1]);
                    } else
                    cbx_split2(st[s][8 * u + 2 * e], st[s][8 * u + 2 * e + 1], h2, l2);
                    ph[e] = h2;
                    pl[e] = l2;
                }
                if (DG(8)) {
                    asm volatile("" ::"v"(ph), "v"(pl));
                    continue;
                }
                const f16x8 pfh = __builtin_bit_cast(f16x8, ph), pfl = __builtin_bit_cast(f16x8, pl);
                const int c = 2 * s + u;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int off = dt * 32 * 128 + row_off + (((2 * c + lh) ^ swz) << 4);
                    const f16x8 vh = *reinterpret_cast<const f16x8*>(st_v + off);
                    const f16x8 vl = *reinterpret_cast<const f16x8*>(st_v + PL_TILE + off);
                    mma3(vh, vl, pfh, pfl, ot[dt], otc[dt]);
                }
            }
    }

#pragma unroll
    for (int d = 0; d < 2; ++d) ot[d] += otc[d] * (1.0f / CBX_F16_LO_SCALE);
    // ---- finalise: both half-waves hold partial sums of the same query.  O^T register r of d-tile dt is d = 32 dt + (r&3) + 8(r>>2) + 4 lh
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (qi < a.Tq) {
        _Float16* op = a.o + (long)z * a.o_sb + (long)qi * a.o_st + head * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned h0, l0, h1, l1;
                cbx_split2(ot[d][g * 4 + 0] * inv, ot[d][g * 4 + 1] * inv, h0, l0);
                cbx_split2(ot[d][g * 4 + 2] * inv, ot[d][g * 4 + 3] * inv, h1, l1);
                *reinterpret_cast<uint2*>(op + d * 32 + 8 * g + 4 * lh) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(op + a.o_lo + d * 32 + 8 * g + 4 * lh) = make_uint2(l0, l1);
            }
    }
}

}  // namespace

extern "C" int cbx_flash_attn_planes(const void* q, const void* k, const void* vt, void* o, const int* key_lens, int nz1, int n_heads, int Tq,
                                     int Tk, long q_sb, long q_st, long q_lo, long k_sb, long k_st, long k_lo, long vt_sb, long vt_sd,
                                     long vt_lo, long o_sb, long o_st, long o_lo, float scale, int causal, void* stream) {
    CBX_REQUIRE(q && k && vt && o, "flash_attn_planes: null operand");
    CBX_REQUIRE(Tq > 0 && Tk > 0 && nz1 > 0 && n_heads > 0, "flash_attn_planes: bad shape");
    CBX_REQUIRE((q_sb | q_st | q_lo | k_sb | k_st | k_lo | vt_sb | vt_sd | vt_lo) % 8 == 0 &&
                    (((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt) & 15) == 0,
                "flash_attn_planes: q / k / v^T planes must be 16-byte aligned (strides and plane offsets multiples of 8 halves)");
    CBX_REQUIRE((o_sb | o_st | o_lo) % 4 == 0 && ((uintptr_t)o & 7) == 0, "flash_attn_planes: output planes must be 8-byte aligned");
    CBX_REQUIRE(vt_sd >= (Tk + 7) / 8 * 8, "flash_attn_planes: a V^T row must hold Tk rounded up to 8 keys (got stride %ld for Tk = %d)", vt_sd, Tk);
    FlashPlArgs a{reinterpret_cast<const _Float16*>(q), reinterpret_cast<const _Float16*>(k), reinterpret_cast<const _Float16*>(vt),
                  reinterpret_cast<_Float16*>(o), key_lens, Tq, Tk, q_sb, q_st, q_lo, k_sb, k_st, k_lo, vt_sb, vt_sd, vt_lo, o_sb, o_st, o_lo,
                  scale, causal, 0};
#ifdef CBX_DIAG
    a.diag = getenv("CBX_ATTN_DIAG") ? atoi(getenv("CBX_ATTN_DIAG")) : 0;
#endif
    dim3 grid((Tq + 127) / 128, n_heads, nz1);
    hipLaunchKernelGGL(flash_attn_pl_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    return cbx_check_launch("flash_attn_planes");
}
