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
constexpr float PL_THR = 3.0f;           // version 2: slack of the running maximum (log2 units) before O is rescaled

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
        mt = fmaxf(mt, cbx_xor_lane<32>(mt));
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
    const float l_tot = l_run + cbx_xor_lane<32>(l_run);
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

// x of lane (l ^ 32): one v_permlane32_swap instead of a ds_bpermute round trip.  The two operands of the swap must be DIFFERENT registers
// (vdst lanes 32-63 <-> src lanes 0-31): hipcc folds permlane32_swap(x, x) into "both results equal", so the copy is made opaque.
__device__ __forceinline__ float half_swap(float x) {
    unsigned a = __builtin_bit_cast(unsigned, x), b = a;
    asm volatile("" : "+v"(b));
    const auto sw = __builtin_amdgcn_permlane32_swap(a, b, false, false);  // sw[0] = {x_lo, x_lo}, sw[1] = {x_hi, x_hi}
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? sw[0] : sw[1]);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Version 2 ("ping-pong"): 8 waves = two groups of 4, each SIMD hosts one wave of each group.  A group's work per KV tile is a MATRIX block
// [PV(t-1), S(t)] (48 MFMAs, LDS reads, its share of the DMA issue) and a VECTOR block [softmax(t): S -> P planes in registers, running
// maximum / sum, O rescale].  The groups run half an iteration apart (group 1 passes one extra barrier first) and every block ends at a
// workgroup barrier, so on every SIMD one wave is in its matrix block while the other is in its vector block: the matrix pipe and the
// VALU work at the same time by construction instead of by the luck of the wave scheduler (measured on version 1: S / PV MFMAs, softmax
// VALU and DMA issue were ADDITIVE, profiles/r03_planes_diag_switches.log).  The two groups own two 128-query blocks and share the K / V^T
// tiles (half the L2 -> LDS traffic per query).  Tiles sit in a 3-stage LDS ring; in its matrix block t group 0 issues K(t+2), group 1
// V^T(t+1) (one tile ahead of its use in PV), by buffer_load ... lds with the key-length bound in the K descriptor (rows >= klen return
// zeros: no per-lane address arithmetic in the loop at all).
// FREE: the free-running loop -- every wave runs S(t), softmax(t), P V(t) back to back on its own 32 queries and meets the others ONCE per
// tile (the ring hand-off: tile t + 1 landed, tile t - 1 free); no stagger, no matrix / vector phases.  The two waves of a SIMD drift
// apart by themselves, which is all the overlap the hardware gives (profiles/r03_mfma_valu_inwave_micro.log).
// One 1 KiB global -> LDS DMA.  A separate __device__ function on purpose: with the builtin called from a lambda of the kernel template hipcc (ROCm 7.2) can
// silently drop a kernel instantiation's HOST stub (undefined symbol at dlopen; gemm_planes.hip has the same note).
__device__ __forceinline__ void pl2_dma16(const __amdgpu_buffer_rsrc_t rs, unsigned char* lds, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)lds, 16, voff, 0, 0, 0);
}

// NQW = 4 (round 5, "version 5", FREE only): 4-wave workgroups of 128 queries, ONE per CU (96 KiB of LDS) -- one wave per SIMD holding ~200 VGPRs, so 3/5 of
// every SIMD's register file and 64 KiB of LDS stay free for the workgroups of ANOTHER stream (the T3 decode step of the next batch in the throughput
// schedule: profiles/r05_overlap_*).  Each wave then issues its share of BOTH operand tiles (the loads of "virtual waves" wid and wid + 4).  Same
// arithmetic per query, same key order: bit-identical to the 8-wave form.
// LATE (round 6, "version 6", FREE only): the DMAs of tile t + 2 are issued BETWEEN the softmax and the PV product of tile t instead of right behind the barrier, where
// all eight waves issue theirs at once while the matrix pipe waits (an LDS-DMA piece costs its wave 60 - 185 cycles of issue, MI355X_MICROARCH.md).  Same stages, same
// waits: the target stage held tile t - 1, which every wave left before the barrier at the top of iteration t.
template <bool PRIO, bool FREE = false, int NQW = 8, bool LATE = false>
__global__ __launch_bounds__(NQW * 64, 2) void flash_attn_pl2_kernel(const FlashPlArgs a) {
    static_assert(NQW == 8 || (NQW == 4 && FREE), "4-wave workgroups: the free-running form only");
    static_assert(!LATE || FREE, "late DMA issue: the free-running form only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];  // 3 stages x [K h, K l, V^T h, V^T l] x 8 KiB

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2;
    const int lr = lane & 31, lh = lane >> 5;
    const int tile = cbx_xcd_remap((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gridDim.x * gridDim.y * gridDim.z);
    const int qt = tile % gridDim.x, head = (tile / gridDim.x) % gridDim.y, z = tile / (gridDim.x * gridDim.y);
    const int q0 = qt * (32 * NQW);
    const int qi = q0 + wid * 32 + lr;  // this lane's query (8 waves: group g owns queries q0 + 128 g .. + 128)
    const _Float16* qb = a.q + (long)z * a.q_sb + head * 64;
    const _Float16* kb = a.k + (long)z * a.k_sb + head * 64;
    const _Float16* vb = a.vt + (long)z * a.vt_sb + (long)head * 64 * a.vt_sd;
    const int klen = a.key_lens ? min(a.Tk, a.key_lens[z]) : a.Tk;
    const int nt = (klen + PKT - 1) / PKT;

    // ---- DMA: a (K, V^T) tile pair is 32 wave-level loads of 1 KiB; (virtual) wave w issues loads 4w .. 4w+3: waves 0-3 (group 0) the K tile
    //      (h rows 0-31, h rows 32-63, l rows 0-31, l rows 32-63), waves 4-7 the V^T tile likewise.  NQW = 4: wave w is virtual waves w and w + 4.
    constexpr int NG = NQW == 8 ? 1 : 2;  // operand tiles a wave loads a share of
    const int plane = (wid >> 1) & 1, rbase = (wid & 1) * 32;
    __amdgpu_buffer_rsrc_t rs[NG];
    int voff[NG][4], tstep[NG], lds_w[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int gj = NQW == 8 ? grp : j;  // 0: K, 1: V^T
        rs[j] = gj == 0
            ? __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(kb), 0, klen > 0 ? (int)((((long)klen - 1) * a.k_st + a.k_lo + 64) * 2) : 0, 0x00020000)
            : __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(vb), 0, (int)(64 * a.vt_sd * 2), 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = rbase + 8 * i + (lane >> 3), pc = lane & 7;
            const int c = pc ^ ((row >> 1) & 7);
            if (gj == 0) {  // row slot `row` holds key row with bits 2 and 3 swapped
                const int key = (row & ~12) | ((row & 4) << 1) | ((row & 8) >> 1);
                voff[j][i] = (int)((key * a.k_st + plane * a.k_lo) * 2) + c * 16;
            } else {
                voff[j][i] = (int)((row * a.vt_sd + plane * a.vt_lo) * 2) + c * 16;
            }
        }
        tstep[j] = gj == 0 ? (int)(PKT * a.k_st * 2) : PKT * 2;  // bytes per tile along the key axis
        lds_w[j] = ((gj ? 2 : 0) + plane) * PL_TILE + rbase * 128;  // this wave's first destination inside a stage
    }
    auto issue = [&](int t) {  // K(t) (group 0) or V^T(t) (group 1) into stage t % 3.  The whole address is in the VECTOR offset: the
#pragma unroll
        for (int j = 0; j < NG; ++j) {  // descriptor's bounds check ignores a scalar offset
            unsigned char* dst = smem2 + (t % 3) * PL_STAGE + lds_w[j];
#pragma unroll
            for (int i = 0; i < 4; ++i) pl2_dma16(rs[j], dst + i * 1024, voff[j][i] + t * tstep[j]);
        }
    };
    if (nt > 0) issue(0);
    if ((FREE || grp == 0) && nt > 1) issue(1);

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
    const float sc = a.scale * 1.4426950408889634f;

    f32x16 ot[2], otc[2], st[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[d][r] = otc[d][r] = st[d][r] = 0.f;
    u32x4 ph[4], pl[4];  // P planes of the previous tile: chunk c = 2s + u holds keys 16c + 8lh .. +8 of the lane's query
#pragma unroll
    for (int c = 0; c < 4; ++c) ph[c] = pl[c] = u32x4{0u, 0u, 0u, 0u};
    float m_run = -INFINITY, mc_run = -INFINITY, l_run = 0.f;
    const int jmax = klen - 1;
    const int swz = (lr >> 1) & 7;
    const int row_off = lr * 128;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                // tile 0 (and K(1)) are in LDS for everybody
    if (!FREE && grp == 1) __builtin_amdgcn_s_barrier();  // the stagger: group 1's matrix block t runs beside group 0's vector block t
    __builtin_amdgcn_sched_barrier(0);

    for (int t = 0; t <= nt; ++t) {
        if constexpr (FREE) {
            if (t == nt) break;
            if (t > 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of tile t + 1 (issued an iteration ago)
                __builtin_amdgcn_s_barrier();                      // tile t + 1 is in LDS for everybody; everybody is done with tile t - 1
            }
            if (!LATE && t + 2 < nt) issue(t + 2);  // into the stage of tile t - 1
            __builtin_amdgcn_sched_barrier(0);
        }
        // ================= matrix block: PV(t-1), S(t)   [FREE: S(t) here, PV(t) after the softmax]
        if constexpr (!FREE) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's loads of the previous block (an iteration old): published by the barriers below
            if (grp == 0) {
                if (t + 2 < nt) issue(t + 2);
            } else {
                if (t + 1 < nt) issue(t + 1);
            }
        }
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        auto pv = [&](int tv) {
            const unsigned char* st_v = smem2 + (tv % 3) * PL_STAGE + 2 * PL_TILE;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f16x8 pfh = __builtin_bit_cast(f16x8, ph[c]), pfl = __builtin_bit_cast(f16x8, pl[c]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int off = dt * 32 * 128 + row_off + (((2 * c + lh) ^ swz) << 4);
                    const f16x8 vh = *reinterpret_cast<const f16x8*>(st_v + off);
                    const f16x8 vl = *reinterpret_cast<const f16x8*>(st_v + PL_TILE + off);
                    otc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, pfh, otc[dt], 0, 0, 0);  // V_l h'  (1/2048 accumulator)
                    ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pfl, ot[dt], 0, 0, 0);    // V_h l''
                    ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pfh, ot[dt], 0, 0, 0);    // V_h h'
                }
            }
        };
        if (!FREE && t > 0) pv(t - 1);
        if (t < nt) {
            const unsigned char* st_k = smem2 + (t % 3) * PL_STAGE;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f32x16 stc;
#pragma unroll
                for (int r = 0; r < 16; ++r) st[s][r] = stc[r] = 0.f;
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
                    const int off = s * 32 * 128 + row_off + (((2 * kc + lh) ^ swz) << 4);
                    const f16x8 kh = *reinterpret_cast<const f16x8*>(st_k + off);
                    const f16x8 kl = *reinterpret_cast<const f16x8*>(st_k + PL_TILE + off);
                    mma3(kh, kl, qh[kc], ql[kc], st[s], stc);
                }
                st[s] += stc * (1.0f / CBX_F16_LO_SCALE);
            }
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (!FREE) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);

        // ================= vector block: softmax(t) -> P planes
        if (t < nt) {
            const int j0 = t * PKT;
            float mt = -INFINITY;
            if (j0 + PKT <= klen) {  // wave-uniform: nothing to mask in this tile
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
            mt = fmaxf(mt, half_swap(mt));  // the other half-wave holds the other keys of the same query
            // Running maximum with a slack of THR (in units of log2): the shift only moves when the tile maximum exceeds it by more than THR,
            // so O is rescaled rarely; P' = 2048 p then stays below 2048 * 2^THR (fp16 range: THR <= 4).  THR = 0 is the textbook update.
            const float m_new = mt * sc > mc_run + PL_THR ? fmaxf(m_run, mt) : m_run;
            const float mc_new = m_new > -INFINITY ? m_new * sc : 0.f;
            const float alpha = __builtin_amdgcn_exp2f(mc_run - mc_new);
            // P is produced SCALED by 2048 (the +11 in the exponent; the scale cancels in O / l): P' = h' + l'' with h' = RNE16(P') and the
            // residual l'' = P' - h' taken UNSCALED (|l''| <= 2^-11 P' needs no further scaling to stay a normal fp16 for every P that matters),
            // so the split is one cvt_pk per pair and one mixed fma per element, no multiplies; in the second product l'' V_h then has the
            // scale of h' V_h and shares its accumulator, only h' V_l goes to the 1/2048 accumulator.
            const float shift = 11.0f - mc_new;
            float ls = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    u32x4 h4, l4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[s][8 * u + 2 * e], sc, shift));
                        const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[s][8 * u + 2 * e + 1], sc, shift));
                        ls += p0 + p1;
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        typedef _Float16 h2t __attribute__((ext_vector_type(2)));
                        const unsigned h2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{p0, p1}, h2t));
                        unsigned l2;
                        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l2) : "v"(h2), "v"(p0));
                        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l2) : "v"(h2), "v"(p1));
                        h4[e] = h2;
                        l4[e] = l2;
                    }
                    ph[2 * s + u] = h4;
                    pl[2 * s + u] = l4;
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
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FREE) {
            if constexpr (LATE) {
                if (t + 2 < nt) issue(t + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            pv(t);
        } else {
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!FREE && grp == 0) __builtin_amdgcn_s_barrier();  // pairs with group 1's stagger barrier

#pragma unroll
    for (int d = 0; d < 2; ++d) ot[d] += otc[d] * (1.0f / CBX_F16_LO_SCALE);
    const float l_tot = l_run + half_swap(l_run);
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

// default 4 since round 4: same-box A/B at the bench shape (profiles/r04_attn_planes_ab.log) 111.1-111.2 us against 114.4-115.5 for version 2, twice in a row
static int g_attn_pl_version = 0;  // TEST HOOK (cbx_set_attn_planes_version; 0 = automatic); callers pass their own version per call
extern "C" int cbx_set_attn_planes_version(int v) {
    g_attn_pl_version = v;
    return 0;
}

extern "C" int cbx_flash_attn_planes_v(const void* q, const void* k, const void* vt, void* o, const int* key_lens, int nz1, int n_heads, int Tq,
                                       int Tk, long q_sb, long q_st, long q_lo, long k_sb, long k_st, long k_lo, long vt_sb, long vt_sd,
                                       long vt_lo, long o_sb, long o_st, long o_lo, float scale, int causal, int version, void* stream) {
    CBX_REQUIRE(version >= 0 && version <= 6, "flash_attn_planes: version %d (0 = default, 1 .. 6)", version);
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
    // versions 2 / 4 (256 queries per workgroup; 4 = the free-running loop, the default) serve the non-causal case with 31-bit K / V^T offsets; cbx_set_attn_planes_version(1)
    // keeps the one-group kernel (A/B: scripts/bench_planes.py), 3 = version 2 with s_setprio 1 around the matrix block (measured: no gain)
    const int forced = version ? version : g_attn_pl_version;  // the call's own choice first (ABI v13); the process-wide knob is a test hook
    // automatic = version 4 (256 queries per workgroup, three stages); on small grids its 128-query twin (below); for SHORT key sequences on a filled chip the one-group
    // kernel of version 1 (128 queries, two workgroups per CU): the 9 - 12 KV tiles of T <= 768 do not amortise version 4's prologue and its last, part-filled query
    // tile -- 16 rows: T 530 59.3 -> 54.0 us, T 730 76.9 -> 72.4, T 1000 110.8 against 120.3 (profiles/r06_ag_plane_attention_short_sequences.log)
    const bool short_seq = forced == 0 && Tk <= 768 && (long)((Tq + 255) / 256) * n_heads * nz1 > 128;
    const int ver = forced ? forced : short_seq ? 1 : 4;
    const bool v2ok = !causal && k_st >= k_lo + 64 && (long)Tk * k_st * 2 < 0x7fffffffL && 64 * vt_sd * 2 < 0x7fffffffL && vt_sd >= vt_lo;
    if (ver >= 2 && v2ok) {
        constexpr int lds = 3 * PL_STAGE;
        static unsigned long long configured = 0;  // one bit per device ordinal
        const int dev = cbx_device();
        if (!(configured >> dev & 1)) {
            hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn_pl2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn_pl2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void*>((flash_attn_pl2_kernel<false, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipError_t e4 = hipFuncSetAttribute(reinterpret_cast<const void*>((flash_attn_pl2_kernel<false, true, 4>)), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipError_t e5 = hipFuncSetAttribute(reinterpret_cast<const void*>((flash_attn_pl2_kernel<false, true, 8, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e4 == hipSuccess) e4 = e5;
            if (e3 == hipSuccess) e3 = e4;
            if (e2 == hipSuccess) e2 = e3;
            if (e1 != hipSuccess || e2 != hipSuccess) return cbx_set_error((int)(e1 != hipSuccess ? e1 : e2), "flash_attn_planes: cannot reserve %d B of LDS", lds);
            configured |= 1ull << dev;
        }
        dim3 grid2((Tq + 255) / 256, n_heads, nz1);
        // a part-filled chip (batch 1: 2 rows x 8 heads x 4 query tiles = 64 workgroups of 256 queries): the 128-query form doubles the workgroups -- 44.0 -> 31.5 us at
        // 2 rows x T 1000, 44.8 -> 35.3 at 4 rows; from 224 workgroups on the 256-query form is ahead again (profiles/r06_z_plane_attention_small_grids.log).  The
        // automatic choice only: an explicit version (per call or through the test hook) is what runs.
        const bool small_grid = forced == 0 && (long)grid2.x * n_heads * nz1 <= 128;
        if (ver == 5 || small_grid) hipLaunchKernelGGL((flash_attn_pl2_kernel<false, true, 4>), dim3((Tq + 127) / 128, n_heads, nz1), dim3(256), lds, (hipStream_t)stream, a);
        else if (ver == 6) hipLaunchKernelGGL((flash_attn_pl2_kernel<false, true, 8, true>), grid2, dim3(512), lds, (hipStream_t)stream, a);
        else if (ver == 3) hipLaunchKernelGGL(flash_attn_pl2_kernel<true>, grid2, dim3(512), lds, (hipStream_t)stream, a);
        else if (ver == 4) hipLaunchKernelGGL((flash_attn_pl2_kernel<false, true>), grid2, dim3(512), lds, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(flash_attn_pl2_kernel<false>, grid2, dim3(512), lds, (hipStream_t)stream, a);
        return cbx_check_launch("flash_attn_planes");
    }
    dim3 grid((Tq + 127) / 128, n_heads, nz1);
    hipLaunchKernelGGL(flash_attn_pl_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    return cbx_check_launch("flash_attn_planes");
}

extern "C" int cbx_flash_attn_planes(const void* q, const void* k, const void* vt, void* o, const int* key_lens, int nz1, int n_heads, int Tq,
                                     int Tk, long q_sb, long q_st, long q_lo, long k_sb, long k_st, long k_lo, long vt_sb, long vt_sd,
                                     long vt_lo, long o_sb, long o_st, long o_lo, float scale, int causal, void* stream) {
    return cbx_flash_attn_planes_v(q, k, vt, o, key_lens, nz1, n_heads, Tq, Tk, q_sb, q_st, q_lo, k_sb, k_st, k_lo, vt_sb, vt_sd, vt_lo, o_sb, o_st, o_lo, scale,
                                   causal, 0, stream);
}
