// Attention kernels (fp32 exact): flash-style tiled attention on v_mfma_f32_32x32x2_f32, single-query decode
// attention streaming the KV cache, and the materialised rel-pos softmax of the conformer encoder.
#include <stdlib.h>
#include "cbx_common.h"

namespace {
CBX_TRC_TU


// ------------------------------------------------------------------------------------------------------------
// Flash attention, head_dim 64.  Workgroup = 4 waves x 32 queries; KV walked in 64-key tiles staged in LDS.
//
// "Swapped" formulation so that softmax statistics are lane-local: the wave computes S^T = K Q^T, whose MFMA
// C/D map puts query (lane&31) in the column and 16 keys in the lane's registers; the row max / sum of a query
// is then 16 in-register ops + ONE cross-half shuffle.  O^T = V^T P^T uses the same column = query map, so the
// exponentiated P registers feed the second MFMA directly as its B operand (no LDS round trip for P) and the
// online-softmax rescale is a per-lane scalar.  The d (resp. key) contraction order is permuted so that the two
// half-waves own d in [0,32) / [32,64) (resp. keys r and r+4) -- legal because MFMA sums both halves' k.
// ------------------------------------------------------------------------------------------------------------
constexpr int KT = 64;        // keys per LDS tile
constexpr int K_LD = 65;      // K tile row stride (floats): b32 reads by 32 consecutive keys are conflict-free
constexpr int V_LD = 64;

struct FlashArgs {
    const float* q; const float* k; const float* v; float* o; const int* key_lens;
    int Tq, Tk;
    long q_sb, q_st, k_sb, k_st, v_sb, v_st, o_sb, o_st;
    float scale;
    int causal;  // 0 = none; else key j allowed iff j <= i + (Tk - Tq)
    long k_sh = 64, v_sh = 64;  // head strides of K / V (ABI v15: 64 = heads side by side in a token row; the KV cache has them max_ctx * 64 apart)
};

template <bool PREFETCH>
__global__ __launch_bounds__(256) void flash_attn_f32_kernel(const FlashArgs a) {
    __shared__ __attribute__((aligned(16))) float Ks[KT * K_LD];
    __shared__ __attribute__((aligned(16))) float Vs[KT * V_LD];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    // the query tiles of one (row, head) share K/V: keep them on one XCD
    const int tile = cbx_xcd_remap((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gridDim.x * gridDim.y * gridDim.z);
    const int qt = tile % gridDim.x, head = (tile / gridDim.x) % gridDim.y, z = tile / (gridDim.x * gridDim.y);
    const int q0 = qt * 128;
    const int qi = q0 + wid * 32 + lr;  // this lane's query
    const float* qb = a.q + (long)z * a.q_sb + head * 64;
    const float* kb = a.k + (long)z * a.k_sb + head * a.k_sh;
    const float* vb = a.v + (long)z * a.v_sb + head * a.v_sh;
    const int klen = a.key_lens ? min(a.Tk, a.key_lens[z]) : a.Tk;
    const int coff = a.Tk - a.Tq;

    // Q fragment: d = s + 32*lh, pre-scaled
    float qreg[32];
    {
        const bool ok = qi < a.Tq;
        const float* qp = qb + (long)(ok ? qi : 0) * a.q_st + 32 * lh;
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
            f32x4 t = *reinterpret_cast<const f32x4*>(qp + s4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) qreg[s4 * 4 + e] = ok ? t[e] * a.scale : 0.f;
        }
    }

    f32x16 ot[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    int kend = klen;
    if (a.causal) kend = min(kend, q0 + 128 + coff);  // keys beyond the block's last query are never visible
    const int ld_row = tid >> 2, ld_c = (tid & 3) * 16;

    // K/V staging: the next tile's 8 float4 per thread are fetched into registers while the current tile is consumed
    f32x4 kreg[4], vreg[4];
    auto fetch = [&](int j0) {
        const int j = j0 + ld_row;
        const bool ok = j < klen;
        const float* kp = kb + (long)(ok ? j : 0) * a.k_st + ld_c;
        const float* vp = vb + (long)(ok ? j : 0) * a.v_st + ld_c;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            kreg[c] = ok ? *reinterpret_cast<const f32x4*>(kp + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            vreg[c] = ok ? *reinterpret_cast<const f32x4*>(vp + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) Ks[ld_row * K_LD + ld_c + c * 4 + e] = kreg[c][e];
            *reinterpret_cast<f32x4*>(&Vs[ld_row * V_LD + ld_c + c * 4]) = vreg[c];
        }
    };
    if (PREFETCH && kend > 0) fetch(0);

    for (int j0 = 0; j0 < kend; j0 += KT) {
        if (!PREFETCH) fetch(j0);
        stage();
        __syncthreads();
        if (PREFETCH && j0 + KT < kend) fetch(j0 + KT);

        // ---- S^T = K Q^T  (2 sub-tiles of 32 keys)
        f32x16 st[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
            const float* kr = &Ks[(t * 32 + lr) * K_LD + 32 * lh];
#pragma unroll
            for (int s = 0; s < 32; ++s) st[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[s], qreg[s], st[t], 0, 0, 0);
        }

        // ---- mask + online softmax (lane owns query qi; registers hold keys row(r) + 4*lh of each sub-tile)
        float mt = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int j = j0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                bool vis = j < klen && (!a.causal || j <= qi + coff);
                float sv = vis ? st[t][r] : -INFINITY;
                st[t][r] = sv;
                mt = fmaxf(mt, sv);
            }
        mt = fmaxf(mt, cbx_xor_lane<32>(mt));
        const float m_new = fmaxf(m_run, mt);
        float alpha = 1.f;
        if (m_new > -INFINITY) alpha = __expf(m_run - m_new);  // m_run = -inf -> 0
        float ls = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float pv = (m_new > -INFINITY) ? __expf(st[t][r] - m_new) : 0.f;
                st[t][r] = pv;
                ls += pv;
            }
        l_run = l_run * alpha + ls;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[d][r] *= alpha;

        // ---- O^T += V^T P^T : MFMA step (t,r) contracts keys t*32 + row(r) (+4 for the upper half-wave)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float* vr = &Vs[key * V_LD + lr];
                ot[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[0], st[t][r], ot[0], 0, 0, 0);
                ot[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[32], st[t][r], ot[1], 0, 0, 0);
            }
        __syncthreads();
    }

    // ---- finalise: both half-waves hold partial sums of the same query
    const float l_tot = l_run + cbx_xor_lane<32>(l_run);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (qi < a.Tq) {
        float* op = a.o + (long)z * a.o_sb + (long)qi * a.o_st + head * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 t = {ot[d][g * 4 + 0] * inv, ot[d][g * 4 + 1] * inv, ot[d][g * 4 + 2] * inv, ot[d][g * 4 + 3] * inv};
                *reinterpret_cast<f32x4*>(op + d * 32 + 8 * g + 4 * lh) = t;
            }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Flash attention with the relative-position term of the conformer encoder (RelPositionMultiHeadedAttention, reference
// transformer/attention.py:249-330), exact fp32:  o_i = softmax_j( scale * ((q_i + u) k_j + (q_i + v) p_{T-1-i+j}) ) V  without the
// (T, T) and (T, 2T-1) score tensors of the materialised path (cbx_gemm_f32 x 3 + cbx_softmax_relpos_f32: 16 T^2 floats per head).
//
// Same swapped formulation as flash_attn_f32_kernel (S^T = K Q^T: a lane owns one query, softmax statistics are lane-local).  The
// position term of a 32-key x 32-query block is a diagonal band of G^T = P_rows (q + v)^T:  bd^T[jj][c] = G^T[R0 + jj - c][c], a shift
// along the ROW index by the lane's own column -- i.e. along the register index of the MFMA C layout, by a lane-dependent amount.  That
// needs one trip through memory: every wave keeps a ring of three 32-row G^T blocks in LDS ([position row][query], row stride 40 floats:
// bank = 40 row + c is conflict-free for the C-layout stores and for the diagonal loads alike) and reads its band back.  Position rows
// advance with the keys: a 64-key tile needs blocks 2n, 2n + 1, 2n + 2 of the wave's row window, and block 2n is the previous tile's
// 2n + 2 -- so the term costs 64 MFMAs per key tile, as many as q k^T itself (not the 2x of the full (T, 2T-1) product).
// The rows of P are read straight from global memory as the MFMA A operand (P is (2T-1, n_heads * 64) floats: L2-resident).
// ------------------------------------------------------------------------------------------------------------
constexpr int G_LD = 40;               // row stride (floats) of a G^T block
constexpr int G_WAVE = 3 * 32 * G_LD;  // floats per wave: ring of three blocks
constexpr int RELPOS_LDS = (KT * K_LD + KT * V_LD + 4 * G_WAVE) * 4;

struct FlashRelArgs {
    const float* qu; const float* qv; const float* k; const float* v; const float* pp; float* o; const int* key_lens;
    int T, P;
    long q_sb, q_st, pp_st, o_sb, o_st;
    float scale;
};

__global__ __launch_bounds__(256) void flash_relpos_f32_kernel(const FlashRelArgs a) {
    extern __shared__ __attribute__((aligned(16))) float rel_smem[];
    float* Ks = rel_smem;
    float* Vs = rel_smem + KT * K_LD;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    float* Gw = rel_smem + KT * K_LD + KT * V_LD + wid * G_WAVE;
    const int tile = cbx_xcd_remap((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gridDim.x * gridDim.y * gridDim.z);
    const int qt = tile % gridDim.x, head = (tile / gridDim.x) % gridDim.y, z = tile / (gridDim.x * gridDim.y);
    const int q0 = qt * 128;
    const int i0w = q0 + wid * 32;  // the wave's first query
    const int qi = i0w + lr;        // this lane's query
    const long zoff = (long)z * a.q_sb + head * 64;
    const float* kb = a.k + zoff;
    const float* vb = a.v + zoff;
    const float* pb = a.pp + head * 64 + 32 * lh;
    const int klen = a.key_lens ? min(a.T, a.key_lens[z]) : a.T;
    const int rb0 = a.T - 1 - i0w - 31;  // position row of (block 0, row 0) of this wave's window: block b covers rows rb0 + 32 b .. + 31

    // (q + u) and (q + v) fragments of this lane's query: d = s + 32*lh, pre-scaled
    float qreg[32], qvreg[32];
    {
        const bool ok = qi < a.T;
        const long off = zoff + (long)(ok ? qi : 0) * a.q_st + 32 * lh;
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
            f32x4 t = *reinterpret_cast<const f32x4*>(a.qu + off + s4 * 4);
            f32x4 w = *reinterpret_cast<const f32x4*>(a.qv + off + s4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                qreg[s4 * 4 + e] = ok ? t[e] * a.scale : 0.f;
                qvreg[s4 * 4 + e] = ok ? w[e] * a.scale : 0.f;
            }
        }
    }

    f32x16 ot[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int ld_row = tid >> 2, ld_c = (tid & 3) * 16;
    f32x4 kreg[4], vreg[4];
    auto fetch = [&](int j0) {
        const int j = j0 + ld_row;
        const bool ok = j < klen;
        const float* kp = kb + (long)(ok ? j : 0) * a.q_st + ld_c;
        const float* vp = vb + (long)(ok ? j : 0) * a.q_st + ld_c;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            kreg[c] = ok ? *reinterpret_cast<const f32x4*>(kp + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            vreg[c] = ok ? *reinterpret_cast<const f32x4*>(vp + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) Ks[ld_row * K_LD + ld_c + c * 4 + e] = kreg[c][e];
            *reinterpret_cast<f32x4*>(&Vs[ld_row * V_LD + ld_c + c * 4]) = vreg[c];
        }
    };
    // G^T block b of this wave's window -> ring slot b % 3:  G^T[row][c] = sum_d P[rb0 + 32 b + row][d] * scale (q + v)[i0w + c][d].
    // The rows of P of the NEXT key tile's two blocks are requested right after this tile's barrier (with its K / V rows) and held in
    // registers across the tile: with one wave per SIMD (the ring's LDS) nothing else would hide their L2 round trip.
    auto p_load = [&](int b, float (&dst)[32]) {
        const int r = min(max(rb0 + 32 * b + lr, 0), a.P - 1);  // rows outside P only meet masked keys / absent queries
        const float* pr = pb + (long)r * a.pp_st;
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
            f32x4 t = *reinterpret_cast<const f32x4*>(pr + s4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[s4 * 4 + e] = t[e];
        }
    };
    auto g_block = [&](int b, const float (&preg)[32]) {
        f32x16 g;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) g[rr] = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) g = __builtin_amdgcn_mfma_f32_32x32x2f32(preg[s], qvreg[s], g, 0, 0, 0);
        float* gs = Gw + (b % 3) * (32 * G_LD) + lr;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) gs[((rr & 3) + 8 * (rr >> 2) + 4 * lh) * G_LD] = g[rr];
    };
    float pn0[32], pn1[32];
    if (klen > 0) {
        fetch(0);
        p_load(1, pn0);
        p_load(2, pn1);
    }

    for (int j0 = 0, n = 0; j0 < klen; j0 += KT, ++n) {
        stage();
        if (n == 0) {
            float p0[32];
            p_load(0, p0);
            g_block(0, p0);
        }
        g_block(2 * n + 1, pn0);
        g_block(2 * n + 2, pn1);
        __syncthreads();  // K / V tile staged by all waves; this wave's G^T blocks written by all of its lanes
        if (j0 + KT < klen) {
            fetch(j0 + KT);
            p_load(2 * n + 3, pn0);
            p_load(2 * n + 4, pn1);
        }

        // ---- S^T = K (q + u)^T  (2 sub-tiles of 32 keys) + the diagonal band of G^T
        f32x16 st[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
            const float* kr = &Ks[(t * 32 + lr) * K_LD + 32 * lh];
#pragma unroll
            for (int s = 0; s < 32; ++s) st[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[s], qreg[s], st[t], 0, 0, 0);
            // key jj = row(r) of sub-tile t, query c = lr: window row 31 + 64 n + 32 t + jj - c = block 2n + t (jj <= c) or 2n + t + 1, row (31 + jj - c) & 31
            const float* gA = Gw + ((2 * n + t) % 3) * (32 * G_LD) + lr;
            const float* gB = Gw + ((2 * n + t + 1) % 3) * (32 * G_LD) + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jj = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int dlt = 31 + jj - lr;
                st[t][r] += (dlt >= 32 ? gB : gA)[(dlt & 31) * G_LD];
            }
        }

        // ---- mask + online softmax (lane owns query qi; registers hold keys row(r) + 4*lh of each sub-tile)
        float mt = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float sv = j < klen ? st[t][r] : -INFINITY;
                st[t][r] = sv;
                mt = fmaxf(mt, sv);
            }
        mt = fmaxf(mt, cbx_xor_lane<32>(mt));
        const float m_new = fmaxf(m_run, mt);
        float alpha = 1.f;
        if (m_new > -INFINITY) alpha = __expf(m_run - m_new);  // m_run = -inf -> 0
        float ls = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = (m_new > -INFINITY) ? __expf(st[t][r] - m_new) : 0.f;
                st[t][r] = pv;
                ls += pv;
            }
        l_run = l_run * alpha + ls;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[d][r] *= alpha;

        // ---- O^T += V^T P^T : MFMA step (t,r) contracts keys t*32 + row(r) (+4 for the upper half-wave)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float* vr = &Vs[key * V_LD + lr];
                ot[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[0], st[t][r], ot[0], 0, 0, 0);
                ot[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[32], st[t][r], ot[1], 0, 0, 0);
            }
        __syncthreads();  // the tile (and ring slots 2n, 2n + 1: overwritten as 2n + 3, 2n + 4) are free again
    }

    // ---- finalise: both half-waves hold partial sums of the same query
    const float l_tot = l_run + cbx_xor_lane<32>(l_run);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (qi < a.T) {
        float* op = a.o + (long)z * a.o_sb + (long)qi * a.o_st + head * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 t = {ot[d][g * 4 + 0] * inv, ot[d][g * 4 + 1] * inv, ot[d][g * 4 + 2] * inv, ot[d][g * 4 + 3] * inv};
                *reinterpret_cast<f32x4*>(op + d * 32 + 8 * g + 4 * lh) = t;
            }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Decode attention: one query row per (row, head), KV cache [row][head][pos][64] streamed once.
// 4 waves split the context; inside a wave 16 lanes x float4 cover one key row (4 keys per wave-iteration,
// 1 KiB contiguous per load instruction).  Scores go through LDS (two-pass softmax), then V is streamed.
// ------------------------------------------------------------------------------------------------------------
constexpr int DEC_MAX_CTX = 8192;

__global__ __launch_bounds__(256) void decode_attn_f32_kernel(const float* __restrict__ q, const float* __restrict__ kc,
                                                              const float* __restrict__ vc, float* __restrict__ o,
                                                              const int* __restrict__ ctx_lens, long q_ld, long o_ld,
                                                              long row_stride, long head_stride, float scale) {
    __shared__ float sc[DEC_MAX_CTX];
    __shared__ float red[8];
    __shared__ __attribute__((aligned(16))) float oacc[4][64];
    const int row = blockIdx.y, head = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int sub = lane >> 4, l16 = lane & 15;  // key-in-group, float4 index within the 64-d row
    const int ctx = min(ctx_lens[row], DEC_MAX_CTX);
    const float* kb = kc + (long)row * row_stride + (long)head * head_stride;
    const float* vb = vc + (long)row * row_stride + (long)head * head_stride;
    f32x4 qv = *reinterpret_cast<const f32x4*>(q + (long)row * q_ld + head * 64 + l16 * 4);
    qv *= scale;

    // pass 1: scores
    float mx = -INFINITY;
    for (int p0 = wid * 4; p0 < ctx; p0 += 16) {
        int pos = p0 + sub;
        float d = 0.f;
        if (pos < ctx) {
            f32x4 kv = *reinterpret_cast<const f32x4*>(kb + (long)pos * 64 + l16 * 4);
            d = kv[0] * qv[0] + kv[1] * qv[1] + kv[2] * qv[2] + kv[3] * qv[3];
        }
        d += cbx_xor_lane<8>(d);
        d += cbx_xor_lane<4>(d);
        d += cbx_xor_lane<2>(d);
        d += cbx_xor_lane<1>(d);
        if (pos < ctx) {
            if (l16 == 0) sc[pos] = d;
            mx = fmaxf(mx, d);
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wid] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int p = tid; p < ctx; p += 256) {
        float e = __expf(sc[p] - mx);
        sc[p] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wid] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);

    // pass 2: weighted V
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int p0 = wid * 4; p0 < ctx; p0 += 16) {
        int pos = p0 + sub;
        if (pos < ctx) {
            f32x4 vv = *reinterpret_cast<const f32x4*>(vb + (long)pos * 64 + l16 * 4);
            float pw = sc[pos];
            acc += vv * pw;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        acc[e] += cbx_xor_lane<16>(acc[e]);
        acc[e] += cbx_xor_lane<32>(acc[e]);
    }
    if (sub == 0) *reinterpret_cast<f32x4*>(&oacc[wid][l16 * 4]) = acc;
    __syncthreads();
    if (tid < 64) {
        float t = (oacc[0][tid] + oacc[1][tid] + oacc[2][tid] + oacc[3][tid]) * inv;
        o[(long)row * o_ld + head * 64 + tid] = t;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Fused decode attention: RoPE(q,k) + KV-cache append + single-pass online-softmax attention, one workgroup per
// (row, head).  Replaces three launches (rope_kv, decode_attn and their round trip through HBM) on the per-token
// critical path.  Each 16-lane group owns one key row per step (float4 per lane = one 256-B K row and V row), four
// steps are issued back to back so that 8 KiB of K/V per wave are in flight; the running (max, sum, acc) state of the
// 16 lane groups is merged through LDS at the end.  The new token's k/v are taken from LDS (never re-read from HBM).
// ------------------------------------------------------------------------------------------------------------
// U = key rows per lane group in flight per step (env CBX_DA_U: 4 / 8 / 16).  The first step's K/V loads are issued BEFORE the
// RoPE / LDS hand-off of q (they do not depend on it), so the q path's global round trip overlaps the cache stream.
// PIPE (cbx_set_decode_attn_pipeline, ABI v9; written without GPU access at the end of round 3, verified on the SIMT emulator, to be timed in
// round 4): the K/V rows of step i + 1 are requested BEFORE step i is multiplied (two register sets, loop unrolled by two; the loads
// stay unconditional -- clamped addresses -- so that hipcc waits with a COUNTED vmcnt and the younger set stays in flight).  The plain form
// issues a step's loads only after the previous step's arithmetic, i.e. it pays one memory round trip per step (64 positions with U = 4:
// the measured slope of 1.6 us per 64 positions, profiles/r02_decode_micro.log, is that round trip).  Same arithmetic, same order, same results.
// NT (cbx_set_decode_attn_pipeline(2 | 3), same provenance): the K / V rows are read with the non-temporal policy -- a (row, head)'s cache is
// streamed once per token step by one CU, the case for which MI355X_MICROARCH.md measures nt loads 5-10 % ahead on a decode layer.
// SPEC (cbx_set_decode_attn_pipeline bit 2, 4 rows per step; written after the GPU budget of round 3 was spent, emulator-verified, timed by
// the autotuner): the FIRST step's K / V rows (positions 0 .. 63 of the (row, head)) are requested before positions[row] has arrived -- their
// addresses do not depend on it -- instead of after it: one dependent memory round trip less on a launch that is a chain of five or six.
// Rows at or past the new position are then discarded (k, v := 0 before the step: they are masked to -inf / weight 0 either way, and the
// clamped form multiplied row 0's finite values by the same 0), so a cache holding anything at all past its context stays harmless.
// Needs 64 positions of cache behind every (row, head) (head_stride >= 64 * 64; rows beyond it are clamped to row 0).
template <int DA_U, bool SPLIT, bool PIPE = false, bool NT = false, bool SPEC = false>
__global__ __launch_bounds__(256) void decode_attn_rope_kernel(const float* __restrict__ qkv, const int* __restrict__ positions,
                                                               const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                               float* __restrict__ kc, float* __restrict__ vc, float* __restrict__ o,
                                                               int n_heads, long ld_qkv, long o_ld, int o_packed, long row_stride,
                                                               long head_stride, float scale, float* split_ws, int* split_cnt,
                                                               int split_min_ctx, int qkv_nparts, long qkv_part_stride,
                                                               const float* __restrict__ qkv_ssq, float rms_dim, float rms_eps) {
    __builtin_amdgcn_s_setprio(3);  // decode-step kernels are latency-bound and issue little: beside a co-resident workgroup of another stream (the throughput schedule, profiles/r05_overlap_*) their waves go first at the SIMD's issue arbiter; alone on the CU it changes nothing
    // gridDim.z = S > 1: the context of a (row, head) is split over S workgroups (on S CUs: one workgroup cannot pull a long context
    // faster than its CU's memory path, 50-60 GB/s); each leaves {max, sum, 64 numerators} in split_ws and the LAST to arrive (a ticket on
    // split_cnt, agent-scope release before it, acquire after it: cdna_hip_programming.md guideline 16) merges them in split order.
    __shared__ __attribute__((aligned(16))) float q_s[64], k_new[64], v_new[64];
    __shared__ __attribute__((aligned(16))) float st_acc[16][64];
    __shared__ float st_m[16], st_l[16];
    __shared__ int s_last[1];
    CBX_TRC_DECL;
    CBX_TRC_STAMP(0);  // entry
    const int row = blockIdx.y, head = blockIdx.x;
    // !SPLIT: the one-workgroup form, compiled without any of the hand-off.  SPLIT: a row whose context is shorter than split_min_ctx is
    // still walked by ONE workgroup (split 0; the others leave at once): the hand-off costs ~5 us on the critical path (release, ticket,
    // acquire, merge), more than the 1-2 round trips a short context takes (measured: batch-1 Llama decode +15 % with an unconditional split)
    const int sp = SPLIT ? (int)blockIdx.z : 0;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int sub = lane >> 4, l16 = lane & 15;
    float* kb = kc + (long)row * row_stride + (long)head * head_stride;
    float* vb = vc + (long)row * row_stride + (long)head * head_stride;
    // lane group g = wid*4 + sub handles positions g, g+16, g+32, ...
    f32x4 kv[DA_U], vv[DA_U];
    f32x4 kv2[PIPE ? DA_U : 1], vv2[PIPE ? DA_U : 1];  // PIPE: the second register set
    if constexpr (SPEC) {  // split 0's first step (positions wid*4 + sub + 16 u), before anything that depends on positions[]
#pragma unroll
        for (int u = 0; u < DA_U; ++u) {
            const int p = wid * 4 + sub + 16 * u;
            const int pc = (long)(p + 1) * 64 <= head_stride ? p : 0;
            if constexpr (NT) {
                kv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(kb + (long)pc * 64 + l16 * 4));
                vv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vb + (long)pc * 64 + l16 * 4));
            } else {
                kv[u] = *reinterpret_cast<const f32x4*>(kb + (long)pc * 64 + l16 * 4);
                vv[u] = *reinterpret_cast<const f32x4*>(vb + (long)pc * 64 + l16 * 4);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const int S = SPLIT && positions[blockIdx.y] + 1 >= split_min_ctx ? (int)gridDim.z : 1;
    if (SPLIT && S == 1 && sp != 0) return;
    const int pos = positions[row];
    const int ctx = pos + 1;
    auto load_rows = [&](f32x4* kd, f32x4* vd, int p0) {
#pragma unroll
        for (int u = 0; u < DA_U; ++u) {
            const int p = p0 + 16 * u;
            const int pc = p < pos ? p : 0;  // clamped: loads are unconditional (never the new position, never past the end)
            if constexpr (NT) {
                kd[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(kb + (long)pc * 64 + l16 * 4));
                vd[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vb + (long)pc * 64 + l16 * 4));
            } else {
                kd[u] = *reinterpret_cast<const f32x4*>(kb + (long)pc * 64 + l16 * 4);
                vd[u] = *reinterpret_cast<const f32x4*>(vb + (long)pc * 64 + l16 * 4);
            }
        }
    };
    auto load_chunk = [&](int p0) { load_rows(kv, vv, p0); };
    // this workgroup's positions: [p_lo, p_hi), 16-aligned slices of [0, ctx); the new token (position pos) belongs to the last slice
    const int slice = S > 1 ? ((ctx + 16 * S - 1) / (16 * S)) * 16 : ctx;
    const int p_lo = sp * slice, p_hi = min(ctx, p_lo + slice);
    int p0 = p_lo + wid * 4 + sub;
    if (!SPEC || p_lo != 0) load_chunk(p0);  // SPEC: already on its way (a later split of a long context starts elsewhere: requested now)
    // SPEC: rows of the first step at or past the new position hold whatever the cache holds there -- dropped when the step consumes them
    auto drop_unwritten = [&](f32x4* kd, f32x4* vd, int p0) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < DA_U; ++u)
            if (p0 + 16 * u >= pos) kd[u] = z4, vd[u] = z4;
    };
    // (pos == 0, any form: the clamped loads fetched row 0, which is the not yet written slot of the new token itself -- the only case in
    // which the clamp does not land on a finite row of a cache that holds garbage past its context)
    bool spec_first = (SPEC && p_lo == 0) || pos == 0;

    if (wid == 0) {
        const float* qp = qkv + (long)row * ld_qkv + head * 64;
        float qv = qp[lane], kn0 = qp[(long)n_heads * 64 + lane], vn0 = qp[(long)n_heads * 128 + lane];
        if (qkv_nparts > 1) {  // uniform (kernel argument): qkv holds split-K partial sums of the RMSNorm-folded projection (cbx_gemv_t.col_tiles,
            // ksplit > 1): added in fixed order, then rstd[row] of the projection's input.  Loads unconditional (clamped part index), one round trip
            float sq = qkv_ssq[row];
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                const bool onj = j < qkv_nparts;
                const float* qj = qp + (onj ? j : 0) * qkv_part_stride;
                const float a = qj[lane], b = qj[(long)n_heads * 64 + lane], cc = qj[(long)n_heads * 128 + lane], sj = qkv_ssq[(onj ? j : 0) * 16 + row];
                if (onj) qv += a, kn0 += b, vn0 += cc, sq += sj;
            }
            const float rstd = rsqrtf(sq / rms_dim + rms_eps);  // the expression of gemv_kernel / gemv_ct_kernel with ksplit == 1
            qv *= rstd, kn0 *= rstd, vn0 *= rstd;
        }
        const float c = cos_t ? cos_t[(long)pos * 64 + lane] : 1.f, s = cos_t ? sin_t[(long)pos * 64 + lane] : 0.f;  // GPT-2: no RoPE
        const float sgn = lane < 32 ? -1.f : 1.f;
        const float qn = qv * c + sgn * cbx_xor_lane<32>(qv) * s;
        const float kn = kn0 * c + sgn * cbx_xor_lane<32>(kn0) * s;
        q_s[lane] = qn * scale;
        k_new[lane] = kn;
        v_new[lane] = vn0;
        if (sp == S - 1) {  // one workgroup appends the new token to the cache
            cbx_store_out(kb + (long)pos * 64 + lane, kn);
            cbx_store_out(vb + (long)pos * 64 + lane, vn0);
        }
    }
    __syncthreads();
    CBX_TRC_STAMP(1);  // q / k / v of the new token roped and in LDS (positions[], qkv row and cos / sin have arrived)
    const f32x4 qv4 = *reinterpret_cast<const f32x4*>(&q_s[l16 * 4]);
    float m = -INFINITY, l = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // one step: DA_U key rows of this lane group (positions p0, p0 + 16, ...) folded into the running (m, l, acc)
    auto step = [&](f32x4* kd, f32x4* vd, int p0) {
        float d[DA_U];
        float mt = m;
#pragma unroll
        for (int u = 0; u < DA_U; ++u) {
            const int p = p0 + 16 * u;
            if (p == pos) {  // the new token's k/v come from LDS, never from HBM
                kd[u] = *reinterpret_cast<const f32x4*>(&k_new[l16 * 4]);
                vd[u] = *reinterpret_cast<const f32x4*>(&v_new[l16 * 4]);
            }
            float t = kd[u][0] * qv4[0] + kd[u][1] * qv4[1] + kd[u][2] * qv4[2] + kd[u][3] * qv4[3];
            t += cbx_xor_lane<8>(t);
            t += cbx_xor_lane<4>(t);
            t += cbx_xor_lane<2>(t);
            t += cbx_xor_lane<1>(t);
            d[u] = p < p_hi ? t : -INFINITY;
            mt = fmaxf(mt, d[u]);
        }
        if (mt > -INFINITY) {
            const float a = __expf(m - mt);
            acc *= a;
            l *= a;
#pragma unroll
            for (int u = 0; u < DA_U; ++u) {
                const float pw = __expf(d[u] - mt);
                l += pw;
                acc += vd[u] * pw;
            }
            m = mt;
        }
    };
    if constexpr (PIPE) {
        constexpr int STEP = 16 * DA_U;
        while (true) {
            load_rows(kv2, vv2, p0 + STEP);  // requested before the arithmetic on the set that has landed
            __builtin_amdgcn_sched_barrier(0);
            if (spec_first) drop_unwritten(kv, vv, p0), spec_first = false;
            step(kv, vv, p0);
            p0 += STEP;
            if (p0 >= p_hi) break;
            load_rows(kv, vv, p0 + STEP);
            __builtin_amdgcn_sched_barrier(0);
            step(kv2, vv2, p0);
            p0 += STEP;
            if (p0 >= p_hi) break;
        }
    } else {
        if (spec_first) drop_unwritten(kv, vv, p0);
        while (true) {
            step(kv, vv, p0);
            p0 += 16 * DA_U;
            if (p0 >= p_hi) break;
            load_chunk(p0);
        }
    }
#ifdef CBX_TRACE
    asm volatile("s_nop 0" ::"v"(acc[0]));
    CBX_TRC_STAMP(2);  // context walked (wave 0)
#endif
    const int g = wid * 4 + sub;
    *reinterpret_cast<f32x4*>(&st_acc[g][l16 * 4]) = acc;
    if (l16 == 0) {
        st_m[g] = m;
        st_l[g] = l;
    }
    __syncthreads();
    float M = -INFINITY, num = 0.f, den = 0.f;
    if (tid < 64) {
        M = st_m[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) M = fmaxf(M, st_m[i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float f = st_m[i] > -INFINITY ? __expf(st_m[i] - M) : 0.f;
            num += f * st_acc[i][tid];
            den += f * st_l[i];
        }
    }
    if (SPLIT && S > 1) {
        float* wsb = split_ws + ((long)(row * n_heads + head) * S) * 66;
        if (tid < 64) {
            wsb[sp * 66 + 2 + tid] = num;
            if (tid == 0) {
                wsb[sp * 66] = M;
                wsb[sp * 66 + 1] = den;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int ticket = __hip_atomic_fetch_add(&split_cnt[row * n_heads + head], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last[0] = ticket == S - 1;
            if (s_last[0]) {
                __hip_atomic_store(&split_cnt[row * n_heads + head], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
        __syncthreads();
        if (!s_last[0]) return;
        if (tid < 64) {  // merge in split order (fixed: the result does not depend on who arrived last)
            // the per-split scalars through agent-scope loads (a uniform plain load may take the scalar cache, which the acquire does not cover)
            M = -INFINITY;
            for (int i = 0; i < S; ++i) M = fmaxf(M, __hip_atomic_load(&wsb[i * 66], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            num = den = 0.f;
            for (int i = 0; i < S; ++i) {
                const float mi = __hip_atomic_load(&wsb[i * 66], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float f = mi > -INFINITY ? __expf(mi - M) : 0.f;
                num += f * wsb[i * 66 + 2 + tid];
                den += f * __hip_atomic_load(&wsb[i * 66 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (tid < 64) {
        long oi = (long)row * o_ld + head * 64 + tid;
        if (o_packed) {  // operand layout of the o-projection GEMV (include/cbx.h "packed GEMV weight layout", K = n_heads * 64)
            const int n = head * 64 + tid;
            oi = (((long)(row >> 4) * (n_heads * 2) + (n >> 5)) * 2 + ((n >> 2) & 1)) * 256 + ((((n >> 3) & 3) << 4) + (row & 15)) * 4 + (n & 3);
        }
        cbx_store_out(o + oi, num / den);
    }
#ifdef CBX_TRACE
    CBX_TRC_STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CBX_TRC_STAMP(4);
    CBX_TRC_FLUSH(0x20000000u | (unsigned)(PIPE ? 1 : 0) | (unsigned)(NT ? 2 : 0) | (unsigned)(SPEC ? 4 : 0) | (unsigned)(SPLIT ? 8 : 0));
#endif
}

// ------------------------------------------------------------------------------------------------------------
// Conformer rel-pos softmax over materialised scores: one wave per query row.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_relpos_kernel(const float* __restrict__ ac, const float* __restrict__ bd,
                                                             float* __restrict__ p, const int* __restrict__ key_lens,
                                                             int nz2, int Tq, int Tk, long ld_ac, long ld_bd, long ld_p,
                                                             long zs_ac, long zs_bd, long zs_p, float scale) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int z = blockIdx.y;
    if (i >= Tq) return;
    const int klen = key_lens ? min(Tk, key_lens[z / nz2]) : Tk;
    const float* ar = ac + (long)z * zs_ac + (long)i * ld_ac;
    const float* br = bd ? bd + (long)z * zs_bd + (long)i * ld_bd + (Tk - 1 - i) : nullptr;
    float* pr = p + (long)z * zs_p + (long)i * ld_p;
    float mx = -INFINITY;
    for (int j = lane; j < klen; j += 64) {
        float s = ar[j] + (br ? br[j] : 0.f);
        mx = fmaxf(mx, s * scale);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < klen; j += 64) {
        float s = (ar[j] + (br ? br[j] : 0.f)) * scale;
        sum += __expf(s - mx);
    }
    sum = wave_sum(sum);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    for (int j = lane; j < ld_p; j += 64) {
        float v = 0.f;
        if (j < klen) {
            float s = (ar[j] + (br ? br[j] : 0.f)) * scale;
            v = __expf(s - mx) * inv;
        }
        pr[j] = v;
    }
}

}  // namespace
CBX_TRC_SETTER(cbx_trace_set_attention)

extern "C" int cbx_flash_attn_f32(const float* q, const float* k, const float* v, float* o, const int* key_lens,
                                  int nz1, int n_heads, int Tq, int Tk, long q_sb, long q_st, long k_sb, long k_st,
                                  long v_sb, long v_st, long o_sb, long o_st, float scale, int causal, void* stream) {
    CBX_REQUIRE(q && k && v && o, "flash_attn: null operand");
    CBX_REQUIRE(Tq > 0 && Tk > 0 && nz1 > 0 && n_heads > 0, "flash_attn: bad shape");
    CBX_REQUIRE((q_st | k_st | v_st | o_st | q_sb | k_sb | v_sb | o_sb) % 4 == 0, "flash_attn: strides must be multiples of 4");
    CBX_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) == 0, "flash_attn: 16-byte alignment");
    FlashArgs a{q, k, v, o, key_lens, Tq, Tk, q_sb, q_st, k_sb, k_st, v_sb, v_st, o_sb, o_st, scale, causal};
    dim3 grid((Tq + 127) / 128, n_heads, nz1);
    hipLaunchKernelGGL(flash_attn_f32_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);  // (<false>: without the K / V register prefetch; measured slower in round 2)
    return cbx_check_launch("flash_attn");
}

// ABI v15: K / V with their own head strides -- the prefill of T3's text positions reads the keys of the (cached) conditioning prefix and its own from the KV cache
// ([row][head][max_ctx][64]: k_st = 64, k_sh = max_ctx * 64), queries from the q | k | v workspace; causal with Tq < Tk: query i sees keys j <= i + (Tk - Tq).
extern "C" int cbx_flash_attn_kv_f32(const float* q, const float* k, const float* v, float* o, const int* key_lens, int nz1, int n_heads, int Tq, int Tk,
                                     long q_sb, long q_st, long k_sb, long k_st, long k_sh, long v_sb, long v_st, long v_sh, long o_sb, long o_st, float scale,
                                     int causal, void* stream) {
    CBX_REQUIRE(q && k && v && o, "flash_attn_kv: null operand");
    CBX_REQUIRE(Tq > 0 && Tk > 0 && nz1 > 0 && n_heads > 0, "flash_attn_kv: bad shape");
    CBX_REQUIRE((q_st | k_st | v_st | o_st | q_sb | k_sb | v_sb | o_sb | k_sh | v_sh) % 4 == 0, "flash_attn_kv: strides must be multiples of 4");
    CBX_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) == 0, "flash_attn_kv: 16-byte alignment");
    FlashArgs a{q, k, v, o, key_lens, Tq, Tk, q_sb, q_st, k_sb, k_st, v_sb, v_st, o_sb, o_st, scale, causal, k_sh, v_sh};
    dim3 grid((Tq + 127) / 128, n_heads, nz1);
    hipLaunchKernelGGL(flash_attn_f32_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    return cbx_check_launch("flash_attn_kv");
}

extern "C" int cbx_flash_relpos_f32(const float* qu, const float* qv, const float* k, const float* v, const float* pp, float* o,
                                    const int* key_lens, int nz1, int n_heads, int T, long q_sb, long q_st, long pp_st, long o_sb, long o_st,
                                    float scale, void* stream) {
    CBX_REQUIRE(qu && qv && k && v && pp && o, "flash_relpos: null operand");
    CBX_REQUIRE(T > 0 && nz1 > 0 && n_heads > 0, "flash_relpos: bad shape");
    CBX_REQUIRE((q_st | q_sb | pp_st | o_st | o_sb) % 4 == 0, "flash_relpos: strides must be multiples of 4");
    CBX_REQUIRE((((uintptr_t)qu | (uintptr_t)qv | (uintptr_t)k | (uintptr_t)v | (uintptr_t)pp | (uintptr_t)o) & 15) == 0, "flash_relpos: 16-byte alignment");
    static unsigned long long configured = 0;  // one bit per device ordinal: hipFuncSetAttribute is per device
    const int dev = cbx_device();
    if (!(configured >> dev & 1)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(flash_relpos_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RELPOS_LDS);
        if (e != hipSuccess) return cbx_set_error((int)e, "flash_relpos: cannot reserve %d B of LDS", RELPOS_LDS);
        configured |= 1ull << dev;
    }
    FlashRelArgs a{qu, qv, k, v, pp, o, key_lens, T, 2 * T - 1, q_sb, q_st, pp_st, o_sb, o_st, scale};
    hipLaunchKernelGGL(flash_relpos_f32_kernel, dim3((T + 127) / 128, n_heads, nz1), dim3(256), RELPOS_LDS, (hipStream_t)stream, a);
    return cbx_check_launch("flash_relpos");
}

extern "C" int cbx_decode_attn_f32(const float* q, const float* kc, const float* vc, float* o, const int* ctx_lens,
                                   int rows, int n_heads, long q_ld, long o_ld, long cache_row_stride,
                                   long cache_head_stride, float scale, void* stream) {
    CBX_REQUIRE(q && kc && vc && o && ctx_lens, "decode_attn: null operand");
    CBX_REQUIRE(q_ld % 4 == 0 && cache_row_stride % 4 == 0 && cache_head_stride % 4 == 0, "decode_attn: alignment");
    hipLaunchKernelGGL(decode_attn_f32_kernel, dim3(n_heads, rows), dim3(256), 0, (hipStream_t)stream, q, kc, vc, o,
                       ctx_lens, q_ld, o_ld, cache_row_stride, cache_head_stride, scale);
    return cbx_check_launch("decode_attn");
}

// ---- process-wide TEST HOOKS of the positional entry point cbx_decode_attn_rope_f32 (tests / A-B scripts).  The engines do not use them: their
// geometry and their split-context workspace travel per call in cbx_decode_attn_t (ABI v10), so two engines in one process -- or a hipGraph
// captured earlier -- never see each other's settings.
static int g_da_u = 0;  // 0: the default, 4
extern "C" int cbx_set_decode_attn_unroll(int u) {
    CBX_REQUIRE(u == 4 || u == 8 || u == 16, "decode_attn unroll must be 4, 8 or 16");
    g_da_u = u;
    return 0;
}
static float* g_da_ws[64] = {nullptr};
static int* g_da_cnt[64] = {nullptr};
static long g_da_pairs[64] = {0};
extern "C" int cbx_set_decode_attn_workspace(float* ws, int* zeroed_counters, long max_pairs) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return cbx_set_error(CBX_EINVAL, "set_decode_attn_workspace: no current device");
    g_da_ws[d] = ws;
    g_da_cnt[d] = zeroed_counters;
    g_da_pairs[d] = ws && zeroed_counters ? max_pairs : 0;
    return 0;
}
static int g_da_pipe = 0;
extern "C" int cbx_set_decode_attn_pipeline(int on) {  // bit 0: pipelined K / V stream; bit 1: non-temporal K / V loads; bit 2: speculative first step
    g_da_pipe = on & 7;
    return 0;
}
constexpr int DA_MAX_SPLIT = 8;
static int g_da_split_min = 512;
extern "C" int cbx_set_decode_attn_split_min(int min_ctx) {
    CBX_REQUIRE(min_ctx >= 1, "decode_attn split threshold must be >= 1");
    g_da_split_min = min_ctx;
    return 0;
}

extern "C" int cbx_decode_attn_rope(const cbx_decode_attn_t* pd, void* stream) {
    CBX_REQUIRE(pd, "decode_attn_rope: null descriptor");
    const cbx_decode_attn_t& a = *pd;
    CBX_REQUIRE(a.qkv && a.positions && a.kc && a.vc && a.o && (!a.cos_t == !a.sin_t), "decode_attn_rope: null operand");
    CBX_REQUIRE(a.ld_qkv % 4 == 0 && a.cache_row_stride % 4 == 0 && a.cache_head_stride % 4 == 0, "decode_attn_rope: alignment");
    CBX_REQUIRE(a.unroll == 0 || a.unroll == 4 || a.unroll == 8 || a.unroll == 16, "decode_attn_rope: unroll must be 0 (= 4), 4, 8 or 16");
    CBX_REQUIRE(a.pipeline >= 0 && a.pipeline <= 7 && a.split_min >= 0, "decode_attn_rope: pipeline in 0 .. 7, split_min >= 0");
    CBX_REQUIRE(!(a.pipeline & 4) || a.cache_head_stride >= 64 * 64, "decode_attn_rope: the speculative first step needs 64 cache positions per (row, head)");
    CBX_REQUIRE(a.qkv_nparts >= 0 && a.qkv_nparts <= 4 && (a.qkv_nparts <= 1 || (a.qkv_ssq && a.rms_dim > 0 && a.rows <= 16 && a.qkv_part_stride % 4 == 0)),
                "decode_attn_rope: qkv_nparts in 0 .. 4; partial sums need qkv_ssq, rms_dim and rows <= 16");
    const int nparts = a.qkv_nparts > 1 ? a.qkv_nparts : 1;
    const float inv_dim = a.rms_dim > 0 ? (float)a.rms_dim : 1.f;  // (passed as the dimension itself)
    // contexts shorter than this are walked by one workgroup even on a split grid (the hand-off costs more than it saves below ~3 round trips)
    const int split_min = a.split_min > 0 ? a.split_min : 512;
    const long pairs = (long)a.rows * a.n_heads;
    int S = 1;
    // the split-context form needs a CALLER-OWNED workspace (66 * 8 floats + one zeroed int per (row, head)): launches that may be in flight
    // at the same time (two engines, two streams) must not share one
    if (pairs < 128 && a.split_ws && a.split_cnt && a.split_pairs >= pairs) {
        S = (int)(256 / pairs);  // fill the chip: Turbo / Nano at batch 1 = 12-16 pairs -> 8 workgroups each
        S = S > DA_MAX_SPLIT ? DA_MAX_SPLIT : S < 1 ? 1 : S;
    }
    const dim3 grid(a.n_heads, a.rows, S), block(256);
    hipStream_t st = (hipStream_t)stream;
    float* ws = S > 1 ? a.split_ws : nullptr;
    int* cnt = S > 1 ? a.split_cnt : nullptr;
    // One workgroup per (row, head[, split]) walks its context, 4 key rows per 16-lane group and step: best on the batched Llama grid
    // (16 rows x 16 heads, profiles/r02_t3_decode_variants.log) AND on the small grids of batch 1 -- a same-box A/B of Multilingual /
    // Nano / Turbo at batch 1 (profiles/r03_decode_attn_b1_ab.log) has 4 rows + no split ahead of 16 rows in flight and of a split
    // at contexts of 200-700 by 1-4 %; the split engages on contexts >= split_min (512): a 1000-token Turbo generation
    // (contexts to ~1450) decodes at 1.03 ms / token with it, 1.20 without (profiles/r03_turbo_long_context.log).
    const int da_u = a.unroll > 0 ? a.unroll : 4;
#define CBX_DA_LAUNCH(U, P, N, SP)                                                                                                         \
    do {                                                                                                                                   \
        if (S > 1)                                                                                                                         \
            hipLaunchKernelGGL((decode_attn_rope_kernel<U, true, P, N, SP>), grid, block, 0, st, a.qkv, a.positions, a.cos_t, a.sin_t, a.kc, a.vc, a.o, a.n_heads, \
                               a.ld_qkv, a.o_ld, a.o_packed, a.cache_row_stride, a.cache_head_stride, a.scale, ws, cnt, split_min, nparts,  \
                               a.qkv_part_stride, a.qkv_ssq, inv_dim, a.rms_eps);                                                          \
        else                                                                                                                               \
            hipLaunchKernelGGL((decode_attn_rope_kernel<U, false, P, N, SP>), grid, block, 0, st, a.qkv, a.positions, a.cos_t, a.sin_t, a.kc, a.vc, a.o, a.n_heads, \
                               a.ld_qkv, a.o_ld, a.o_packed, a.cache_row_stride, a.cache_head_stride, a.scale, ws, cnt, split_min, nparts,  \
                               a.qkv_part_stride, a.qkv_ssq, inv_dim, a.rms_eps);                                                          \
    } while (0)
    const int pipe = a.pipeline;
    if (pipe & 4) {  // speculative first step (4 rows per lane group and step), with or without the pipelined stream / non-temporal loads
        switch (pipe & 3) {
            case 0: CBX_DA_LAUNCH(4, false, false, true); break;
            case 1: CBX_DA_LAUNCH(4, true, false, true); break;
            case 2: CBX_DA_LAUNCH(4, false, true, true); break;
            default: CBX_DA_LAUNCH(4, true, true, true); break;
        }
    } else if (pipe & 2) {  // non-temporal K / V loads (4 rows per lane group and step), plain or pipelined
        if (pipe & 1) CBX_DA_LAUNCH(4, true, true, false);
        else CBX_DA_LAUNCH(4, false, true, false);
    } else if (pipe) {  // two register sets, the next step's rows requested before this step's arithmetic (4 or 8 rows per lane group and step)
        if (da_u == 8) CBX_DA_LAUNCH(8, true, false, false);
        else CBX_DA_LAUNCH(4, true, false, false);
    } else if (da_u == 8) CBX_DA_LAUNCH(8, false, false, false);
    else if (da_u == 16) CBX_DA_LAUNCH(16, false, false, false);
    else CBX_DA_LAUNCH(4, false, false, false);
#undef CBX_DA_LAUNCH
    return cbx_check_launch("decode_attn_rope");
}

// positional form (ABI <= 9): geometry from the process-wide test hooks above, workspace from cbx_set_decode_attn_workspace
extern "C" int cbx_decode_attn_rope_f32(const float* qkv, const int* positions, const float* cos_t, const float* sin_t, float* kc,
                                        float* vc, float* o, int rows, int n_heads, long ld_qkv, long o_ld, int o_packed,
                                        long cache_row_stride, long cache_head_stride, float scale, void* stream) {
    cbx_decode_attn_t a{};
    a.qkv = qkv, a.positions = positions, a.cos_t = cos_t, a.sin_t = sin_t, a.kc = kc, a.vc = vc, a.o = o;
    a.rows = rows, a.n_heads = n_heads, a.ld_qkv = ld_qkv, a.o_ld = o_ld, a.o_packed = o_packed;
    a.cache_row_stride = cache_row_stride, a.cache_head_stride = cache_head_stride, a.scale = scale;
    a.unroll = g_da_u > 0 ? g_da_u : 0, a.pipeline = g_da_pipe, a.split_min = g_da_split_min;
    int d = 0;
    if (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64) a.split_ws = g_da_ws[d], a.split_cnt = g_da_cnt[d], a.split_pairs = g_da_pairs[d];
    return cbx_decode_attn_rope(&a, stream);
}

extern "C" int cbx_softmax_relpos_f32(const float* ac, const float* bd, float* p, const int* key_lens, int nz1, int nz2,
                                      int Tq, int Tk, long ld_ac, long ld_bd, long ld_p, long zs_ac, long zs_bd,
                                      long zs_p, float scale, void* stream) {
    CBX_REQUIRE(ac && p && Tq > 0 && Tk > 0, "softmax_relpos: bad args");
    hipLaunchKernelGGL(softmax_relpos_kernel, dim3((Tq + 3) / 4, nz1 * nz2), dim3(256), 0, (hipStream_t)stream, ac, bd, p,
                       key_lens, nz2, Tq, Tk, ld_ac, ld_bd, ld_p, zs_ac, zs_bd, zs_p, scale);
    return cbx_check_launch("softmax_relpos");
}
