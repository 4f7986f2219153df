// Flash attention (head_dim 64) on the bf16 matrix cores with split fp32 operands ("bf16x3" / "bf16x6", see
// gemm_split.hip for the arithmetic): same swapped formulation, tiling and masking semantics as flash_attn_f32_kernel
// (attention.hip), 5.3x / 2.7x fewer matrix-core cycles per KV tile.  F16 = the f16x3 form (two fp16 planes, the second scaled by
// 2048 and its products kept in a second accumulator; fp32-level at the bf16x3 cost, fp16 operand range -> cbx_set_range_flag).
//
//   S^T = K Q^T : A = K tile from LDS planes [key][d]   (lane: key = lane&31, d = 16kc + 8*(lane>>5) .. +8)
//                 B = Q^T held in registers as planes   (lane: query = lane&31, same d)
//   O^T = V^T P^T: A = V^T from LDS planes [d][key']    (lane: d = lane&31, 8 keys of chunk c)
//                  B = P^T straight from the S accumulators: registers 8u..8u+7 of sub-tile t are keys
//                      32t + 16u + 8(j>>2) + 4*(lane>>5) + (j&3), j = 0..7, so the V^T image stores key k of a 16-key chunk
//                      at position (k with bits 2 and 3 swapped): the lane's 8 keys are then one ds_read_b128.
// The MFMA sums all 16 k of both half-waves, so any k <-> (half, j) assignment is legal as long as A and B agree.
#include "cbx_common.h"

namespace {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int FKT = 64;   // keys per tile
constexpr int FLD = 72;   // LDS row stride in bf16 (144 B: odd multiple of 16 B -> conflict-free ds_read_b128)
constexpr int FPLANE = 64 * FLD;

struct FlashSplitArgs {
    const float* q; const float* k; const float* v; float* o; const int* key_lens;
    int Tq, Tk;
    long q_sb, q_st, k_sb, k_st, v_sb, v_st, o_sb, o_st;
    float scale;
    int causal;
    int o_planes;  // o is a PLANE-format tensor (two fp16 planes, gemm_planes.hip): o_sb / o_st in halves, l plane o_lo halves after h
    long o_lo;
};

template <int NP, bool F16>
__device__ __forceinline__ void split8(const f32x8 v, bf16x8 (&out)[NP]) {
    if constexpr (F16) {
        const f16x8 h = __builtin_convertvector(v, f16x8);
        const f32x8 t = v * CBX_F16_LO_SCALE;
        f32x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = __builtin_fmaf((float)h[e], -CBX_F16_LO_SCALE, t[e]);  // 2048 (v - h), exact
        out[0] = __builtin_bit_cast(bf16x8, h);
        out[1] = __builtin_bit_cast(bf16x8, __builtin_convertvector(r, f16x8));
        return;
    }
    out[0] = __builtin_convertvector(v, bf16x8);
    f32x8 r = v - __builtin_convertvector(out[0], f32x8);
    out[1] = __builtin_convertvector(r, bf16x8);
    if constexpr (NP == 3) {
        r = r - __builtin_convertvector(out[1], f32x8);
        out[2] = __builtin_convertvector(r, bf16x8);
    }
}

// acc += sum over the plane products above the fp32 rounding level (smallest first)
// F16: acc += h*h, accc += h*l + l*h (2048 times too large)
__device__ __forceinline__ void mma_f16(const bf16x8 (&a)[2], const bf16x8 (&b)[2], f32x16& acc, f32x16& accc) {
    const f16x8 ah = __builtin_bit_cast(f16x8, a[0]), al = __builtin_bit_cast(f16x8, a[1]);
    const f16x8 bh = __builtin_bit_cast(f16x8, b[0]), bl = __builtin_bit_cast(f16x8, b[1]);
    accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accc, 0, 0, 0);
    accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
}

template <int NP>
__device__ __forceinline__ f32x16 mma_split(const bf16x8 (&a)[NP], const bf16x8 (&b)[NP], f32x16 acc) {
    if constexpr (NP == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    return acc;
}

// VALU diet of the KV loop (the loop was VALU-bound: 735 vector instructions against 48 MFMAs per tile and wave):
//   * K / V rows come through raw buffer loads whose offset is pushed past num_records for keys >= klen: the hardware returns zeros, so
//     neither the fetch nor the staging selects anything;
//   * tiles that no lane has to mask (all but the last one without causality) take a softmax path without compares / selects;
//   * scores are kept in the log2 domain (log2 e folded into the Q scale): p = exp2(s - m) is one subtract and one v_exp_f32;
//   * O is only rescaled when some lane's running maximum moved.
template <int NP, bool F16 = false>
__global__ __launch_bounds__(256, 2) void flash_attn_split_kernel(const FlashSplitArgs a, int* range_flag) {
    static_assert(!F16 || NP == 2, "the fp16 form has two planes");
    float amax = 0.f;  // F16: largest |operand| (fp16 range check at the end)
    // planes: K [NP][64 keys][FLD], then V^T [NP][64 d][FLD]
    __shared__ __attribute__((aligned(16))) __bf16 Ks[NP * FPLANE];
    __shared__ __attribute__((aligned(16))) __bf16 Vt[NP * FPLANE];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int tile = cbx_xcd_remap((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gridDim.x * gridDim.y * gridDim.z);
    const int qt = tile % gridDim.x, head = (tile / gridDim.x) % gridDim.y, z = tile / (gridDim.x * gridDim.y);
    const int q0 = qt * 128;
    const int qi = q0 + wid * 32 + lr;  // this lane's query
    const float* qb = a.q + (long)z * a.q_sb + head * 64;
    const float* kb = a.k + (long)z * a.k_sb + head * 64;
    const float* vb = a.v + (long)z * a.v_sb + head * 64;
    const int klen = a.key_lens ? min(a.Tk, a.key_lens[z]) : a.Tk;
    const int coff = a.Tk - a.Tq;

    // Q planes: chunk kc covers d = 16kc + 8lh .. +8, pre-scaled by scale * log2(e)
    bf16x8 qf[4][NP];
    {
        const bool ok = qi < a.Tq;
        const float* qp = qb + (long)(ok ? qi : 0) * a.q_st + 8 * lh;
        const float sc = ok ? a.scale * 1.4426950408889634f : 0.f;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(qp + 16 * kc) * sc;
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(qp + 16 * kc + 4) * sc;
            const f32x8 t = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
            if constexpr (F16) {
                cbx_amax4(amax, t0);
                cbx_amax4(amax, t1);
            }
            split8<NP, F16>(t, qf[kc]);
        }
    }

    f32x16 ot[2], otc[2];  // otc: cross-product accumulator of the F16 form
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[d][r] = otc[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    int kend = klen;
    if (a.causal) kend = min(kend, q0 + 128 + coff);

    // staging maps.  K: thread = (key row, 16 consecutive d).  V: thread = (group of 4 consecutive keys, 4 consecutive d)
    // so the transposed image is written 4 keys (8 B) at a time; the (dq, g) <-> tid map keeps those stores conflict-free.
    const int k_row = tid >> 2, k_c = (tid & 3) * 16;
    const int v_dq = ((tid >> 4) & 7) * 2 + (tid & 1);      // d quad 0..15
    const int v_g = (tid >> 7) * 8 + ((tid >> 1) & 7);      // key group 0..15 (keys 4g..4g+3)
    // position of key 4g inside the V^T row: bits 2 and 3 of the key index swapped
    const int v_pos = (v_g & ~3) * 4 + (v_g & 1) * 8 + ((v_g >> 1) & 1) * 4;

    constexpr int OOB = (int)0x80000000;  // byte offset >= num_records: the buffer load returns 0 (host checks that real offsets fit 31 bits)
    const __amdgpu_buffer_rsrc_t k_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(kb), 0, OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vb), 0, OOB, 0x00020000);
    const int k_st4 = (int)a.k_st * 4, v_st4 = (int)a.v_st * 4;
    const int k_lin = k_row * k_st4 + k_c * 4, v_lin = 4 * v_g * v_st4 + 16 * v_dq;  // byte offsets inside tile 0

    f32x4 kreg[4], vreg[4];
    auto fetch = [&](int j0) {
        const int kvo = j0 + k_row < klen ? k_lin : OOB;
        const int ks = j0 * k_st4, vs = j0 * v_st4;  // scalar part of the address
#pragma unroll
        for (int c = 0; c < 4; ++c)
            kreg[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(k_rs, kvo + 16 * c, ks, 0));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int vvo = j0 + 4 * v_g + i < klen ? v_lin + i * v_st4 : OOB;
            vreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(v_rs, vvo, vs, 0));
        }
    };
    auto stage = [&]() {
        // K planes: 16 d of one key -> two 16-B stores per plane
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const f32x4 x0 = kreg[2 * c2], x1 = kreg[2 * c2 + 1];
            const f32x8 x = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
            bf16x8 pl[NP];
            if constexpr (F16) {
                cbx_amax4(amax, x0);
                cbx_amax4(amax, x1);
            }
            split8<NP, F16>(x, pl);
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<bf16x8*>(&Ks[q * FPLANE + k_row * FLD + k_c + 8 * c2]) = pl[q];
        }
        // V^T planes: for each of the thread's 4 d, the 4 keys of its group are contiguous (8 B)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x4 col = {vreg[0][e], vreg[1][e], vreg[2][e], vreg[3][e]};  // keys 4g..4g+3 at d = 4dq + e
            __bf16* dst = &Vt[(4 * v_dq + e) * FLD + v_pos];
            if constexpr (F16) {
                const f16x4 hh = __builtin_convertvector(col, f16x4);
                const f32x4 tt = col * CBX_F16_LO_SCALE;
                f32x4 rr;
#pragma unroll
                for (int q = 0; q < 4; ++q) rr[q] = __builtin_fmaf((float)hh[q], -CBX_F16_LO_SCALE, tt[q]);
                *reinterpret_cast<f16x4*>(dst) = hh;
                *reinterpret_cast<f16x4*>(dst + FPLANE) = __builtin_convertvector(rr, f16x4);
                cbx_amax4(amax, col);
                continue;
            }
            bf16x4 h = __builtin_convertvector(col, bf16x4);
            f32x4 r = col - __builtin_convertvector(h, f32x4);
            bf16x4 m = __builtin_convertvector(r, bf16x4);
            *reinterpret_cast<bf16x4*>(dst) = h;
            *reinterpret_cast<bf16x4*>(dst + FPLANE) = m;
            if constexpr (NP == 3) {
                r = r - __builtin_convertvector(m, f32x4);
                *reinterpret_cast<bf16x4*>(dst + 2 * FPLANE) = __builtin_convertvector(r, bf16x4);
            }
        }
    };
    if (kend > 0) fetch(0);

    const int q_lo = q0 + wid * 32 + coff;  // causal: the wave's first query sees keys <= q_lo
    const int jmax = a.causal ? min(klen - 1, qi + coff) : klen - 1;  // last key this lane's query sees
    for (int j0 = 0; j0 < kend; j0 += FKT) {
        stage();
        __syncthreads();
        // unconditional prefetch (rows past the end return zeros): keeps hipcc's vmcnt exact
        fetch(j0 + FKT);

        // ---- S^T = K Q^T  (2 sub-tiles of 32 keys, 4 d-chunks of 16), in units of log2
        f32x16 st[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 stc;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[t][r] = stc[r] = 0.f;
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                bf16x8 kf[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    kf[q] = *reinterpret_cast<const bf16x8*>(&Ks[q * FPLANE + (t * 32 + lr) * FLD + 16 * kc + 8 * lh]);
                if constexpr (F16) mma_f16(kf, qf[kc], st[t], stc);
                else st[t] = mma_split<NP>(kf, qf[kc], st[t]);
            }
            if constexpr (F16) st[t] += stc * (1.0f / CBX_F16_LO_SCALE);
        }

        // ---- online softmax (lane owns query qi; registers hold keys row(r) + 4*lh of each sub-tile)
        const bool full = j0 + FKT <= klen && (!a.causal || j0 + FKT - 1 <= q_lo);  // wave-uniform: nothing to mask in this tile
        float mt = -INFINITY;
        if (full) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; r += 2) mt = fmaxf(fmaxf(mt, st[t][r]), st[t][r + 1]);
        } else {
            const int jl = jmax - j0 - 4 * lh;  // key offset inside the tile (without the lane's 4*lh) up to which this query sees
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sv = t * 32 + (r & 3) + 8 * (r >> 2) <= jl ? st[t][r] : -INFINITY;
                    st[t][r] = sv;
                    mt = fmaxf(mt, sv);
                }
        }
        mt = fmaxf(mt, cbx_xor_lane<32>(mt));
        const float m_new = fmaxf(m_run, mt);
        const float m_sub = m_new > -INFINITY ? m_new : 0.f;  // a row that has seen no key yet: exp2(-inf - 0) = 0
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_sub);
        float ls = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(st[t][r] - m_sub);
                st[t][r] = pv;
                ls += pv;
            }
        l_run = l_run * alpha + ls;
        m_run = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                ot[d] *= alpha;
                if constexpr (F16) otc[d] *= alpha;
            }
        }

        // ---- O^T += V^T P^T : chunk c = 2t + u contracts the 16 keys held in registers 8u..8u+7 of both half-waves
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x8 pv = {st[t][8 * u + 0], st[t][8 * u + 1], st[t][8 * u + 2], st[t][8 * u + 3],
                                  st[t][8 * u + 4], st[t][8 * u + 5], st[t][8 * u + 6], st[t][8 * u + 7]};
                bf16x8 pf[NP];
                split8<NP, F16>(pv, pf);
                const int c = 2 * t + u;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    bf16x8 vf[NP];
#pragma unroll
                    for (int q = 0; q < NP; ++q)
                        vf[q] = *reinterpret_cast<const bf16x8*>(&Vt[q * FPLANE + (dt * 32 + lr) * FLD + 16 * c + 8 * lh]);
                    if constexpr (F16) mma_f16(vf, pf, ot[dt], otc[dt]);
                    else ot[dt] = mma_split<NP>(vf, pf, ot[dt]);
                }
            }
        __syncthreads();
    }

    if constexpr (F16) {
        if (amax > 65504.f && range_flag) atomicOr(range_flag, 1);
#pragma unroll
        for (int d = 0; d < 2; ++d) ot[d] += otc[d] * (1.0f / CBX_F16_LO_SCALE);
    }
    // ---- finalise: both half-waves hold partial sums of the same query
    const float l_tot = l_run + cbx_xor_lane<32>(l_run);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (qi < a.Tq && a.o_planes) {  // attention output feeds the to_out projection only: written as the two fp16 planes that GEMM consumes
        _Float16* op = reinterpret_cast<_Float16*>(a.o) + (long)z * a.o_sb + (long)qi * a.o_st + head * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 t = {ot[d][g * 4 + 0] * inv, ot[d][g * 4 + 1] * inv, ot[d][g * 4 + 2] * inv, ot[d][g * 4 + 3] * inv};
                const f16x4 h = __builtin_convertvector(t, f16x4);
                const f32x4 t2 = t * CBX_F16_LO_SCALE;
                f32x4 dl;
#pragma unroll
                for (int e = 0; e < 4; ++e) dl[e] = __builtin_fmaf((float)h[e], -CBX_F16_LO_SCALE, t2[e]);
                *reinterpret_cast<f16x4*>(op + d * 32 + 8 * g + 4 * lh) = h;
                *reinterpret_cast<f16x4*>(op + a.o_lo + d * 32 + 8 * g + 4 * lh) = __builtin_convertvector(dl, f16x4);
            }
    } else if (qi < a.Tq) {
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

}  // namespace

static int flash_split_launch(const float* q, const float* k, const float* v, void* o, const int* key_lens, int nz1, int n_heads, int Tq,
                              int Tk, long q_sb, long q_st, long k_sb, long k_st, long v_sb, long v_st, long o_sb, long o_st, float scale,
                              int causal, int precision, int o_planes, long o_lo, void* stream) {
    CBX_REQUIRE(q && k && v && o, "flash_attn_split: null operand");
    CBX_REQUIRE(Tq > 0 && Tk > 0 && nz1 > 0 && n_heads > 0, "flash_attn_split: bad shape");
    CBX_REQUIRE(precision == 3 || precision == 6 || precision == 16, "flash_attn_split: precision must be 3, 6 or 16 (got %d)", precision);
    CBX_REQUIRE((q_st | k_st | v_st | o_st | q_sb | k_sb | v_sb | o_sb | o_lo) % 4 == 0, "flash_attn_split: strides must be multiples of 4");
    CBX_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0 && ((uintptr_t)o & (o_planes ? 7 : 15)) == 0, "flash_attn_split: alignment");
    CBX_REQUIRE((long)(Tk + 64) * k_st * 4 < 0x7fffffffL && (long)(Tk + 64) * v_st * 4 < 0x7fffffffL && k_st > 0 && v_st > 0,
                "flash_attn_split: one (batch, head) slice of K / V must span less than 2 GiB");
    CBX_REQUIRE(!o_planes || precision == 16, "flash_attn_split: plane-format output is an f16x3 (precision 16) feature");
    FlashSplitArgs a{q, k, v, reinterpret_cast<float*>(o), key_lens, Tq, Tk, q_sb, q_st, k_sb, k_st, v_sb, v_st, o_sb, o_st, scale, causal, o_planes, o_lo};
    dim3 grid((Tq + 127) / 128, n_heads, nz1);
    int* flag = cbx_range_flag();
    if (precision == 16) {
        hipLaunchKernelGGL((flash_attn_split_kernel<2, true>), grid, dim3(256), 0, (hipStream_t)stream, a, flag);
    } else if (precision == 3) {
        hipLaunchKernelGGL((flash_attn_split_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, a, flag);
    } else {
        hipLaunchKernelGGL((flash_attn_split_kernel<3>), grid, dim3(256), 0, (hipStream_t)stream, a, flag);
    }
    return cbx_check_launch("flash_attn_split");
}

extern "C" int cbx_flash_attn_split_f32(const float* q, const float* k, const float* v, float* o, const int* key_lens,
                                        int nz1, int n_heads, int Tq, int Tk, long q_sb, long q_st, long k_sb, long k_st,
                                        long v_sb, long v_st, long o_sb, long o_st, float scale, int causal, int precision,
                                        void* stream) {
    return flash_split_launch(q, k, v, o, key_lens, nz1, n_heads, Tq, Tk, q_sb, q_st, k_sb, k_st, v_sb, v_st, o_sb, o_st, scale, causal,
                              precision, 0, 0, stream);
}

extern "C" int cbx_flash_attn_split_po(const float* q, const float* k, const float* v, void* o_planes, const int* key_lens,
                                       int nz1, int n_heads, int Tq, int Tk, long q_sb, long q_st, long k_sb, long k_st,
                                       long v_sb, long v_st, long o_sb, long o_st, long o_lo, float scale, int causal, void* stream) {
    return flash_split_launch(q, k, v, o_planes, key_lens, nz1, n_heads, Tq, Tk, q_sb, q_st, k_sb, k_st, v_sb, v_st, o_sb, o_st, scale,
                              causal, 16, 1, o_lo, stream);
}
