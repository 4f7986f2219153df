// Few-row weight-streaming GEMV for autoregressive decode at batch 1 .. 4 of the GPT-2 backbones (Turbo / Nano, configs[1]):
//   out[m][n] = epi( x'[m] . W[n][:] ),  m < M <= 4.
//
// Roofline: HBM.  With 1 - 4 activation rows there is nothing for a matrix core to amortise (the 16 x 16 x 4 MFMA form of gemv_decode.hip
// spends 12 - 15 of its 16 rows on padding, a 16-row operand image and an LDS reduction over the K slices of 8-16 waves), so this kernel is the
// plain streaming form: W stays in the checkpoint's row-major [N][K] layout (a row IS 3 - 16 KiB of contiguous memory), a WAVE owns R output
// columns over the WHOLE of K, lane l holds k = 256 j + 4 l .. + 3 of every activation row in registers, every wave-level load is 1 KiB
// contiguous, ALL of a wave's weight loads (<= 32 of them, <= 32 KiB) are requested before anything else happens, and the K reduction is
// R M butterflies inside the wave (DPP / permlane: cbx_xor_lane): no LDS, no barrier, no partial images, one memory round trip per launch.
// Prologues (what the x operand is), all in the consuming launch so that a layer stays five launches:
//   PLAIN  x as stored;
//   LN     LayerNorm(x) (HF GPT2Block ln_1 / ln_2 / ln_f): every wave normalises its own register copy of x (two-pass variance, the expression of
//          F.layer_norm) -- L2 reads instead of a launch;
//   ATTN   the merge of the split-context partial results {m, l, 64 numerators} that cbx_decode_attn_parts leaves per (row, head, slice): the 256
//          threads of the workgroup merge the heads in fixed slice order into LDS (one barrier) while the weight loads are in flight.
// Epilogue: + bias, activation, + residual (may alias out: the lane that writes an element is the one that read it).
// Deterministic: per column 4 K/256 fmaf in lane order per row, then the xor butterfly 32, 16, .. 1.
// Replaces F.linear / HF Conv1D at q_len == 1 inside T3.inference_turbo's loop (reference models/t3/t3.py:435-460).
// (Round 6 also built LlamaRMSNorm + SwiGLU forms of this kernel and ran the Llama T3 at batch 1 / 2 on them: 1.022 against 0.960 ms / token at 2 rows,
// 1.367 against 0.966 at 4 -- the 16-row MFMA step wins there, the forms are gone: profiles/r06_k_few_row_path_small_batches_ab.log.)
#include "cbx_common.h"

namespace {

constexpr int PRO_PLAIN = 0, PRO_LN = 1, PRO_ATTN = 2;
constexpr int PARTS_MAXS = 16;  // slices per (row, head) the merge prologue is unrolled for (cbx_decode_attn_parts: n_splits <= 16)

// MR = activation rows held in registers per pass; MP = true: M may exceed MR and the rows are walked in passes of MR (K >= 3072 with 3 - 4 rows: the weights
// stay in registers across passes); MP = false (every other case): one pass, the straight-line code of the single-row kernel.
template <int R, int KB, int PRO, int MR, bool MP>
__global__ __launch_bounds__(256) void gemv_row_kernel(const cbx_gemv_row_t p) {
    __shared__ __attribute__((aligned(16))) float xs[PRO == PRO_ATTN ? MR * 1024 : 4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n0 = (blockIdx.x * 4 + w) * R;
    // ---- every weight load of this wave, first (columns past N re-read column N - 1: loads stay unconditional, results are dropped)
    f32x4 wv[R][KB];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int n = n0 + r < p.N ? n0 + r : p.N - 1;
        const float* wr = p.W + (long)n * p.ldw + lane * 4;
#pragma unroll
        for (int j = 0; j < KB; ++j) wv[r][j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wr + j * 256));
    }
    // epilogue operands of the elements this lane finishes (lane r < R: column n0 + r of every row), requested now
    const int ne = n0 + (lane < R ? lane : 0);
    const bool eok = lane < R && ne < p.N;
    const int nl = eok ? ne : (n0 < p.N ? n0 : p.N - 1);
    const float e_bias = p.bias ? p.bias[nl] : 0.f;
    float e_res[MR];
    if constexpr (!MP) {
#pragma unroll
        for (int m = 0; m < MR; ++m) e_res[m] = (p.res && (MR == 1 || m < p.M)) ? p.res[(long)m * p.ldr + nl] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);

    for (int mb = 0; mb < (MP ? p.M : 1); mb += MR) {
    auto rowok = [&](int m) { return (MR == 1 && !MP) || mb + m < p.M; };  // (M >= 1: the single-row form has nothing to test)
    if constexpr (MP) {
#pragma unroll
        for (int m = 0; m < MR; ++m) e_res[m] = (p.res && mb + m < p.M) ? p.res[(long)(mb + m) * p.ldr + nl] : 0.f;
    }
    f32x4 xv[MR][KB];
    if constexpr (PRO == PRO_ATTN) {
        // thread t: head t / 16, dims 4 (t % 16) .. + 3 of that head.  All slice records of the head are requested before the first use.
        const int head = tid >> 4, kq = tid & 15;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            if (head < p.n_heads && rowok(mr)) {
                const float* rec = p.parts + (long)(mb + mr) * p.parts_row_stride + (long)head * p.n_parts * CBX_ATTN_PART_REC;
                float m[PARTS_MAXS], l[PARTS_MAXS];
                f32x4 nu[PARTS_MAXS];
#pragma unroll
                for (int i = 0; i < PARTS_MAXS; ++i) {
                    const float* ri = rec + (long)(i < p.n_parts ? i : 0) * CBX_ATTN_PART_REC;
                    m[i] = ri[0], l[i] = ri[1];
                    nu[i] = *reinterpret_cast<const f32x4*>(ri + 4 + kq * 4);
                }
                float M = -INFINITY;
#pragma unroll
                for (int i = 0; i < PARTS_MAXS; ++i)
                    if (i < p.n_parts) M = fmaxf(M, m[i]);
                f32x4 num = {0.f, 0.f, 0.f, 0.f};
                float den = 0.f;
#pragma unroll
                for (int i = 0; i < PARTS_MAXS; ++i) {
                    const float f = (i < p.n_parts && m[i] > -INFINITY) ? __expf(m[i] - M) : 0.f;
                    num += nu[i] * f;
                    den += l[i] * f;
                }
                const float inv = 1.0f / den;
                *reinterpret_cast<f32x4*>(&xs[mr * 1024 + head * 64 + kq * 4]) = num * inv;
            }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int j = 0; j < KB; ++j) xv[m][j] = rowok(m) ? *reinterpret_cast<const f32x4*>(&xs[m * 1024 + j * 256 + lane * 4]) : f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const float* xr = p.x + (long)(rowok(m) ? mb + m : 0) * p.ldx + lane * 4;  // rows past M re-read row 0 (their results are never stored)
#pragma unroll
            for (int j = 0; j < KB; ++j) xv[m][j] = *reinterpret_cast<const f32x4*>(xr + j * 256);
        }
        if constexpr (PRO == PRO_LN) {
            f32x4 gv[KB], bv[KB];
#pragma unroll
            for (int j = 0; j < KB; ++j) {
                gv[j] = *reinterpret_cast<const f32x4*>(p.ln_w + j * 256 + lane * 4);
                bv[j] = *reinterpret_cast<const f32x4*>(p.ln_b + j * 256 + lane * 4);
            }
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < KB; ++j) s += (xv[m][j][0] + xv[m][j][1]) + (xv[m][j][2] + xv[m][j][3]);
                const float mean = wave_sum(s) / (float)(KB * 256);
                float q = 0.f;
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    xv[m][j] -= mean;
                    q += (xv[m][j][0] * xv[m][j][0] + xv[m][j][1] * xv[m][j][1]) + (xv[m][j][2] * xv[m][j][2] + xv[m][j][3] * xv[m][j][3]);
                }
                const float rstd = rsqrtf(wave_sum(q) / (float)(KB * 256) + p.eps);
#pragma unroll
                for (int j = 0; j < KB; ++j) xv[m][j] = xv[m][j] * rstd * gv[j] + bv[j];
            }
        }
    }

    float acc[R][MR];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < KB; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s) a = __builtin_fmaf(xv[m][j][s], wv[r][j][s], a);
            acc[r][m] = wave_sum(a);
        }
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        float v = acc[0][m];
#pragma unroll
        for (int r = 1; r < R; ++r) v = lane == r ? acc[r][m] : v;
        if (eok && rowok(m)) {
            v += e_bias;
            if (p.act) v = cbx_act(v, p.act, 0.f, 0.f);
            v += e_res[m];
            p.out[(long)(mb + m) * p.ldo + ne] = v;
        }
    }
    if constexpr (PRO == PRO_ATTN && MP) __syncthreads();  // (a second pass rewrites xs)
    }  // passes over the activation rows
}

template <int R, int KB, int MR, bool MP>
int launch_pro(const cbx_gemv_row_t& p, hipStream_t st) {
    const int waves = (p.N + R - 1) / R;
    const dim3 grid((waves + 3) / 4), block(256);
    if (p.parts) {
        if constexpr (R <= 2 && !MP) hipLaunchKernelGGL((gemv_row_kernel<R, KB, PRO_ATTN, MR, false>), grid, block, 0, st, p);
        else return cbx_set_error(CBX_EINVAL, "gemv_row: the attention-merge prologue serves rows_per_wave <= 2");
    } else if (p.ln_w) hipLaunchKernelGGL((gemv_row_kernel<R, KB, PRO_LN, MR, MP>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemv_row_kernel<R, KB, PRO_PLAIN, MR, MP>), grid, block, 0, st, p);
    return cbx_check_launch("gemv_row");
}

template <int R, int KB>
int launch_m(const cbx_gemv_row_t& p, hipStream_t st) {
    if (p.M == 1) return launch_pro<R, KB, 1, false>(p, st);
    if (p.M == 2) return launch_pro<R, KB, 2, false>(p, st);
    if constexpr (KB <= 4) return launch_pro<R, KB, 4, false>(p, st);
    else return launch_pro<R, KB, 2, true>(p, st);  // K >= 3072: two rows of x per pass, two passes
}

template <int KB>
int launch_r(const cbx_gemv_row_t& p, int R, hipStream_t st) {
    if constexpr (KB <= 4) {
        if (R >= 8) return launch_m<8, KB>(p, st);
        if (R >= 4) return launch_m<4, KB>(p, st);
        if (R == 3) return launch_m<3, KB>(p, st);
    }
    if (R >= 2) return launch_m<2, KB>(p, st);
    return launch_m<1, KB>(p, st);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Decode attention for small (row, head) grids that leaves its split-context PARTIAL results to the consumer (no ticket, no last-arriver
// merge: cbx_gemv_row_f32's ATTN prologue merges them).  grid = (heads, rows, S): workgroup (h, r, sp) owns the 16-position chunks c with
// c % S == sp of row r's context -- an assignment that does not depend on the context length, so the K / V rows of its first CH chunks are
// requested at kernel entry, before positions[] has arrived (positions past the context are masked afterwards; addresses are clamped to the
// cache).  16-lane group g = position g of a chunk, lane l of the group = dims 4 l .. + 3.  The new token's q / k / v come from the qkv
// row, its k / v are appended to the cache by the workgroup that owns its chunk; RoPE optional (GPT-2: none).
// Record of (row, head, sp) = CBX_ATTN_PART_REC floats: {max, sum, -, -, 64 numerators}.
template <int CH>
__global__ __launch_bounds__(256) void decode_attn_parts_kernel(const cbx_attn_parts_t a) {
    __shared__ __attribute__((aligned(16))) float st_acc[16][64];
    __shared__ float st_m[16], st_l[16];
    const int head = blockIdx.x, row = blockIdx.y, sp = blockIdx.z, S = gridDim.z;
    const int tid = threadIdx.x, g = tid >> 4, l16 = tid & 15;
    const float* kb = a.kc + (long)row * a.cache_row_stride + (long)head * a.cache_head_stride;
    const float* vb = a.vc + (long)row * a.cache_row_stride + (long)head * a.cache_head_stride;
    f32x4 kv[CH], vv[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {  // speculative: chunk sp + S u, position g of it
        const int pp = (sp + S * u) * 16 + g;
        const long pc = pp < a.max_ctx ? pp : 0;
        kv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(kb + pc * 64 + l16 * 4));
        vv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vb + pc * 64 + l16 * 4));
    }
    const int pos = a.positions[row];
    const float* qp = a.qkv + (long)row * a.ld_qkv + head * 64 + l16 * 4;
    f32x4 q4 = *reinterpret_cast<const f32x4*>(qp);
    f32x4 kn = *reinterpret_cast<const f32x4*>(qp + (long)a.n_heads * 64);
    const f32x4 vn = *reinterpret_cast<const f32x4*>(qp + (long)a.n_heads * 128);
    __builtin_amdgcn_sched_barrier(0);
    if (a.cos_t) {  // rotate_half RoPE (HF apply_rotary_pos_emb): dim d pairs with d +- 32, i.e. with lane l16 ^ 8 of the group
        const f32x4 c = *reinterpret_cast<const f32x4*>(a.cos_t + (long)pos * 64 + l16 * 4);
        const f32x4 s = *reinterpret_cast<const f32x4*>(a.sin_t + (long)pos * 64 + l16 * 4);
        const float sgn = l16 < 8 ? -1.f : 1.f;
        f32x4 qo, ko;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            qo[e] = cbx_xor_lane<8>(q4[e]);
            ko[e] = cbx_xor_lane<8>(kn[e]);
        }
        q4 = q4 * c + qo * s * sgn;
        kn = kn * c + ko * s * sgn;
    }
    q4 *= a.scale;
    float m = -INFINITY, l = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = sp;; c0 += S * CH) {
        float d[CH];
        float mt = m;
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int pp = (c0 + S * u) * 16 + g;
            if (pp == pos) {  // the new token: from the qkv row, and into the cache
                kv[u] = kn, vv[u] = vn;
                *reinterpret_cast<f32x4*>(a.kc + (kb - a.kc) + (long)pos * 64 + l16 * 4) = kn;
                *reinterpret_cast<f32x4*>(a.vc + (vb - a.vc) + (long)pos * 64 + l16 * 4) = vn;
            }
            float t = kv[u][0] * q4[0] + kv[u][1] * q4[1] + kv[u][2] * q4[2] + kv[u][3] * q4[3];
            t += cbx_xor_lane<8>(t);
            t += cbx_xor_lane<4>(t);
            t += cbx_xor_lane<2>(t);
            t += cbx_xor_lane<1>(t);
            d[u] = pp <= pos ? t : -INFINITY;
            mt = fmaxf(mt, d[u]);
        }
        if (mt > -INFINITY) {
            const float f = __expf(m - mt);
            acc *= f;
            l *= f;
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const float pw = __expf(d[u] - mt);
                l += pw;
                if (pw > 0.f) acc += vv[u] * pw;  // (a masked row may hold anything: never multiplied)
            }
            m = mt;
        }
        const int cn = c0 + S * CH;
        if (cn * 16 > pos) break;  // uniform: the context ends before this workgroup's next chunk
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int pp = (cn + S * u) * 16 + g;
            const long pc = pp < a.max_ctx ? pp : 0;
            kv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(kb + pc * 64 + l16 * 4));
            vv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vb + pc * 64 + l16 * 4));
        }
    }
    // merge the 16 lane groups (fixed order) and leave the record
    *reinterpret_cast<f32x4*>(&st_acc[g][l16 * 4]) = acc;
    if (l16 == 0) st_m[g] = m, st_l[g] = l;
    __syncthreads();
    if (tid < 64) {
        float M = st_m[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) M = fmaxf(M, st_m[i]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float f = st_m[i] > -INFINITY ? __expf(st_m[i] - M) : 0.f;
            num += f * st_acc[i][tid];
            den += f * st_l[i];
        }
        float* rec = a.parts + ((long)(row * a.n_heads + head) * S + sp) * CBX_ATTN_PART_REC;
        rec[4 + tid] = num;
        if (tid == 0) rec[0] = M, rec[1] = den;
    }
}

}  // namespace

extern "C" int cbx_gemv_row_f32(const cbx_gemv_row_t* pp, void* stream) {
    CBX_REQUIRE(pp, "gemv_row: null descriptor");
    cbx_gemv_row_t p = *pp;
    if (p.M <= 0) p.M = 1;
    CBX_REQUIRE(p.W && p.out && (p.x || p.parts), "gemv_row: null operand");
    CBX_REQUIRE(p.M <= 4, "gemv_row: M=%d rows (served: 1 .. 4; more rows belong on cbx_gemv_f32's MFMA tiles)", p.M);
    CBX_REQUIRE(p.N > 0 && p.K > 0 && p.K % 256 == 0 && p.ldw % 4 == 0 && p.ldw >= p.K, "gemv_row: N=%d K=%d ldw=%ld (K %% 256 == 0, ldw %% 4 == 0)", p.N, p.K, p.ldw);
    CBX_REQUIRE(p.M == 1 || ((p.parts || (p.ldx % 4 == 0 && p.ldx >= p.K)) && p.ldo >= p.N && (!p.res || p.ldr >= p.N)), "gemv_row: row strides ldx / ldo / ldr of M > 1 rows");
    CBX_REQUIRE((((uintptr_t)p.W | (uintptr_t)p.x | (uintptr_t)p.ln_w | (uintptr_t)p.ln_b | (uintptr_t)p.parts) & 15) == 0, "gemv_row: 16-byte alignment");
    CBX_REQUIRE(!p.ln_w == !p.ln_b && !(p.ln_w && p.parts), "gemv_row: LayerNorm needs ln_w and ln_b; one prologue per launch");
    CBX_REQUIRE(!p.parts || (p.n_parts >= 1 && p.n_parts <= PARTS_MAXS && p.n_heads >= 1 && p.n_heads <= 16 && p.K == p.n_heads * 64 &&
                             (p.M == 1 || (p.parts_row_stride % 4 == 0 && p.parts_row_stride >= (long)p.n_heads * p.n_parts * CBX_ATTN_PART_REC))),
                "gemv_row: the attention-merge prologue needs 1 <= n_parts <= %d, n_heads <= 16, K == 64 n_heads (M > 1: parts_row_stride)", PARTS_MAXS);
    const int KB = p.K / 256;
    CBX_REQUIRE(KB == 1 || KB == 3 || KB == 4 || KB == 12 || KB == 16, "gemv_row: K=%d (served: 256, 768, 1024, 3072, 4096)", p.K);
    const int maxR = p.parts ? 2 : KB <= 4 ? 8 : 2;
    int R = p.rows_per_wave > 0 ? p.rows_per_wave : (p.N + 1023) / 1024;  // auto: ~1024 waves (4 per CU) when N allows it
    R = R > maxR ? maxR : R;
    hipStream_t st = (hipStream_t)stream;
    switch (KB) {
        case 1: return launch_r<1>(p, R, st);
        case 3: return launch_r<3>(p, R, st);
        case 4: return launch_r<4>(p, R, st);
        case 12: return launch_r<12>(p, R, st);
        default: return launch_r<16>(p, R, st);
    }
}

extern "C" int cbx_decode_attn_parts(const cbx_attn_parts_t* pa, void* stream) {
    CBX_REQUIRE(pa, "decode_attn_parts: null descriptor");
    const cbx_attn_parts_t& a = *pa;
    CBX_REQUIRE(a.qkv && a.positions && a.kc && a.vc && a.parts && (!a.cos_t == !a.sin_t), "decode_attn_parts: null operand");
    CBX_REQUIRE(a.rows >= 1 && a.n_heads >= 1 && a.n_splits >= 1 && a.n_splits <= PARTS_MAXS && a.max_ctx >= 1, "decode_attn_parts: rows, heads, 1 <= n_splits <= %d, max_ctx", PARTS_MAXS);
    CBX_REQUIRE(a.ld_qkv % 4 == 0 && a.cache_row_stride % 4 == 0 && a.cache_head_stride % 4 == 0 && a.cache_head_stride >= (long)a.max_ctx * 64,
                "decode_attn_parts: alignment / max_ctx positions of cache behind every (row, head)");
    CBX_REQUIRE((((uintptr_t)a.qkv | (uintptr_t)a.kc | (uintptr_t)a.vc | (uintptr_t)a.parts | (uintptr_t)a.cos_t | (uintptr_t)a.sin_t) & 15) == 0, "decode_attn_parts: 16-byte alignment");
    const dim3 grid(a.n_heads, a.rows, a.n_splits), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (a.chunks == 8) hipLaunchKernelGGL(decode_attn_parts_kernel<8>, grid, block, 0, st, a);
    else if (a.chunks == 2) hipLaunchKernelGGL(decode_attn_parts_kernel<2>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(decode_attn_parts_kernel<4>, grid, block, 0, st, a);
    return cbx_check_launch("decode_attn_parts");
}
