// Two dependent decode GEMVs in one launch (cbx_gemv_pair_f32): the o projection (producer) and the RMSNorm-folded gate | up SwiGLU GEMV
// (consumer) of a Llama decoder layer inside T3.inference's loop (reference models/t3/t3.py:378-386 via HF LlamaDecoderLayer) -- or a down
// projection that adds the residual itself and the next layer's RMSNorm-folded q/k/v GEMV.
//
// The body below is the body of gemv_kernel (gemv_decode.hip: operand layouts, load batches, MFMA order, fixed-order reduction, epilogue --
// see the comments there) with the block indices as arguments and a consumer mode; it lives in its own translation unit so that the
// instruction streams of the established gemv_kernel instantiations -- the measured decode path -- stay exactly what they were.
// Written after the GPU budget of round 3 was spent: verified on the SIMT emulator (bit-identical to the two launches), never run on
// hardware, off by default (T3Engine.tune["pair_ogu"], ["pair_dq"] / CBX_T3_TUNE="pair_ogu=1,pair_dq=1,od_tc=4,d_ks2=1,d_nw2=8").
#include <stdlib.h>
#include "cbx_common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 bf16x4_widen(const u32x4 u, int h) {
    const unsigned a = h ? u[2] : u[0], b = h ? u[3] : u[1];
    return f32x4{__uint_as_float(a << 16), __uint_as_float(a & 0xffff0000u), __uint_as_float(b << 16), __uint_as_float(b & 0xffff0000u)};
}

// cbx_gemv_t.half_tile -> output columns per workgroup (0: 16; 1 or 8: 8; 12; 4)
__host__ __device__ __forceinline__ int gemv_tile_cols(int half_tile) { return half_tile == 0 ? 16 : half_tile == 1 ? 8 : half_tile; }

// GemvDep (DEP = true, cbx_gemv_pair_f32): this workgroup is the CONSUMER of another GEMV that runs in the same launch.  Its weight stream
// does not depend on the producer, so all of its first load batch is requested at once; only then does it wait for the producer's
// workgroups (done[x] counts the finished producers with blockIdx % 8 == x: eight words, so that 128-256 arrivals do not serialise on one
// address), and only then does it read x / res.  `spins` bounds the wait: a consumer that never sees its producers raises *err and goes on
// (wrong data, reported by the host) instead of hanging the GPU.
struct GemvDep {
    int* done;     // [8] arrivals of the producer role, reset by the last consumer through `passed`
    int* passed;   // consumers that have seen all producers
    int* err;      // set to 1 by a consumer whose wait ran out
    int n_prod, n_cons, spins;
};

__device__ __forceinline__ void gemv_wait_producers(const GemvDep& dep, const int tid) {
    if (tid < 64) {  // wave 0 polls: lane x < 8 watches the producers with blockIdx % 8 == x
        const int want = tid < 8 ? (dep.n_prod + 7 - tid) >> 3 : 0;
        int spins = 0;
        while (true) {
            const int got = tid < 8 ? __hip_atomic_load(dep.done + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            if (__all(got >= want)) break;
            if (++spins > dep.spins) {  // never on a healthy run: the producers were dispatched before this workgroup
                if (tid == 0) __hip_atomic_store(dep.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (tid == 0) {  // the last consumer through re-arms the counters for the next launch (everybody has stopped reading them)
        const int t = __hip_atomic_fetch_add(dep.passed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == dep.n_cons - 1) {
            for (int x = 0; x < 8; ++x) __hip_atomic_store(dep.done + x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dep.passed, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int MT, int NW, bool SWIGLU, bool PK, bool XPK, bool RMS, int NP, bool WB = false, bool D8 = false, bool DEP = false>
__device__ __forceinline__ void gemv_body(const cbx_gemv_t& p, const unsigned bx, const unsigned by, const GemvDep& dep) {
    static_assert(NP == 0 || (RMS && MT == 1), "partial-sum operand: RMS variant, one row tile");
    static_assert(!DEP || (MT == 1 && PK && XPK && NP == 0 && !D8), "consumer role: packed operands, one row tile");
    static_assert(!WB || (PK && XPK), "bf16 weights: packed operands only");
    __shared__ __attribute__((aligned(16))) float red[(SWIGLU ? 2 : 1) * NW * MT * 256];
    __shared__ float ssq[RMS ? NW * MT * 16 : 1];
    __shared__ float ssx[RMS ? NW * MT * 16 : 1];  // row sums (LayerNorm form only)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    // tc = output columns per workgroup: 16, or the narrow tiles 12 / 8 / 4 (cbx_gemv_t.half_tile) that give a projection 4/3, 2 or 4
    // times the workgroups (lanes c >= tc idle in the B operand): 8 for the two N = 1024 projections, 12 puts q/k/v (N = 3072) and 4 the
    // o / down projections on exactly 256 workgroups
    const int tc = PK ? gemv_tile_cols(p.half_tile) : 16;
    const int n0 = bx * tc, ks = by;
    const int kper = p.K / (p.ksplit * NW);
    const int kbeg = (ks * NW + w) * kper;
    const int nit = kper / 32;

    // row of W streamed by this lane (swiglu: gate row, the matching up row is 32 rows further in the packed image)
    long wrow;
    bool wok;
    const float *wp, *wp2;
    // floats between consecutive 32-deep K blocks of this lane's stream / between the two 16-B halves of a block; the bf16 image is half
    // as large (one 16-byte load per block: 8 bf16 per lane)
    const int WBLK = PK ? (32 * tc) / (WB ? 2 : 1) : 32;
    const int WHALF = PK ? 16 * tc : 4;
    if constexpr (PK) {
        // packed image: tile-major [tile][K/32][2][64 lanes][4]; swiglu: feature tile f -> tiles 2f (gate), 2f+1 (up); N is padded
        // to whole tiles by the packer, so every load is in range
        const long kb = p.K >> 5;
        const long tile = SWIGLU ? 2L * bx : (long)bx;
        wok = c < tc;
        const int cl = wok ? c : tc - 1;  // idle lanes re-read the tile's last row (loads are unconditional)
        if constexpr (WB) {  // [tile][K/32][lanes][8 bf16] = 4 floats per lane per block
            wp = p.W + (tile * kb + (kbeg >> 5)) * (16 * tc) + (q * tc + cl) * 4;
            wp2 = wp + kb * 256;
        } else {
            wp = p.W + (tile * kb + (kbeg >> 5)) * (32 * tc) + (q * tc + cl) * 4;
            wp2 = wp + kb * 512;
        }
    } else {
        if constexpr (SWIGLU) {
            const int f = n0 + c;  // feature index
            wok = f < p.N;
            wrow = (long)(f >> 5) * 64 + (f & 31);
        } else {
            wok = (n0 + c) < p.N;
            wrow = n0 + c;
        }
        wp = p.W + (wok ? wrow : 0) * p.ldw + kbeg + 8 * q;
        wp2 = wp + 32 * p.ldw;
    }
    const float* xp[MT];
    bool xok[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int m = t * 16 + c;
        xok[t] = m < p.M;
        if constexpr (XPK)  // packed x: [row tile][K/32][2][64][4], rows padded to whole tiles (pad rows hold finite values)
            xp[t] = p.x + ((long)t * (p.K >> 5) + (kbeg >> 5)) * 512 + lane * 4;
        else
            xp[t] = p.x + (long)(xok[t] ? m : 0) * p.ldx + kbeg + 8 * q;
    }

    const float* nwp = RMS ? p.norm_w + kbeg + 8 * q : nullptr;  // this lane's k indices: kbeg + 32*blk + 8*q + 4*h + s
    float ss[MT], sx[MT];
    f32x4 acc[MT], acc2[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        ss[t] = 0.f;
        sx[t] = 0.f;
    }
    // ---- epilogue operands requested FIRST (opt-in: cbx_set_gemv_epilogue_prefetch / p.reserved1, written after the GPU budget of round 3
    // was spent: emulator-verified, timed by the autotuner).  The element(s) a thread finishes after the reduction are known now; its
    // residual, bias and LayerNorm-fold constants do not depend on the contraction, so their loads go out with the first weight batch instead
    // of after the LDS reduction -- where each is a dependent global round trip (~1 us) on the critical path of a launch that lasts 5-9 us.
    // Same values, same order of the additions: results unchanged bit for bit.  (res may alias out: a thread reads exactly the element it
    // writes.)
    const bool PRE = !DEP && p.reserved1 != 0;  // uniform (kernel argument); a consumer's residual may be its producer's output
    constexpr int EIT = (MT * 256 + NW * 64 - 1) / (NW * 64);
    float e_res[EIT], e_bias[EIT], e_cw[EIT], e_cb[EIT];
    long e_o[EIT];
    int e_n[EIT];
#pragma unroll
    for (int j = 0; j < EIT; ++j) {
        const int e = tid + j * NW * 64;
        const int t = e >> 8, rc = e & 255, row = rc >> 4, col = rc & 15;
        // threads without an element (idle waves, rows >= M, columns past the tile / N) address this workgroup's element (0, n0): the loads
        // below stay unconditional per lane (a lane-predicated load would put a wait in front of the weight stream)
        const bool ok = e < MT * 256 && (t * 16 + row) < p.M && (n0 + col) < p.N && col < tc;
        const int tt = ok ? t : 0, rr = ok ? row : 0, n = ok ? n0 + col : n0;
        if (p.out_packed)  // the consumer's lane-ordered operand layout (its K = this N): see cbx.h
            e_o[j] = (long)ks * p.part_stride + (((long)tt * (p.N >> 5) + (n >> 5)) * 2 + ((n >> 2) & 1)) * 256 + ((((n >> 3) & 3) << 4) + rr) * 4 + (n & 3);
        else
            e_o[j] = (long)ks * p.part_stride + (long)(tt * 16 + rr) * p.ldo + n;
        e_n[j] = n;
        e_res[j] = e_bias[j] = e_cw[j] = e_cb[j] = 0.f;
    }
    if (PRE && p.res) {  // uniform branches (kernel arguments)
#pragma unroll
        for (int j = 0; j < EIT; ++j) e_res[j] = p.res[e_o[j]];
    }
    if constexpr (!SWIGLU) {
        if (PRE && p.bias && ks == 0) {
#pragma unroll
            for (int j = 0; j < EIT; ++j) e_bias[j] = p.bias[e_n[j]];
        }
    }
    if constexpr (RMS) {
        if (PRE && p.ln_cw) {
#pragma unroll
            for (int j = 0; j < EIT; ++j) e_cw[j] = p.ln_cw[e_n[j]], e_cb[j] = p.ln_cb[e_n[j]];
        }
    }
    // (consumer of the SwiGLU form: two K blocks of gate AND up per batch -- 64 KiB per workgroup before the wait -- keep it at <= 128 VGPRs, i.e. two
    // workgroups per CU: the same bytes per CU in flight, twice the consumers co-resident with their producers; same block order, same results)
    constexpr int DEPTH = D8 ? 8 : (DEP && SWIGLU) ? 2 : (MT == 1) ? 4 : 2;  // K blocks (2 KiB of W per wave each) issued before the first MFMA
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // Loads are UNCONDITIONAL (out-of-range lanes / K blocks re-read a valid address and are zeroed by a select): a predicated
    // load makes hipcc join all of them behind one vmcnt(0); unconditional ones get counted waits, so the MFMAs of K block d
    // start while blocks d+1.. are still in flight.
    for (int it0 = 0; it0 < nit; it0 += DEPTH) {
        f32x4 wv[DEPTH][2], uv[DEPTH][2], xv[DEPTH][MT][2], nv[DEPTH][2], pv[DEPTH][NP > 0 ? NP : 1][2];
        bool on[DEPTH];
        if constexpr (DEP) {  // consumer role: the whole batch of weights first, then the wait, then x
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int off = ((it0 + d) < nit ? (it0 + d) : 0) * WBLK;
                wv[d][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp + off));
                if constexpr (!WB) wv[d][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp + off + WHALF));
                if constexpr (SWIGLU) {
                    uv[d][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp2 + off));
                    if constexpr (!WB) uv[d][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp2 + off + WHALF));
                }
                if constexpr (RMS) {  // norm weights: constants as well
                    nv[d][0] = *reinterpret_cast<const f32x4*>(nwp + ((it0 + d) < nit ? (it0 + d) : 0) * 32);
                    nv[d][1] = *reinterpret_cast<const f32x4*>(nwp + ((it0 + d) < nit ? (it0 + d) : 0) * 32 + 4);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (it0 == 0) gemv_wait_producers(dep, tid);
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            on[d] = (it0 + d) < nit;
            const int blk = on[d] ? (it0 + d) : 0;
            const int off = blk * WBLK;
            if constexpr (!DEP) {
                wv[d][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp + off));
                if constexpr (!WB) wv[d][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp + off + WHALF));
                if constexpr (SWIGLU) {
                    uv[d][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp2 + off));
                    if constexpr (!WB) uv[d][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp2 + off + WHALF));
                }
            }
            const int xoff = blk * (XPK ? 512 : 32);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                xv[d][t][0] = *reinterpret_cast<const f32x4*>(xp[t] + xoff);
                xv[d][t][1] = *reinterpret_cast<const f32x4*>(xp[t] + xoff + (XPK ? 256 : 4));
            }
            if constexpr (RMS && !DEP) {
                nv[d][0] = *reinterpret_cast<const f32x4*>(nwp + blk * 32);
                nv[d][1] = *reinterpret_cast<const f32x4*>(nwp + blk * 32 + 4);
            }
            if constexpr (NP > 0) {
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const float* pp = p.xpart + (long)j * p.xpart_stride + (xp[0] - p.x);
                    pv[d][j][0] = *reinterpret_cast<const f32x4*>(pp + xoff);
                    pv[d][j][1] = *reinterpret_cast<const f32x4*>(pp + xoff + 256);
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the issue order block by block, so block d's wait is vmcnt(later blocks)
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const bool won = on[d] && wok;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 wq, uq = zero4;
                if constexpr (WB) {
                    wq = won ? bf16x4_widen(__builtin_bit_cast(u32x4, wv[d][0]), h) : zero4;
                    if constexpr (SWIGLU) uq = won ? bf16x4_widen(__builtin_bit_cast(u32x4, uv[d][0]), h) : zero4;
                } else {
                    wq = won ? wv[d][h] : zero4;
                    if constexpr (SWIGLU) uq = won ? uv[d][h] : zero4;
                }
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    f32x4 xq = xv[d][t][h];
                    if constexpr (NP > 0) {
#pragma unroll
                        for (int j = 0; j < NP; ++j) xq += pv[d][j][h];
                        if (p.x_out && bx == 0 && on[d])  // the reduced residual stream, same packed address as x
                            *reinterpret_cast<f32x4*>(p.x_out + (xp[0] - p.x) + (it0 + d) * 512 + h * 256) = xq;
                    }
                    xq = (on[d] && xok[t]) ? xq : zero4;
                    if constexpr (RMS) {
                        ss[t] += (xq[0] * xq[0] + xq[1] * xq[1]) + (xq[2] * xq[2] + xq[3] * xq[3]);
                        sx[t] += (xq[0] + xq[1]) + (xq[2] + xq[3]);
                        xq *= nv[d][h];
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xq[s], wq[s], acc[t], 0, 0, 0);
                        if constexpr (SWIGLU) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xq[s], uq[s], acc2[t], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- fixed-order reduction over the NW K-slices of this workgroup.  D map: row = q*4 + r, col = c.
    float* r1 = red;
    float* r2 = red + NW * MT * 256;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            r1[(w * MT + t) * 256 + (q * 4 + r) * 16 + c] = acc[t][r];
            if constexpr (SWIGLU) r2[(w * MT + t) * 256 + (q * 4 + r) * 16 + c] = acc2[t][r];
        }
    if constexpr (RMS) {  // lanes (c, q = 0..3) hold the four q-parts of row c's sum over this wave's K slice
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            float v = ss[t];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            float u = sx[t];
            u += __shfl_xor(u, 16);
            u += __shfl_xor(u, 32);
            if (q == 0) {
                ssq[(w * MT + t) * 16 + c] = v;
                ssx[(w * MT + t) * 16 + c] = u;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < EIT; ++j) {
        const int e = tid + j * NW * 64;
        if (e >= MT * 256) break;
        const int t = e >> 8, rc = e & 255, row = rc >> 4, col = rc & 15;
        const int m = t * 16 + row, n = n0 + col;
        if (m >= p.M || n >= p.N || col >= tc) continue;
        float v = 0.f, v2 = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
            v += r1[(ww * MT + t) * 256 + rc];
            if constexpr (SWIGLU) v2 += r2[(ww * MT + t) * 256 + rc];
        }
        if constexpr (RMS) {
            float sq = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) sq += ssq[(ww * MT + t) * 16 + row];
            if (p.ln_cw) {  // LayerNorm form (GPT-2): y = (x - mean) rstd w + b  =>  out = rstd (acc - mean cw[n]) + cb[n]
                float su = 0.f;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) su += ssx[(ww * MT + t) * 16 + row];
                const float mean = su / (float)p.K;
                const float rstd = rsqrtf(fmaxf(sq / (float)p.K - mean * mean, 0.f) + p.eps);
                v = PRE ? rstd * (v - mean * e_cw[j]) + e_cb[j] : rstd * (v - mean * p.ln_cw[n]) + p.ln_cb[n];
            } else {
                const float rstd = rsqrtf(sq / (float)p.K + p.eps);
                v *= rstd;
                v2 *= rstd;
            }
        }
        if constexpr (SWIGLU) {
            v = (v / (1.0f + __expf(-v))) * v2;
        } else {
            if (p.bias && ks == 0) v += PRE ? e_bias[j] : p.bias[n];
            if (p.act) v = cbx_act(v, p.act, 0.f, 0.f);  // only meaningful with ksplit == 1
        }
        if (p.res) v += PRE ? e_res[j] : p.res[e_o[j]];  // residual stream in the same layout as out (in place is fine: one thread per element, read before written)
        p.out[e_o[j]] = v;
    }
}

// Two dependent GEMVs in ONE launch (cbx_gemv_pair_f32; written after the GPU budget of round 3 was spent: emulator-verified, never run on
// hardware, off by default).  Workgroups [0, n_prod) run the producer (a plain packed GEMV with its residual epilogue: the o projection),
// workgroups [n_prod, n_prod + n_cons) the consumer (RMSNorm-folded SwiGLU GEMV: gate | up), whose x operand is the producer's output.  The
// hardware dispatches workgroups in index order, so every producer is resident or finished before the first consumer starts: a consumer
// requests ALL its weights (128 KiB, one load batch per wave -- they do not depend on x), then waits on the producers' arrival counters, then
// reads x.  What this removes from the chain of dependent launches: one kernel boundary and the consumer's cold start (launch, first HBM
// round trip), which now overlap the producer.  Same arithmetic in the same order as the two launches: bit-identical results.
// CSW: the consumer is the SwiGLU form (gate | up after the o projection); else the plain RMSNorm-folded GEMV (the next layer's q/k/v after a
// down projection that adds the residual itself).
template <bool CSW>
__global__ __launch_bounds__(512) void gemv_pair_kernel(const cbx_gemv_t pa, const cbx_gemv_t pb, const GemvDep dep) {
    if ((int)blockIdx.x < dep.n_prod) {
        gemv_body<1, 8, false, true, true, false, 0, false, false, false>(pa, blockIdx.x, 0, dep);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(dep.done + (blockIdx.x & 7), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        gemv_body<1, 8, CSW, true, true, true, 0, false, false, true>(pb, blockIdx.x - dep.n_prod, 0, dep);
    }
}

// The chain o projection -> gate | up -> down projection -> (next layer's q/k/v | speech head) of a Llama decoder layer as ONE launch
// (cbx_gemv_chain_f32): four roles in block-index order, every role but the first in the consumer mode above (its first weight batch is in
// flight before it waits for the role in front of it), every role but the last signalling its own arrival counters.  One counter set per
// edge: sync + 16 * edge = done[8], passed.  With attention that is 2 launches per layer instead of 5.
struct GemvChain {
    cbx_gemv_t op[4];
    int first[5];  // first block of role r; first[4] = grid size
    int* sync;     // 3 edges x 16 ints (zeroed once), sync[63] = error word
    int spins;
};

// SW1: role 1 is the SwiGLU form (Llama: gate | up); else a plain normalisation-folded GEMV with its activation epilogue (GPT-2: ln_2 + c_fc +
// gelu_new; roles 0 / 2 then carry a bias, roles 1 / 3 the LayerNorm form of cbx_gemv_t.ln_cw / ln_cb -- all of it epilogue work of the body).
template <bool SW1>
__global__ __launch_bounds__(512) void gemv_chain_kernel(const GemvChain c) {
    const unsigned b = blockIdx.x;
    const int r = (b >= (unsigned)c.first[1]) + (b >= (unsigned)c.first[2]) + (b >= (unsigned)c.first[3]);  // uniform
    const unsigned lb = b - c.first[r];
    GemvDep dep{};
    if (r > 0) {
        dep.done = c.sync + 16 * (r - 1), dep.passed = dep.done + 8, dep.err = c.sync + 63;
        dep.n_prod = c.first[r] - c.first[r - 1], dep.n_cons = c.first[r + 1] - c.first[r], dep.spins = c.spins;
    }
    switch (r) {
        case 0: gemv_body<1, 8, false, true, true, false, 0, false, false, false>(c.op[0], lb, 0, dep); break;  // o projection (+ residual)
        case 1: gemv_body<1, 8, SW1, true, true, true, 0, false, false, true>(c.op[1], lb, 0, dep); break;      // norm + gate | up + SwiGLU (or c_fc + gelu)
        case 2: gemv_body<1, 8, false, true, true, false, 0, false, false, true>(c.op[2], lb, 0, dep); break;   // down projection (+ residual)
        default: gemv_body<1, 8, false, true, true, true, 0, false, false, true>(c.op[3], lb, 0, dep); break;   // RMSNorm + q/k/v (or the head)
    }
    if (r < 3) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(c.sync + 16 * r + (lb & 7), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace

extern "C" int cbx_gemv_chain_f32(const cbx_gemv_t* ops4, int* sync_ws, int spins, void* stream) {
    CBX_REQUIRE(ops4 && sync_ws, "gemv_chain: null operand");
    GemvChain c;
    c.first[0] = 0;
    for (int r = 0; r < 4; ++r) {
        cbx_gemv_t& p = c.op[r];
        p = ops4[r];
        p.ksplit = 1, p.reserved1 = 0;
        const bool rms = r == 1 || r == 3;
        CBX_REQUIRE(p.x && p.W && p.out, "gemv_chain: null operand in role %d", r);
        CBX_REQUIRE(p.M >= 1 && p.M <= 16 && p.M == ops4[0].M, "gemv_chain: 1..16 rows, the same for all four");
        CBX_REQUIRE(p.w_packed && p.x_packed && !p.w_bf16 && p.nw == 8 && p.K % 256 == 0, "gemv_chain: role %d needs packed fp32 operands, 8 waves, K %% 256 == 0", r);
        CBX_REQUIRE(!p.n_xpart, "gemv_chain: role %d: no partial-sum operand", r);
        CBX_REQUIRE(rms ? (p.norm_w && !p.res && (!p.ln_cw || (p.ln_cb && !p.bias && !p.swiglu))) : (!p.norm_w && !p.ln_cw && !p.act),
                    "gemv_chain: roles 1 and 3 are normalisation-folded (RMSNorm, or the LayerNorm form with its bias folded into ln_cb), roles 0 and 2 plain (+ bias, + residual)");
        CBX_REQUIRE(!p.swiglu || r == 1, "gemv_chain: only role 1 may be the SwiGLU form");
        CBX_REQUIRE(!p.act || (r == 1 && !p.swiglu), "gemv_chain: an activation epilogue belongs to a non-SwiGLU role 1");
        CBX_REQUIRE(p.half_tile == 0 || p.half_tile == 1 || p.half_tile == 8 || p.half_tile == 12 || p.half_tile == 4, "gemv_chain: half_tile");
        CBX_REQUIRE(!p.swiglu || (p.half_tile == 0 && p.N % 32 == 0), "gemv_chain: the SwiGLU role uses 16-column tiles, N %% 32 == 0");
        CBX_REQUIRE(!p.out_packed || p.N % 32 == 0, "gemv_chain: out_packed needs N %% 32 == 0");
        const int tc = gemv_tile_cols(p.half_tile);
        c.first[r + 1] = c.first[r] + (p.N + tc - 1) / tc;
    }
    c.sync = sync_ws, c.spins = spins > 0 ? spins : (1 << 16);
    if (c.op[1].swiglu) hipLaunchKernelGGL(gemv_chain_kernel<true>, dim3(c.first[4]), dim3(512), 0, (hipStream_t)stream, c);
    else hipLaunchKernelGGL(gemv_chain_kernel<false>, dim3(c.first[4]), dim3(512), 0, (hipStream_t)stream, c);
    return cbx_check_launch("gemv_chain");
}

extern "C" int cbx_gemv_pair_f32(const cbx_gemv_t* pa, const cbx_gemv_t* pb, int* sync_ws, int spins, void* stream) {
    cbx_gemv_t a = *pa, b = *pb;
    a.ksplit = b.ksplit = 1;
    a.reserved1 = b.reserved1 = 0;
    CBX_REQUIRE(sync_ws && a.x && a.W && a.out && b.x && b.W && b.out, "gemv_pair: null operand");
    CBX_REQUIRE(a.M >= 1 && a.M <= 16 && b.M == a.M, "gemv_pair: 1..16 rows, the same for both");
    CBX_REQUIRE(a.w_packed && a.x_packed && b.w_packed && b.x_packed && !a.w_bf16 && !b.w_bf16, "gemv_pair: packed fp32 operands");
    CBX_REQUIRE(a.nw == 8 && b.nw == 8, "gemv_pair: 8 waves per workgroup");
    CBX_REQUIRE(!a.swiglu && !a.norm_w && !a.n_xpart && !a.act && !a.bias, "gemv_pair: the producer is a plain GEMV (+ residual)");
    CBX_REQUIRE(b.norm_w && !b.n_xpart && !b.ln_cw && !b.act && !b.bias && !b.res, "gemv_pair: the consumer is an RMSNorm-folded GEMV (SwiGLU or plain)");
    CBX_REQUIRE(!b.swiglu || (b.half_tile == 0 && b.N % 32 == 0), "gemv_pair: the SwiGLU consumer uses 16-column tiles, N %% 32 == 0");
    CBX_REQUIRE(b.half_tile == 0 || b.half_tile == 1 || b.half_tile == 8 || b.half_tile == 12 || b.half_tile == 4, "gemv_pair: half_tile");
    CBX_REQUIRE(!b.out_packed || b.N % 32 == 0, "gemv_pair: out_packed needs N %% 32 == 0");
    CBX_REQUIRE(a.half_tile == 0 || a.half_tile == 1 || a.half_tile == 8 || a.half_tile == 12 || a.half_tile == 4, "gemv_pair: half_tile");
    CBX_REQUIRE(a.K % 256 == 0 && b.K % 256 == 0, "gemv_pair: K must be a multiple of 32 * 8 waves");
    CBX_REQUIRE(!a.out_packed || a.N % 32 == 0, "gemv_pair: out_packed needs N %% 32 == 0");
    GemvDep dep;
    dep.done = sync_ws, dep.passed = sync_ws + 8, dep.err = sync_ws + 9;
    dep.n_prod = (a.N + gemv_tile_cols(a.half_tile) - 1) / gemv_tile_cols(a.half_tile);
    dep.n_cons = (b.N + gemv_tile_cols(b.half_tile) - 1) / gemv_tile_cols(b.half_tile);
    dep.spins = spins > 0 ? spins : (1 << 16);
    if (b.swiglu) hipLaunchKernelGGL(gemv_pair_kernel<true>, dim3(dep.n_prod + dep.n_cons), dim3(512), 0, (hipStream_t)stream, a, b, dep);
    else hipLaunchKernelGGL(gemv_pair_kernel<false>, dim3(dep.n_prod + dep.n_cons), dim3(512), 0, (hipStream_t)stream, a, b, dep);
    return cbx_check_launch("gemv_pair");
}

