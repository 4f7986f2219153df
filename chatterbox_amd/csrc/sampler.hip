// Device-side T3 sampler: CFG combine + repetition penalty + temperature + min-p / top-k / top-p + softmax +
// inverse-CDF sampling, one 1024-thread workgroup per utterance, the whole vocabulary row resident in LDS.
// Nothing returns to the host: the kernel appends the token, updates the repetition bitmap, the EOS flag and
// the per-row position / context-length scalars that the next decode step (same hipGraph) reads.
//
// Semantics follow the reference loop body (models/t3/t3.py:339-368) and the HF logits processors it calls
// (transformers/generation/logits_process.py: RepetitionPenalty, Temperature, MinP, TopK, TopP); torch.multinomial
// is replaced by inverse-CDF sampling on a caller-provided uniform, which is how the oracle injects its RNG.
#include "cbx_common.h"

namespace {
CBX_TRC_TU

constexpr int NT = 1024;
constexpr int MAXV = 8448;  // >= 8194, multiple of 64

// Block-wide reductions through a PING-PONG pair of LDS rows: reduction k writes row k & 1, so the writes of reduction k cannot overtake the reads of
// reduction k - 2 (every thread passed reduction k - 1's barrier in between) and ONE barrier per reduction suffices (rounds 1-5: two; the top-k / top-p
// bisections of Turbo run 62 reductions per token).  Same values, same order of the additions.
struct Red {
    float* buf;  // 2 x NT / 64 floats, 16-byte aligned
    int k;
};
__device__ __forceinline__ float block_max(float v, Red& red) {
    v = wave_max(v);
    float* b = red.buf + (red.k++ & 1) * (NT / 64);
    if ((threadIdx.x & 63) == 0) b[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = b[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) r = fmaxf(r, b[i]);
    return r;
}
__device__ __forceinline__ float block_sum(float v, Red& red) {
    v = wave_sum(v);
    float* b = red.buf + (red.k++ & 1) * (NT / 64);
    if ((threadIdx.x & 63) == 0) b[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) r += b[i];
    return r;
}
__device__ __forceinline__ double block_sum_d(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) r += red[i];
    return r;
}

// HF RepetitionPenaltyLogitsProcessor on the ids seen so far
__device__ __forceinline__ void rep_penalty(float* l, const unsigned char* seen, int V, float pen) {
    for (int i = threadIdx.x; i < V; i += NT)
        if (seen[i]) l[i] = l[i] < 0.f ? l[i] * pen : l[i] / pen;
}

// (Round 6 measured a 4-way search -- three thresholds per barrier round, half the rounds, same cut -- at 69 us against the bisection's 51 us for Turbo's
// top-k + top-p, 59 us once the wave reductions ran on DPP: like round 3's 16-way variant (108 us) it loses, because every thread re-adds the partial sums of
// all 16 waves per threshold.  The bisection stays; what round 6 changed is underneath it: wave_sum / wave_max are DPP / permlane butterflies now.)
// HF TopPLogitsWarper: drop the ascending-sorted prefix whose cumulative probability is <= 1 - top_p.
// Sort-free: token i is dropped iff mass{p_j <= p_i} <= 1 - top_p; the cut value is found by bisection on the
// (monotone) bit pattern of the un-normalised probabilities.
__device__ void top_p_filter(float* l, int V, float top_p, Red& red) {
    constexpr int EPT = (MAXV + NT - 1) / NT;  // elements per thread: their un-normalised probabilities are computed ONCE and kept in registers
    float m = -INFINITY;
    for (int i = threadIdx.x; i < V; i += NT) m = fmaxf(m, l[i]);
    m = block_max(m, red);
    float e[EPT];
    float z = 0.f;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int i = threadIdx.x + j * NT;
        e[j] = i < V ? __expf(l[i] - m) : 2.0f;  // 2 > every threshold in [0, 1]: a slot past the vocabulary never counts
        z += i < V ? e[j] : 0.f;
    }
    z = block_sum(z, red);
    const float budget = (1.0f - top_p) * z;
    unsigned lo = 0u, hi = __float_as_uint(1.0f);  // e in [0,1]; invariant: mass{e <= lo} <= budget
    while (lo + 1 < hi) {
        unsigned mid = lo + (hi - lo) / 2;
        float thr = __uint_as_float(mid);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < EPT; ++j) s += e[j] <= thr ? e[j] : 0.f;
        s = block_sum(s, red);
        if (s <= budget) lo = mid; else hi = mid;
    }
    const float cut = __uint_as_float(lo);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int i = threadIdx.x + j * NT;
        if (i < V && e[j] <= cut && e[j] < 1.0f) l[i] = -INFINITY;  // the arg-max is always kept (min_tokens_to_keep = 1)
    }
    __syncthreads();
}

// HF TopKLogitsWarper: scores < k-th largest -> -inf (bisection on the order-preserving integer image).
__device__ void top_k_filter(float* l, int V, int k, Red& red) {
    if (k <= 0 || k >= V) return;
    constexpr int EPT = (MAXV + NT - 1) / NT;
    auto key = [](float f) -> unsigned {
        unsigned u = __float_as_uint(f);
        return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    };
    unsigned ky[EPT];  // the thread's keys, computed once (0 for slots past the vocabulary: never >= a threshold >= 1)
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int i = threadIdx.x + j * NT;
        ky[j] = i < V ? key(l[i]) : 0u;
    }
    unsigned lo = 0u, hi = 0xFFFFFFFFu;  // largest key with count{key(l) >= key} >= k
    while (lo < hi) {
        unsigned mid = lo + (hi - lo) / 2 + 1;
        float c = 0.f;
#pragma unroll
        for (int j = 0; j < EPT; ++j) c += ky[j] >= mid ? 1.f : 0.f;
        c = block_sum(c, red);
        if (c >= (float)k) lo = mid; else hi = mid - 1;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int i = threadIdx.x + j * NT;
        if (i < V && ky[j] < lo) l[i] = -INFINITY;
    }
    __syncthreads();
}

__global__ __launch_bounds__(NT) void t3_sample_kernel(const cbx_sampler_t p) {
    __builtin_amdgcn_s_setprio(3);  // decode-step kernels are latency-bound and issue little: beside a co-resident workgroup of another stream (the throughput schedule, profiles/r05_overlap_*) their waves go first at the SIMD's issue arbiter; alone on the CU it changes nothing
    __shared__ float l[MAXV];
    __shared__ __attribute__((aligned(16))) float red_buf[2 * (NT / 64)];
    Red red{red_buf, 0};
    __shared__ double redd[NT / 64];
    __shared__ double wave_base[NT / 64];
    __shared__ int chosen;
    CBX_TRC_DECL;
    CBX_TRC_STAMP(0);

    const int b = blockIdx.x, tid = threadIdx.x, V = p.V;
    // sampling parameters: by value, or -- so that a captured hipGraph serves requests with different settings without being
    // re-captured -- per utterance from device memory
    float cfg_weight = p.cfg_weight, temperature = p.temperature, min_p = p.min_p, top_p = p.top_p, rep_pen = p.rep_penalty;
    int top_k = p.top_k, ban_token = p.ban_token, ban_from = p.ban_from;
    if (p.dev_params) {
        const float* dp = p.dev_params + (long)b * CBX_SAMPLER_NPARAMS;
        cfg_weight = dp[0], temperature = dp[1], min_p = dp[2], top_p = dp[3], rep_pen = dp[4];
        top_k = (int)dp[5], ban_token = (int)dp[6], ban_from = (int)dp[7];
    }
    if (ban_from <= 0) ban_from = V;
    if (p.done[b]) return;
    const int step = p.step[b];
    if (step >= p.max_steps) return;
    unsigned char* seen = p.seen + (long)b * V;
    const float* lc = p.logits + (long)b * p.ld;
    const float* lu = p.logits + (long)(p.B + b) * p.ld;

    for (int i = tid; i < V; i += NT) {
        float c = lc[i];
        l[i] = p.cfg ? c + cfg_weight * (c - lu[i]) : c;
    }
    __syncthreads();

    if (p.order == 0) {  // T3.inference: penalty -> temperature -> min-p -> top-p
        if (rep_pen != 1.0f) rep_penalty(l, seen, V, rep_pen);
        __syncthreads();
        if (temperature != 1.0f)
            for (int i = tid; i < V; i += NT) l[i] = l[i] / temperature;
        __syncthreads();
        if (min_p > 0.f) {
            float m = -INFINITY;
            for (int i = tid; i < V; i += NT) m = fmaxf(m, l[i]);
            m = block_max(m, red);
            float z = 0.f;
            for (int i = tid; i < V; i += NT) z += __expf(l[i] - m);
            z = block_sum(z, red);
            const float top = 1.0f / z;
            for (int i = tid; i < V; i += NT) {
                float pr = __expf(l[i] - m) / z;
                if (pr < min_p * top && l[i] < m) l[i] = -INFINITY;
            }
            __syncthreads();
        }
        if (top_p < 1.0f) top_p_filter(l, V, top_p, red);
    } else {  // T3.inference_turbo: temperature -> top-k -> top-p -> penalty
        if (temperature > 0.f && temperature != 1.0f)
            for (int i = tid; i < V; i += NT) l[i] = l[i] / temperature;
        __syncthreads();
        top_k_filter(l, V, top_k, red);
        if (top_p < 1.0f) top_p_filter(l, V, top_p, red);
        if (rep_pen != 1.0f) rep_penalty(l, seen, V, rep_pen);
        __syncthreads();
    }

    // ---- softmax + inverse CDF.  Thread t owns the contiguous ids [t*per, (t+1)*per); prefix sums in fp64.
    float m = -INFINITY;
    for (int i = tid; i < V; i += NT) m = fmaxf(m, l[i]);
    m = block_max(m, red);
    const int per = (V + NT - 1) / NT;
    const int i0 = tid * per, i1 = min(V, i0 + per);
    double loc = 0.0;
    for (int i = i0; i < i1; ++i) {
        float e = (i == ban_token || i >= ban_from) ? 0.f : __expf(l[i] - m);
        loc += (double)e;
    }
    // exclusive scan of `loc` over threads: inclusive wave scan + wave bases
    double inc = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        double t = __shfl_up(inc, o);
        if ((tid & 63) >= o) inc += t;
    }
    if ((tid & 63) == 63) redd[tid >> 6] = inc;
    __syncthreads();
    if (tid == 0) {
        double run = 0.0;
        for (int w = 0; w < NT / 64; ++w) {
            wave_base[w] = run;
            run += redd[w];
        }
        redd[0] = run;  // total
        chosen = -1;
    }
    __syncthreads();
    const double total = redd[0];
    const double excl = wave_base[tid >> 6] + inc - loc;
    const double target = (double)p.uniforms[(long)b * p.max_steps + step] * total;
    if (loc > 0.0 && target >= excl && target < excl + loc) {
        double run = excl;
        int pick = -1;
        for (int i = i0; i < i1; ++i) {
            float e = (i == ban_token || i >= ban_from) ? 0.f : __expf(l[i] - m);
            if (e > 0.f) {
                pick = i;
                run += (double)e;
                if (run > target) break;
            }
        }
        chosen = pick;
    }
    __syncthreads();
    if (chosen < 0) {
        // No owner: every id that survived the processors is banned (total == 0; happens with ban_from on random-init weights
        // when the arg-max is a banned id and min-p pruned the rest), or the target fell into a 1-ulp seam between two threads'
        // fp64 intervals.  Defined result: the allowed id with the largest CFG-combined raw logit (lowest id on ties), found by a
        // block-wide arg-max -- `chosen` is block-uniform here, so the barriers below are safe.
        float bm = -INFINITY;
        for (int i = tid; i < V; i += NT)
            if (i != ban_token && i < ban_from) {
                const float c = lc[i];
                bm = fmaxf(bm, p.cfg ? c + cfg_weight * (c - lu[i]) : c);
            }
        bm = block_max(bm, red);
        __syncthreads();
        if (tid == 0) chosen = 0x7fffffff;
        __syncthreads();
        for (int i = tid; i < V; i += NT)
            if (i != ban_token && i < ban_from) {
                const float c = lc[i];
                if ((p.cfg ? c + cfg_weight * (c - lu[i]) : c) == bm) atomicMin(&chosen, i);
            }
        __syncthreads();
    }
    if (tid == 0) {
        int tok = chosen;
        if (tok < 0 || tok >= V) tok = 0;  // no allowed id at all (or NaN logits): keep the state machine well-defined
        p.out_tokens[(long)b * p.max_steps + step] = tok;
        seen[tok] = 1;
        p.step[b] = step + 1;
        p.n_generated[b] = step + 1;
        if (tok == p.eos_token) p.done[b] = 1;
        const int nrep = p.cfg ? 2 : 1;
        for (int r = 0; r < nrep; ++r) {
            const int row = b + r * p.B;
            if (p.next_ids) p.next_ids[row] = tok;
            if (p.next_pos_ids) p.next_pos_ids[row] = step + 1;
            if (p.positions) p.positions[row] += 1;
            if (p.ctx_lens) p.ctx_lens[row] += 1;
        }
    }
#ifdef CBX_TRACE
    CBX_TRC_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CBX_TRC_STAMP(2);
    CBX_TRC_FLUSH(0x30000000u);
#endif
}

}  // namespace
CBX_TRC_SETTER(cbx_trace_set_sampler)

extern "C" int cbx_t3_sample(const cbx_sampler_t* p, void* stream) {
    CBX_REQUIRE(p && p->logits && p->seen && p->uniforms && p->step && p->out_tokens && p->done && p->n_generated,
                "t3_sample: null operand");
    CBX_REQUIRE(p->V > 0 && p->V <= MAXV, "t3_sample: V=%d exceeds %d", p->V, MAXV);
    cbx_sampler_t q = *p;
    hipLaunchKernelGGL(t3_sample_kernel, dim3(q.B), dim3(NT), 0, (hipStream_t)stream, q);
    return cbx_check_launch("t3_sample");
}
