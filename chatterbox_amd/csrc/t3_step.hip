// Stage-level C entry point of the T3 decode loop (SURVEY.md 8b "what the C-ABI replacement must export": the body of
// T3.inference's loop, reference models/t3/t3.py:338-386, for every row at once).  One call enqueues ONE token step of the Llama stack
// -- embedding gather, 30 x [q/k/v GEMV with RMSNorm and the partial-sum operand folded in, fused RoPE + cache append + attention,
// o GEMV + residual, gate/up GEMV with RMSNorm + SwiGLU, down GEMV as split-K partial images], head GEMV, device sampler -- on the
// caller's stream, with no allocation and no synchronisation, so a C / C++ host can capture it in a hipGraph and replay it per token
// exactly as chatterbox_amd/t3.py does (which calls this function when CBX_T3_CSTEP=1).  It only sequences the kernel-level entry
// points of this library; all state is caller-owned device memory described by cbx_t3_step_t.
#include "cbx_common.h"

#ifdef CBX_TRACE
extern "C" int cbx_trace_set_gemv(unsigned long long*, unsigned*, unsigned);
extern "C" int cbx_trace_set_attention(unsigned long long*, unsigned*, unsigned);
extern "C" int cbx_trace_set_sampler(unsigned long long*, unsigned*, unsigned);
extern "C" int cbx_trace_set_elementwise(unsigned long long*, unsigned*, unsigned);
// side build only (scripts/trace_decode.sh): device log of 8-word records, its append counter, its capacity in records (cbx_common.h CBX_TRACE)
extern "C" int cbx_trace_set(unsigned long long* buf, unsigned* cnt, unsigned cap) {
    int rc = cbx_trace_set_gemv(buf, cnt, cap);
    if (!rc) rc = cbx_trace_set_attention(buf, cnt, cap);
    if (!rc) rc = cbx_trace_set_sampler(buf, cnt, cap);
    if (!rc) rc = cbx_trace_set_elementwise(buf, cnt, cap);
    return rc;
}
#endif

extern "C" int cbx_t3_decode_step(const cbx_t3_step_t* d, void* stream) {
    CBX_REQUIRE(d && d->layers && d->n_layers > 0, "t3_decode_step: null descriptor");
    CBX_REQUIRE(d->rows >= 1 && d->rows <= 16, "t3_decode_step: rows=%d (this entry point serves the packed <= 16-row path)", d->rows);
    CBX_REQUIRE(d->d_ksplit == 1 || d->d_ksplit == 2 || d->d_ksplit == 4, "t3_decode_step: d_ksplit must be 1, 2 or 4");
    const int qks = d->qkv_ksplit > 1 ? d->qkv_ksplit : 1;  // ABI v11: the q/k/v projection as split-K partial sums folded by the attention launch
    CBX_REQUIRE(qks == 1 || ((qks == 2 || qks == 4) && d->dim % (256 * qks) == 0 && d->qkv_ct >= 1 && d->qkv_ct <= 4 && d->qkv_ssq && d->qkv_tile == 0 && !d->w_bf16),
                "t3_decode_step: qkv_ksplit 2 or 4 (dim %% (256 qkv_ksplit) == 0) needs qkv_ct 1 .. 4, qkv_ssq, the 16-column fp32 q/k/v image");
    CBX_REQUIRE(d->head_ct >= 0 && d->head_ct <= 4 && !(d->head_ct && d->w_bf16), "t3_decode_step: head_ct in 0 .. 4 (fp32 images)");
    const int D = d->dim, F = d->ffn, H = d->n_heads;
    float* cur = d->x_a;
    float* nxt = d->x_b;
    int rc = cbx_embed_f32(d->next_ids, d->speech_emb, d->speech_pos, d->next_pos_ids, cur, d->rows, D, D, 1.0f, 3, stream);
    if (rc) return rc;
    cbx_gemv_t g;
    auto base = [&](const float* x, const float* W, float* out, int N, int K) {
        g = cbx_gemv_t{};
        g.x = x, g.W = W, g.out = out, g.M = d->rows, g.N = N, g.K = K, g.ksplit = 1, g.nw = 8;
        g.w_packed = g.x_packed = 1, g.eps = d->eps, g.ldx = K, g.ldw = K, g.ldo = N, g.w_bf16 = d->w_bf16, g.flags = d->gemv_flags;
    };
    cbx_decode_attn_t da{};  // geometry and split-context workspace of the attention launches: per step descriptor (ABI v10), nothing process-wide
    da.qkv = d->qkv, da.positions = d->positions, da.cos_t = d->cos_t, da.sin_t = d->sin_t, da.o = d->att, da.rows = d->rows, da.n_heads = H;
    da.ld_qkv = 3 * D, da.o_ld = D, da.o_packed = 1, da.cache_row_stride = d->kv_row_stride, da.cache_head_stride = d->kv_head_stride;
    da.scale = d->attn_scale, da.unroll = d->da_unroll, da.pipeline = d->da_pipeline, da.split_min = d->da_split_min;
    da.split_ws = d->da_ws, da.split_cnt = d->da_cnt, da.split_pairs = d->da_pairs;
    if (qks > 1) da.qkv_nparts = qks, da.qkv_part_stride = (long)d->rows * 3 * D, da.qkv_ssq = d->qkv_ssq, da.rms_dim = D, da.rms_eps = d->eps;
    const long img = (long)((d->rows + 15) / 16 * 16) * D;  // floats per packed residual / partial image
    bool pending = false;                                     // split-K partial images of the previous down projection waiting to be summed
    for (int i = 0; i < d->n_layers; ++i) {
        const cbx_t3_layer_t& L = d->layers[i];
        base(cur, L.wqkv, d->qkv, 3 * D, D);
        g.norm_w = L.ln1, g.half_tile = d->qkv_tile;
        if (qks > 1) g.col_tiles = d->qkv_ct, g.ksplit = qks, g.part_stride = (long)d->rows * 3 * D, g.ssq_out = d->qkv_ssq, g.flags = 0;
        if (pending) g.n_xpart = d->d_ksplit, g.xpart = d->pd, g.xpart_stride = img, g.x_out = nxt;
        if ((rc = cbx_gemv_f32(&g, stream))) return rc;
        if (pending) {
            float* t = cur;
            cur = nxt, nxt = t;
        }
        const long kv_layer = (long)d->rows * d->kv_row_stride;
        da.kc = d->kc + i * kv_layer, da.vc = d->vc + i * kv_layer;
        if ((rc = cbx_decode_attn_rope(&da, stream))) return rc;
        base(d->att, L.wo, cur, D, D);
        g.nw = d->o_nw, g.res = cur, g.out_packed = 1, g.half_tile = d->half_tiles;
        if ((rc = cbx_gemv_f32(&g, stream))) return rc;
        base(cur, L.wgu, d->g, F, D);
        g.norm_w = L.ln2, g.swiglu = 1, g.out_packed = 1, g.nw = d->gu_nw;
        if ((rc = cbx_gemv_f32(&g, stream))) return rc;
        if (d->d_ksplit > 1) {
            base(d->g, L.wd, d->pd, D, F);
            g.ksplit = d->d_ksplit, g.part_stride = img;
        } else {  // no partial images: the down projection adds the residual in its epilogue, in place
            base(d->g, L.wd, cur, D, F);
            g.res = cur;
        }
        g.nw = d->d_nw, g.out_packed = 1, g.ldo = D, g.half_tile = d->half_tiles;
        if ((rc = cbx_gemv_f32(&g, stream))) return rc;
        pending = d->d_ksplit > 1;
    }
    base(cur, d->head, d->logits, d->vocab, D);
    g.norm_w = d->final_norm, g.ldo = d->ld_logits;
    if (d->head_ct) g.col_tiles = d->head_ct, g.flags = 0;
    if (pending) g.n_xpart = d->d_ksplit, g.xpart = d->pd, g.xpart_stride = img, g.x_out = nullptr;
    if ((rc = cbx_gemv_f32(&g, stream))) return rc;
    return d->sampler ? cbx_t3_sample(d->sampler, stream) : 0;
}

// Stage-level C entry point of the T3 PREFILL (reference models/t3/t3.py:303-335 through t3_hf_backend.py:71-111: the HF LlamaModel forward over
// the S0 prompt positions of every row, KV cache filled): n_layers x [RMSNorm, fused q/k/v GEMM, RoPE + cache fill, causal flash attention, o GEMM + residual,
// RMSNorm, gate | up GEMM with SwiGLU, down GEMM + residual] on the caller's stream -- the launches chatterbox_amd/t3.py::_layer_prefill issues one by one,
// with the same arguments (bit-identical results).  Together with cbx_t3_decode_step a C host drives the whole device side of T3.inference; the speech head on
// the last position of every row is one more cbx_layernorm_f32 + cbx_gemm_f32 of the caller's.  No allocation, no synchronisation.
extern "C" int cbx_t3_prefill(const cbx_t3_prefill_t* d, void* stream) {
    CBX_REQUIRE(d && d->layers && d->n_layers > 0 && d->x && d->h && d->qkv && d->att && d->g && d->positions && d->cache_rows && d->kc && d->vc,
                "t3_prefill: null descriptor field");
    CBX_REQUIRE(d->rows >= 1 && d->S >= 1 && d->dim == d->n_heads * 64 && d->ffn > 0, "t3_prefill: bad shape (head_dim 64)");
    const int D = d->dim, F = d->ffn, H = d->n_heads;
    const long M = (long)d->rows * d->S;
    auto linear = [&](const float* A, const float* W, float* C, const float* R, int N, int K, int swiglu) {
        cbx_gemm_t g{};
        g.A = A, g.W = W, g.C = C, g.R = R;
        g.M = (int)M, g.N = N, g.K = K, g.Cin = K, g.taps = 1, g.dil = 1, g.stride = 1, g.up = 1, g.nz1 = 1, g.nz2 = 1, g.swiglu = swiglu;
        g.alpha = 1.0f, g.lda = K, g.ldw = K, g.ldc = swiglu ? N / 2 : N, g.ldr = R ? D : 0, g.precision = d->precision;  // as the Python sequence passes it (the SwiGLU form is served by the exact kernel in every mode)
        return cbx_gemm_f32(&g, stream);
    };
    int rc = 0;
    for (int i = 0; i < d->n_layers; ++i) {
        const cbx_t3_layer_t& L = d->layers[i];
        float* kc = d->kc + (long)i * d->kv_layer_stride;
        float* vc = d->vc + (long)i * d->kv_layer_stride;
        if ((rc = cbx_layernorm_f32(d->x, d->h, L.ln1, nullptr, nullptr, M, D, D, D, d->eps, 1, CBX_ACT_NONE, 1.0f, stream))) return rc;
        if ((rc = linear(d->h, L.wqkv, d->qkv, nullptr, 3 * D, D, 0))) return rc;
        if ((rc = cbx_rope_kv_f32(d->qkv, d->positions, d->cos_t, d->sin_t, kc, vc, d->cache_rows, M, H, 3 * D, d->kv_row_stride, d->kv_head_stride, stream))) return rc;
        const long sb = (long)d->S * 3 * D, st = 3 * D;
        if (d->precision == 3 || d->precision == 6 || d->precision == 16)
            rc = cbx_flash_attn_split_f32(d->qkv, d->qkv + D, d->qkv + 2 * D, d->att, nullptr, d->rows, H, d->S, d->S, sb, st, sb, st, sb, st, (long)d->S * D, D, d->attn_scale, 1,
                                          d->precision, stream);
        else
            rc = cbx_flash_attn_f32(d->qkv, d->qkv + D, d->qkv + 2 * D, d->att, nullptr, d->rows, H, d->S, d->S, sb, st, sb, st, sb, st, (long)d->S * D, D, d->attn_scale, 1, stream);
        if (rc) return rc;
        if ((rc = linear(d->att, L.wo, d->x, d->x, D, D, 0))) return rc;
        if ((rc = cbx_layernorm_f32(d->x, d->h, L.ln2, nullptr, nullptr, M, D, D, D, d->eps, 1, CBX_ACT_NONE, 1.0f, stream))) return rc;
        if ((rc = linear(d->h, L.wgu, d->g, nullptr, 2 * F, D, 1))) return rc;
        if ((rc = linear(d->g, L.wd, d->x, d->x, D, F, 0))) return rc;
    }
    return 0;
}

// ---- The prefill of T3.inference_turbo (t3.py:392-468 -> HF GPT2Model over [speaker | prompt tokens | text | start-speech]) as ONE call (ABI v16): the launches
// T3TurboEngine.generate issues per layer -- ln_1, c_attn (+ bias), K / V append, causal attention, attention c_proj (+ bias + residual), ln_2, c_fc (+ bias +
// gelu_new), mlp c_proj (+ bias + residual) -- with the same arguments (bit-identical results).  At batch 1 the prompt's 441 rows keep the chip busy for ~0.1 ms
// per layer while nine Python-issued launches cost ~0.4 ms of host time: the prefill was host-bound (profiles/r06_ao_*.log).  prefix > 0: the first `prefix`
// positions of every row are already in the KV cache (VoicePrefixCache: speaker + prompt tokens of a voice, computed once); x holds the remaining S positions of
// every row and the attention reads keys / values [prefix | S] from the cache (cbx_flash_attn_kv_f32).
extern "C" int cbx_gpt2_prefill(const cbx_gpt2_prefill_t* d, void* stream) {
    CBX_REQUIRE(d && d->layers && d->n_layers > 0 && d->x && d->h && d->qkv && d->att && d->g && d->positions && d->cache_rows && d->kc && d->vc,
                "gpt2_prefill: null descriptor field");
    CBX_REQUIRE(d->rows >= 1 && d->S >= 1 && d->prefix >= 0 && d->dim == d->n_heads * 64, "gpt2_prefill: bad shape (head_dim 64)");
    const int D = d->dim, F = 4 * d->dim, H = d->n_heads;
    const long M = (long)d->rows * d->S;
    auto linear = [&](const float* A, const float* W, const float* bias, float* C, const float* R, int N, int K, int act) {
        cbx_gemm_t g{};
        g.A = A, g.W = W, g.C = C, g.R = R, g.bias = bias;
        g.M = (int)M, g.N = N, g.K = K, g.Cin = K, g.taps = 1, g.dil = 1, g.stride = 1, g.up = 1, g.nz1 = 1, g.nz2 = 1, g.act1 = act;
        g.alpha = 1.0f, g.lda = K, g.ldw = K, g.ldc = N, g.ldr = R ? D : 0, g.precision = 0;
        return cbx_gemm_f32(&g, stream);
    };
    int rc = 0;
    for (int i = 0; i < d->n_layers; ++i) {
        const cbx_gpt2_layer_t& L = d->layers[i];
        float* kc = d->kc + (long)i * d->kv_layer_stride;
        float* vc = d->vc + (long)i * d->kv_layer_stride;
        if ((rc = cbx_layernorm_f32(d->x, d->h, L.ln1_w, L.ln1_b, nullptr, M, D, D, D, d->eps, 0, CBX_ACT_NONE, 1.0f, stream))) return rc;
        if ((rc = linear(d->h, L.wqkv, L.bqkv, d->qkv, nullptr, 3 * D, D, CBX_ACT_NONE))) return rc;
        if ((rc = cbx_rope_kv_f32(d->qkv, d->positions, nullptr, nullptr, kc, vc, d->cache_rows, M, H, 3 * D, d->kv_row_stride, d->kv_head_stride, stream))) return rc;
        const long sb = (long)d->S * 3 * D, st = 3 * D;
        if (d->prefix == 0)
            rc = cbx_flash_attn_f32(d->qkv, d->qkv + D, d->qkv + 2 * D, d->att, nullptr, d->rows, H, d->S, d->S, sb, st, sb, st, sb, st, (long)d->S * D, D, d->attn_scale, 1, stream);
        else
            rc = cbx_flash_attn_kv_f32(d->qkv, kc, vc, d->att, nullptr, d->rows, H, d->S, d->prefix + d->S, sb, st, d->kv_row_stride, 64, d->kv_head_stride,
                                       d->kv_row_stride, 64, d->kv_head_stride, (long)d->S * D, D, d->attn_scale, 1, stream);
        if (rc) return rc;
        if ((rc = linear(d->att, L.wo, L.bo, d->x, d->x, D, D, CBX_ACT_NONE))) return rc;
        if ((rc = cbx_layernorm_f32(d->x, d->h, L.ln2_w, L.ln2_b, nullptr, M, D, D, D, d->eps, 0, CBX_ACT_NONE, 1.0f, stream))) return rc;
        if ((rc = linear(d->h, L.wfc, L.bfc, d->g, nullptr, F, D, CBX_ACT_GELU_TANH))) return rc;
        if ((rc = linear(d->g, L.wpr, L.bpr, d->x, d->x, D, F, CBX_ACT_NONE))) return rc;
    }
    return 0;
}

// ---- The token loop in C (include/cbx.h "handle-level entry points").  The graph is what chatterbox_amd/t3.py captures through torch.cuda.graph: one
// cbx_t3_decode_step.  On the SIMT emulator (tests/simt: no graph API) the steps are issued one by one.
#include <vector>
struct cbx_t3_loop {
    cbx_t3_step_t step;
    std::vector<cbx_t3_layer_t> layers;
    cbx_sampler_t sampler;
    bool has_sampler = false;
    std::vector<int> done_host;
#ifndef CBX_SIMT_EMU
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
#endif
};

extern "C" int cbx_t3_loop_create(const cbx_t3_step_t* step, void* stream, cbx_t3_loop_t** out) {
    CBX_REQUIRE(step && out && step->layers && step->n_layers > 0, "t3_loop_create: null argument");
    cbx_t3_loop* h = new cbx_t3_loop();
    h->step = *step;
    h->layers.assign(step->layers, step->layers + step->n_layers);
    h->step.layers = h->layers.data();
    if (step->sampler) {
        h->sampler = *step->sampler;
        h->step.sampler = &h->sampler;
        h->has_sampler = true;
        h->done_host.assign(h->sampler.B > 0 ? h->sampler.B : 1, 0);
    }
#ifndef CBX_SIMT_EMU
    // captured on a stream of the library's own: the caller's stream may be the legacy default stream, which cannot capture, and nothing executes during a
    // capture anyway -- the graph is LAUNCHED on the caller's stream (cbx_t3_loop_run)
    (void)stream;
    hipStream_t st = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) {
        if (st) hipStreamDestroy(st);
        delete h;
        return cbx_set_error((int)e, "t3_loop_create: hipStreamBeginCapture: %s", hipGetErrorString(e));
    }
    const int rc = cbx_t3_decode_step(&h->step, st);
    e = hipStreamEndCapture(st, &h->graph);  // (always ended, also after a failed step: the stream must leave capture mode)
    hipStreamDestroy(st);
    if (rc != 0 || e != hipSuccess || !h->graph) {
        if (h->graph) hipGraphDestroy(h->graph);
        delete h;
        return rc ? rc : cbx_set_error((int)e, "t3_loop_create: hipStreamEndCapture: %s", hipGetErrorString(e));
    }
    e = hipGraphInstantiate(&h->exec, h->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        hipGraphDestroy(h->graph);
        delete h;
        return cbx_set_error((int)e, "t3_loop_create: hipGraphInstantiate: %s", hipGetErrorString(e));
    }
#else
    (void)stream;
#endif
    *out = h;
    return 0;
}

extern "C" int cbx_t3_loop_run(cbx_t3_loop_t* h, int n_steps, int poll_every, void* stream, int* steps_run) {
    CBX_REQUIRE(h && n_steps >= 0 && poll_every >= 0, "t3_loop_run: bad arguments");
    CBX_REQUIRE(poll_every == 0 || h->has_sampler, "t3_loop_run: polling the done flags needs a sampler in the step descriptor");
    int ran = 0;
    for (int i = 0; i < n_steps; ++i) {
#ifndef CBX_SIMT_EMU
        const hipError_t e = hipGraphLaunch(h->exec, (hipStream_t)stream);
        if (e != hipSuccess) return cbx_set_error((int)e, "t3_loop_run: hipGraphLaunch: %s", hipGetErrorString(e));
#else
        const int rc = cbx_t3_decode_step(&h->step, stream);
        if (rc) return rc;
#endif
        ++ran;
        if (poll_every > 0 && ran % poll_every == 0 && i + 1 < n_steps) {  // the reference tests EOS on the host after every token (t3.py:366)
#ifndef CBX_SIMT_EMU
            hipError_t e2 = hipMemcpyAsync(h->done_host.data(), h->sampler.done, sizeof(int) * h->done_host.size(), hipMemcpyDeviceToHost, (hipStream_t)stream);
            if (e2 == hipSuccess) e2 = hipStreamSynchronize((hipStream_t)stream);
            if (e2 != hipSuccess) return cbx_set_error((int)e2, "t3_loop_run: fetching the done flags: %s", hipGetErrorString(e2));
#else
            for (size_t b = 0; b < h->done_host.size(); ++b) h->done_host[b] = h->sampler.done[b];
#endif
            bool all = true;
            for (int d : h->done_host) all = all && d != 0;
            if (all) break;
        }
    }
    if (steps_run) *steps_run = ran;
    return 0;
}

extern "C" int cbx_t3_loop_destroy(cbx_t3_loop_t* h) {
    if (!h) return 0;
#ifndef CBX_SIMT_EMU
    if (h->exec) hipGraphExecDestroy(h->exec);
    if (h->graph) hipGraphDestroy(h->graph);
#endif
    delete h;
    return 0;
}
