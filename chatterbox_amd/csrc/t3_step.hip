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
    CBX_REQUIRE(qks == 1 || (qks <= 4 && d->qkv_ct >= 1 && d->qkv_ct <= 4 && d->qkv_ssq && d->qkv_tile == 0 && !d->w_bf16),
                "t3_decode_step: qkv_ksplit 2 .. 4 needs qkv_ct 1 .. 4, qkv_ssq, the 16-column fp32 q/k/v image");
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
