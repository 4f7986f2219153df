// Stage-level C entry points of the S3Gen flow decoder and of the HiFT vocoder (SURVEY.md 8b "what the C-ABI replacement must export"):
//   cbx_cfm_solve   -- CausalConditionalCFM.solve_euler (reference models/s3gen/flow_matching.py:78-145, 196-233) around ConditionalDecoder.forward
//                      (models/s3gen/decoder.py:243-333) on plane-format operands: n_steps x [estimator, Euler + CFG update];
//   cbx_hift_decode -- HiFTGenerator.decode (models/s3gen/hifigan.py:412-444) between the STFT of the source and the iSTFT of conv_post.
// Like cbx_t3_decode_step / cbx_t3_prefill they only SEQUENCE the kernel-level entry points of this library, with the arguments
// chatterbox_amd/s3gen.py::FlowEngine.cfm / hift.py::HiFTEngine.decode pass one launch at a time (bit-identical results): caller-owned device memory,
// no allocation, no synchronisation, one stream, hipGraph-capturable.
#include "cbx_common.h"

namespace {

inline cbx_planes_t cols(const cbx_planes_t& P, long c0) { return cbx_planes_t{(char*)P.p + 2 * c0, P.ld, P.lo}; }
inline cbx_planes_t rows_from(const cbx_planes_t& P, long r0) { return cbx_planes_t{(char*)P.p + 2 * r0 * P.ld, P.ld, P.lo}; }

struct CfmCtx {
    const cbx_cfm_t* d;
    void* stream;
    int rows;
    long T, M;

    // F.conv1d (causal: pad_left = taps - 1) over every row group of T positions, plane operands (ops.conv1d_planes)
    int conv(const cbx_planes_t& src, const cbx_planes_t& w, int N, int cin, int taps, float* out, const cbx_planes_t* outp, const float* bias,
             const float* R) const {
        cbx_gemm_pl_t g{};
        g.A = src.p, g.W = w.p, g.C = out, g.P = outp ? outp->p : nullptr, g.bias = bias, g.R = R;
        g.M = (int)T, g.N = N, g.K = taps * cin, g.Cin = cin, g.taps = taps, g.dil = 1, g.stride = 1, g.pad_left = taps - 1, g.Tin = (int)T, g.nz1 = rows;
        g.act = CBX_ACT_NONE, g.act_slope = 0.0f, g.alpha = 1.0f;
        g.lda = src.ld, g.a_lo = src.lo, g.a_s1 = T * src.ld;
        g.ldw = w.ld, g.w_lo = w.lo;
        if (out) g.ldc = N, g.c_s1 = T * N;
        if (R) g.ldr = N, g.r_s1 = T * N;
        if (outp) g.ldp = outp->ld, g.p_lo = outp->lo, g.p_s1 = T * outp->ld;
        g.tile = d->gemm_tile;
        return cbx_gemm_planes(&g, stream);
    }
    // F.linear over all M rows as one batch (ops.linear_planes)
    int linear(const cbx_planes_t& x, const cbx_planes_t& w, int N, int K, float* out, const cbx_planes_t* outp, const float* bias, const float* R,
               int act, const float* ln_w = nullptr, const float* ln_b = nullptr, const cbx_planes_t* lnp = nullptr) const {
        cbx_gemm_pl_t g{};
        g.A = x.p, g.W = w.p, g.C = out, g.P = outp ? outp->p : nullptr, g.bias = bias, g.R = R;
        g.M = (int)M, g.N = N, g.K = K, g.Cin = K, g.taps = 1, g.dil = 1, g.stride = 1, g.nz1 = 1;
        g.act = act, g.alpha = 1.0f;
        g.lda = x.ld, g.a_lo = x.lo, g.ldw = w.ld, g.w_lo = w.lo;
        if (out) g.ldc = N;
        if (R) g.ldr = N;
        if (outp) g.ldp = outp->ld, g.p_lo = outp->lo;
        g.tile = d->gemm_tile;
        if (ln_w) g.ln_w = ln_w, g.ln_b = ln_b, g.LNP = lnp->p, g.ld_lnp = lnp->ld, g.lnp_lo = lnp->lo, g.ln_eps = 1e-5f;  // LayerNorm of the finished row (ops.linear_planes(ln=...))
        return cbx_gemm_planes(&g, stream);
    }
    int ln_planes(const float* x, const cbx_planes_t& out, const float* w, const float* b, const float* post_add, int act) const {
        return cbx_layernorm_planes_f32(x, out.p, w, b, post_add, M, 256, 256, out.ld, out.lo, 1e-5f, act, 1.0f, stream);
    }

    // CausalResnetBlock1D (decoder.py:65-98 via matcha ResnetBlock1D): inP (M x cin planes) -> d->x fp32
    int resnet(const cbx_cfm_stage_t& s, const cbx_planes_t& inP, const float* tb) const {
        int rc;
        if ((rc = conv(inP, s.c1, 256, s.cin, 3, d->ra, nullptr, s.c1_b, nullptr))) return rc;
        if ((rc = ln_planes(d->ra, d->aP, s.n1_w, s.n1_b, tb, CBX_ACT_MISH))) return rc;
        if ((rc = conv(d->aP, s.c2, 256, 256, 3, d->rb, nullptr, s.c2_b, nullptr))) return rc;
        if ((rc = cbx_layernorm_f32(d->rb, d->rb, s.n2_w, s.n2_b, nullptr, M, 256, 256, 256, 1e-5f, 0, CBX_ACT_MISH, 1.0f, stream))) return rc;
        return conv(inP, s.res, 256, s.cin, 1, d->x, nullptr, s.res_b, d->rb);
    }

    // BasicTransformerBlock (matcha/transformer.py:243-316, diffusers Attention / GELU): d->x updated in place, or -- the LAST block of a stage -- written
    // in plane format only (outP)
    // fused_ln: norm3 from the out-projection's epilogue, the NEXT block's norm1 (`next`) from ff2's; `pre_normed`: d->hP already holds norm1(x)
    int tblock(const cbx_cfm_tblock_t& t, const cbx_planes_t* outP, bool pre_normed, const cbx_cfm_tblock_t* next) const {
        int rc;
        float* x = d->x;
        if (!pre_normed && (rc = ln_planes(x, d->hP, t.n1_w, t.n1_b, nullptr, CBX_ACT_NONE))) return rc;
        if (d->fused_qkv && T % 4 == 0) {  // to_q | to_k | to_v as ONE Linear; the v columns are stored transposed per row group
            cbx_gemm_pl_t g{};
            g.A = d->hP.p, g.W = t.wqkv.p, g.P = d->qkP.p;
            g.M = (int)M, g.N = 1536, g.K = 256, g.Cin = 256, g.taps = 1, g.dil = 1, g.stride = 1, g.nz1 = 1, g.alpha = 1.0f;
            g.lda = d->hP.ld, g.a_lo = d->hP.lo, g.ldw = t.wqkv.ld, g.w_lo = t.wqkv.lo, g.ldp = d->qkP.ld, g.p_lo = d->qkP.lo;
            g.PT = d->vtP.p, g.pt_n0 = 1024, g.pt_T = (int)T, g.pt_ld = d->vtP.ld, g.pt_lo = d->vtP.lo, g.pt_zs = 512 * d->vtP.ld;
            g.tile = d->gemm_tile;
            if ((rc = cbx_gemm_planes(&g, stream))) return rc;
        } else {
            if ((rc = linear(d->hP, t.wqkv, 1024, 256, nullptr, &d->qkP, nullptr, nullptr, CBX_ACT_NONE))) return rc;
            cbx_gemm_pl_t g{};  // V^T[z] (512 x T) = W_v h[z]^T: the same products with the operands swapped
            const cbx_planes_t wv = rows_from(t.wqkv, 1024);
            g.A = wv.p, g.W = d->hP.p, g.P = d->vtP.p;
            g.M = 512, g.N = (int)T, g.K = 256, g.Cin = 256, g.taps = 1, g.dil = 1, g.stride = 1, g.nz1 = rows, g.alpha = 1.0f;
            g.lda = wv.ld, g.a_lo = wv.lo, g.ldw = d->hP.ld, g.w_lo = d->hP.lo, g.w_s1 = T * d->hP.ld;
            g.ldp = d->vtP.ld, g.p_lo = d->vtP.lo, g.p_s1 = 512 * d->vtP.ld;
            g.tile = d->gemm_tile;
            if ((rc = cbx_gemm_planes(&g, stream))) return rc;
        }
        const cbx_planes_t q = d->qkP, k = cols(d->qkP, 512);
        if ((rc = cbx_flash_attn_planes_v(q.p, k.p, d->vtP.p, d->attP.p, d->lens, rows, 8, (int)T, (int)T, T * q.ld, q.ld, q.lo, T * k.ld, k.ld, k.lo,
                                          512 * d->vtP.ld, d->vtP.ld, d->vtP.lo, T * d->attP.ld, d->attP.ld, d->attP.lo, 0.125f, 0, d->attn_version, stream)))
            return rc;
        if (d->fused_ln) {
            if ((rc = linear(d->attP, t.wo, 256, 512, x, nullptr, t.bo, x, CBX_ACT_NONE, t.n3_w, t.n3_b, &d->hP))) return rc;
        } else {
            if ((rc = linear(d->attP, t.wo, 256, 512, x, nullptr, t.bo, x, CBX_ACT_NONE))) return rc;
            if ((rc = ln_planes(x, d->hP, t.n3_w, t.n3_b, nullptr, CBX_ACT_NONE))) return rc;
        }
        if ((rc = linear(d->hP, t.w1, 1024, 256, nullptr, &d->ffP, t.b1, nullptr, CBX_ACT_GELU_ERF))) return rc;
        if (next) return linear(d->ffP, t.w2, 256, 1024, x, nullptr, t.b2, x, CBX_ACT_NONE, next->n1_w, next->n1_b, &d->hP);
        return linear(d->ffP, t.w2, 256, 1024, outP ? nullptr : x, outP, t.b2, x, CBX_ACT_NONE);
    }

    int block(int k, const cbx_planes_t& inP, const cbx_planes_t& outP, const float* tbias) const {
        const cbx_cfm_stage_t& s = d->stages[k];
        int rc = resnet(s, inP, tbias + (long)k * 256);
        for (int j = 0; !rc && j < s.n_tb; ++j) {
            const bool last = j == s.n_tb - 1;
            rc = tblock(s.tb[j], last ? &outP : nullptr, d->fused_ln >= 2 && j > 0, (d->fused_ln >= 2 && !last) ? &s.tb[j + 1] : nullptr);
        }
        return rc;
    }

    // ConditionalDecoder.forward (decoder.py:243-333): d->xinP (M x 320 planes) -> d->v (rows, T, 80) fp32
    int estimator(const float* tbias) const {
        const int n = d->n_stages, n_mid = n - 2;
        const cbx_planes_t skip = cols(d->catP, 256), xh = d->catP;  // [x | skip] of the up block: both halves are written in place by their producers
        int rc;
        if ((rc = block(0, d->xinP, skip, tbias))) return rc;
        cbx_planes_t cur = n_mid == 0 ? xh : d->yP;
        if ((rc = conv(skip, d->stages[0].tail, 256, 256, 3, nullptr, &cur, d->stages[0].tail_b, nullptr))) return rc;
        for (int k = 1; k <= n_mid; ++k) {
            const cbx_planes_t nxt = k == n_mid ? xh : d->xP;
            if ((rc = block(k, cur, nxt, tbias))) return rc;
            cur = nxt;
        }
        if ((rc = block(n - 1, d->catP, d->xP, tbias))) return rc;
        if ((rc = conv(d->xP, d->stages[n - 1].tail, 256, 256, 3, nullptr, &d->yP, d->stages[n - 1].tail_b, nullptr))) return rc;
        if ((rc = conv(d->yP, d->fin_c, 256, 256, 3, d->ra, nullptr, d->fin_c_b, nullptr))) return rc;
        if ((rc = ln_planes(d->ra, d->aP, d->fin_n_w, d->fin_n_b, nullptr, CBX_ACT_MISH))) return rc;
        return conv(d->aP, d->fin_proj, 80, 256, 1, d->v, nullptr, d->fin_proj_b, nullptr);
    }
};

}  // namespace

extern "C" int cbx_cfm_solve(const cbx_cfm_t* d, void* stream) {
    CBX_REQUIRE(d && d->stages && d->n_stages >= 2 && d->dt && d->tbias && d->xin && d->v && d->ra && d->rb && d->x, "cfm_solve: null descriptor field");
    CBX_REQUIRE(d->B >= 1 && d->T >= 1 && d->n_steps >= 1 && d->rows == (d->cfg ? 2 * d->B : d->B), "cfm_solve: rows must be B (no CFG) or 2 B (CFG)");
    CBX_REQUIRE(d->T % 2 == 0 && (long)d->rows * d->T > 32 && ((long)d->rows * d->T + 512) * 4096 < (1L << 31),
                "cfm_solve: the plane-format path needs an even T, rows * T > 32 and 31-bit operand offsets (rows * T = %ld)", (long)d->rows * d->T);
    for (int k = 0; k < d->n_stages; ++k) {
        const cbx_cfm_stage_t& s = d->stages[k];
        CBX_REQUIRE(s.n_tb >= 1 && s.tb && s.cin % 32 == 0, "cfm_solve: stage %d needs transformer blocks and cin %% 32 == 0", k);
        CBX_REQUIRE((s.tail.p != nullptr) == (k == 0 || k == d->n_stages - 1), "cfm_solve: the down and the up stage (only) end in a conv");
    }
    CfmCtx c{d, stream, d->rows, d->T, (long)d->rows * d->T};
    for (int k = 0; k < d->n_steps; ++k) {
        int rc;
        // the packed estimator input [x | mu | spk | cond]: mu / spk / cond were split once by the caller, x after every Euler step
        if (k && (rc = cbx_split_planes_f32(d->xin, d->xinP.p, c.M, 80, 320, d->xinP.ld, d->xinP.lo, stream))) return rc;
        if ((rc = c.estimator(d->tbias + (long)k * d->n_stages * 256))) return rc;
        if ((rc = cbx_cfm_euler_f32(d->xin, d->v, d->B, d->T, 80, 320, 80, d->T * 320, d->T * 80, d->dt[k], d->cfg_rate, d->cfg, stream))) return rc;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------ HiFT decode
namespace {

struct HiftCtx {
    const cbx_hift_t* d;
    void* stream;

    // Conv1d, channel-last, as ops.conv1d describes it to cbx_gemm_f32: x (B, Tin, ldx), w packed (N, taps * cin), out (B, Tout, ldo)
    struct Conv {
        const float *x, *w, *bias, *res = nullptr, *p1 = nullptr, *p2 = nullptr;
        float *out, *out2 = nullptr;
        const int* lens = nullptr;
        long Tin, Tout, ldx, xs, ldo, os;  // row / batch strides of x and of out (res, out2 share out's)
        int N, taps, cin, dil = 1, stride = 1, pad_left = 0, act1 = CBX_ACT_NONE, act2 = CBX_ACT_NONE;
        float slope1 = 0.0f, slope2 = 0.0f, alpha = 1.0f, beta = 0.0f;
    };
    int run(const Conv& c) const {
        cbx_gemm_t g{};
        g.A = c.x, g.W = c.w, g.C = c.out, g.bias = c.bias, g.R = c.res, g.C2 = c.out2, g.act1_param = c.p1, g.act2_param = c.p2, g.lens = c.lens;
        g.M = (int)c.Tout, g.N = c.N, g.K = c.taps * c.cin, g.Cin = c.cin, g.taps = c.taps, g.dil = c.dil, g.stride = c.stride, g.pad_left = c.pad_left;
        g.up = 1, g.Tin = (int)c.Tin, g.nz1 = d->B, g.nz2 = 1;
        g.act1 = c.act1, g.act2 = c.act2, g.act1_slope = c.slope1, g.act2_slope = c.slope2, g.alpha = c.alpha, g.beta = c.beta;
        g.lda = c.ldx, g.a_s1 = c.xs, g.ldw = (long)c.taps * c.cin, g.ldc = c.ldo, g.c_s1 = c.os;
        if (c.res) g.ldr = c.ldo, g.r_s1 = c.os;
        if (c.out2) g.ldc2 = c.ldo, g.c2_s1 = c.os;
        g.precision = d->precision;
        return cbx_gemm_f32(&g, stream);
    }

    // ResBlock.forward (hifigan.py:155-161): x (B, L, C) is not modified, a_first = snake(x, alpha1[0]); the last conv accumulates
    // out = beta out + alpha (xt + x) (and out2 = act2(out))
    int resblock(const cbx_hift_resblock_t& rb, int k, const float* x, const float* a_first, int C, long L, const int* lens, float* out, float alpha,
                 float beta, float* out2, float slope2) const {
        const float *cur_x = x, *cur_a = a_first;
        const int dils[3] = {1, 3, 5};
        for (int j = 0; j < 3; ++j) {
            Conv c;
            c.x = cur_a, c.w = rb.c1_w[j], c.bias = rb.c1_b[j], c.out = d->t1, c.lens = lens, c.Tin = c.Tout = L, c.ldx = c.ldo = C, c.xs = c.os = L * C;
            c.N = C, c.taps = k, c.cin = C, c.dil = dils[j], c.pad_left = (k * dils[j] - dils[j]) / 2, c.act1 = CBX_ACT_SNAKE, c.p1 = rb.a2[j];
            int rc = run(c);
            if (rc) return rc;
            Conv e;
            e.x = d->t1, e.w = rb.c2_w[j], e.bias = rb.c2_b[j], e.lens = lens, e.Tin = e.Tout = L, e.ldx = e.ldo = C, e.xs = e.os = L * C;
            e.N = C, e.taps = k, e.cin = C, e.pad_left = (k - 1) / 2, e.res = cur_x;
            if (j < 2) {
                float* nxt = cur_x != d->xa ? d->xa : d->xb;
                e.out = nxt, e.out2 = d->an, e.act2 = CBX_ACT_SNAKE, e.p2 = rb.a1[j + 1];
                if ((rc = run(e))) return rc;
                cur_x = nxt, cur_a = d->an;
            } else {
                e.out = out, e.alpha = alpha, e.beta = beta, e.out2 = out2, e.act2 = CBX_ACT_LRELU, e.slope2 = slope2;
                if ((rc = run(e))) return rc;
            }
        }
        return 0;
    }
};

}  // namespace

extern "C" int cbx_hift_decode(const cbx_hift_t* d, void* stream) {
    CBX_REQUIRE(d && d->mel && d->s && d->wav && d->spec && d->post && d->x0 && d->xs && d->t1 && d->xa && d->xb && d->an && d->si && d->sa && d->acc &&
                    d->a0 && d->nxt[0] && d->nxt[1],
                "hift_decode: null descriptor field");
    CBX_REQUIRE(d->B >= 1 && d->T >= 1, "hift_decode: bad shape");
    static const int UPS[3] = {8, 5, 3}, SRC_DOWN[3][3] = {{15, 30, 7}, {3, 6, 1}, {1, 1, 0}}, RB_K[3] = {3, 7, 11}, SRC_RB_K[3] = {7, 7, 11}, CS[3] = {256, 128, 64};
    HiftCtx h{d, stream};
    const int B = d->B;
    const long T = d->T, L3 = 120 * T + 1;
    const long LS[3] = {8 * T, 40 * T, L3};
    const int* ln[4] = {nullptr, nullptr, nullptr, nullptr};  // valid rows of the mel, of the three upsampled stages' outputs
    const int* len480 = nullptr;
    if (d->lens)
        for (int i = 0; i < 4; ++i) ln[i] = d->lens + (long)i * B;
    if (d->lens) len480 = d->lens + 4L * B;
    int rc;
    if ((rc = cbx_hift_stft_f32(d->s, d->spec, len480, B, 480 * T, 32, stream))) return rc;
    {
        HiftCtx::Conv c;  // conv_pre + leaky ReLU (hifigan.py:414-417)
        c.x = d->mel, c.w = d->conv_pre_w, c.bias = d->conv_pre_b, c.out = d->x0, c.lens = ln[0], c.Tin = c.Tout = T, c.ldx = 80, c.xs = T * 80, c.ldo = 512,
        c.os = T * 512, c.N = 512, c.taps = 7, c.cin = 80, c.pad_left = 3, c.act1 = CBX_ACT_LRELU, c.slope1 = 0.1f;
        if ((rc = h.run(c))) return rc;
    }
    const float* x = d->x0;
    long Tin = T;
    for (int i = 0; i < 3; ++i) {
        const int C = CS[i], st = UPS[i];
        const long L = LS[i];
        float* xs = d->xs;
        HiftCtx::Conv u;  // ConvTranspose1d as its phase-packed 3-tap form: the (Tin, st * C) output IS the (Tin * st, C) upsampled tensor
        u.x = x, u.w = d->ups_w[i], u.bias = d->ups_b[i], u.lens = ln[i], u.Tin = u.Tout = Tin, u.ldx = 2 * C, u.xs = Tin * 2 * C, u.ldo = (long)st * C, u.os = L * C;
        u.N = st * C, u.taps = 3, u.cin = 2 * C, u.pad_left = 1;
        u.out = i < 2 ? xs : xs + C;  // ReflectionPad1d((1, 0)) (hifigan.py:421-422): the conv output goes to rows 1 .., row 0 := row 2
        if ((rc = h.run(u))) return rc;
        if (i == 2 && (rc = cbx_axpby_f32(xs + 2 * C, xs, B, C, L * C, L * C, 1.0f, 0.0f, stream))) return rc;
        // fusion: x += source_resblock(source_down(s_stft)) (hifigan.py:424-427)
        HiftCtx::Conv sd;
        sd.x = d->spec, sd.w = d->src_down_w[i], sd.bias = d->src_down_b[i], sd.lens = ln[3], sd.Tin = L3, sd.Tout = L, sd.ldx = 32, sd.xs = L3 * 32, sd.ldo = C,
        sd.os = L * C, sd.N = C, sd.taps = SRC_DOWN[i][1], sd.cin = 32, sd.stride = SRC_DOWN[i][0], sd.pad_left = SRC_DOWN[i][2];
        sd.out = d->si, sd.out2 = d->sa, sd.act2 = CBX_ACT_SNAKE, sd.p2 = d->src_rb[i].a1[0];
        if ((rc = h.run(sd))) return rc;
        if ((rc = h.resblock(d->src_rb[i], SRC_RB_K[i], d->si, d->sa, C, L, ln[i + 1], xs, 1.0f, 1.0f, nullptr, 0.0f))) return rc;
        // mean of the three ResBlocks; the last one also emits the leaky ReLU the next stage consumes
        float* nxt = d->nxt[i & 1];
        for (int j = 0; j < 3; ++j) {
            const cbx_hift_resblock_t& rb = d->rb[i * 3 + j];
            if ((rc = cbx_act_f32(xs, d->a0, rb.a1[0], (long)B * L, C, C, C, CBX_ACT_SNAKE, 0.0f, stream))) return rc;
            if ((rc = h.resblock(rb, RB_K[j], xs, d->a0, C, L, ln[i + 1], d->acc, (float)(1.0 / 3), j == 0 ? 0.0f : 1.0f, j == 2 ? nxt : nullptr,
                                 i < 2 ? 0.1f : 0.01f)))
                return rc;
        }
        x = nxt, Tin = L;
    }
    if (hipMemsetAsync(d->post, 0, sizeof(float) * B * L3 * 32, (hipStream_t)stream) != hipSuccess) return cbx_set_error(CBX_EINVAL, "hift_decode: hipMemsetAsync failed");
    HiftCtx::Conv p;
    p.x = x, p.w = d->conv_post_w, p.bias = d->conv_post_b, p.out = d->post, p.lens = ln[3], p.Tin = p.Tout = L3, p.ldx = 64, p.xs = L3 * 64, p.ldo = 32, p.os = L3 * 32;
    p.N = 18, p.taps = 7, p.cin = 64, p.pad_left = 3;
    if ((rc = h.run(p))) return rc;
    return cbx_hift_istft_f32(d->post, d->wav, B, L3, 32, 0.99f, d->fade ? 480 : 0, stream);
}

// ------------------------------------------------------------------------------------------------------------------ S3Gen token encoder
namespace {

struct EncCtx {
    const cbx_s3enc_t* d;
    void* stream;

    // F.linear as ops.linear describes it to cbx_gemm_f32
    int linear(const float* x, const float* w, float* out, const float* bias, const float* R, long M, int N, int K, int act) const {
        cbx_gemm_t g{};
        g.A = x, g.W = w, g.C = out, g.bias = bias, g.R = R;
        g.M = (int)M, g.N = N, g.K = K, g.Cin = K, g.taps = 1, g.dil = 1, g.stride = 1, g.up = 1, g.nz1 = 1, g.nz2 = 1;
        g.act1 = act, g.alpha = 1.0f, g.lda = K, g.ldw = K, g.ldc = N, g.ldr = R ? N : 0, g.precision = d->precision;
        return cbx_gemm_f32(&g, stream);
    }
    // Conv1d over (B, Tin, 512) -> (B, Tout, 512), channel-last (ops.conv1d)
    int conv(const float* x, const float* w, float* out, const float* bias, const float* R, const int* lens, long Tin, long Tout, int taps, int pad_left, int up,
             int act, float slope) const {
        cbx_gemm_t g{};
        g.A = x, g.W = w, g.C = out, g.bias = bias, g.R = R, g.lens = lens;
        g.M = (int)Tout, g.N = 512, g.K = taps * 512, g.Cin = 512, g.taps = taps, g.dil = 1, g.stride = 1, g.pad_left = pad_left, g.up = up, g.Tin = (int)Tin;
        g.nz1 = d->B, g.nz2 = 1, g.act1 = act, g.act1_slope = slope, g.alpha = 1.0f;
        g.lda = 512, g.a_s1 = Tin * 512, g.ldw = (long)taps * 512, g.ldc = 512, g.c_s1 = Tout * 512;
        if (R) g.ldr = 512, g.r_s1 = Tout * 512;
        g.precision = d->precision;
        return cbx_gemm_f32(&g, stream);
    }
    int ln(const float* x, float* y, const float* w, const float* b, long rows, float eps, float scale) const {
        return cbx_layernorm_f32(x, y, w, b, nullptr, rows, 512, 512, 512, eps, 0, CBX_ACT_NONE, scale, stream);
    }
    // LinearNoSubsampling / the up_embed twin: Linear -> LayerNorm -> * sqrt(d_model) (the positional encoding's xscale)
    int embed(const float* xin, float* y, const float* w, const float* b, const float* lnw, const float* lnb, long M) const {
        int rc = linear(xin, w, y, b, nullptr, M, 512, 512, CBX_ACT_NONE);
        return rc ? rc : ln(y, y, lnw, lnb, M, 1e-5f, (float)sqrt(512.0));
    }
    int conformer(const cbx_conformer_t& L, float* x, long T, const float* pe, const int* lens) const {
        const long M = (long)d->B * T;
        int rc;
        if ((rc = ln(x, d->h, L.ln_mha_w, L.ln_mha_b, M, 1e-12f, 1.0f))) return rc;
        if ((rc = linear(d->h, L.w4, d->q4, L.b4, nullptr, M, 2048, 512, CBX_ACT_NONE))) return rc;
        if ((rc = linear(pe, L.wpos, d->pp, nullptr, nullptr, 2 * T - 1, 512, 512, CBX_ACT_NONE))) return rc;
        if ((rc = cbx_flash_relpos_f32(d->q4, d->q4 + 512, d->q4 + 1024, d->q4 + 1536, d->pp, d->att, lens, d->B, 8, (int)T, T * 2048, 2048, 512, T * 512, 512,
                                       0.125f, stream)))
            return rc;
        if ((rc = linear(d->att, L.wo, x, L.bo, x, M, 512, 512, CBX_ACT_NONE))) return rc;
        if ((rc = ln(x, d->h, L.ln_ff_w, L.ln_ff_b, M, 1e-12f, 1.0f))) return rc;
        if ((rc = linear(d->h, L.w1, d->ff, L.b1, nullptr, M, 2048, 512, CBX_ACT_SILU))) return rc;
        return linear(d->ff, L.w2, x, L.b2, x, M, 512, 2048, CBX_ACT_NONE);
    }
};

}  // namespace

extern "C" int cbx_s3gen_encode(const cbx_s3enc_t* d, void* stream) {
    CBX_REQUIRE(d && d->ids && d->lens && d->lens2 && d->emb && d->pe && d->pe2 && d->mu && d->x0 && d->xa && d->y1 && d->x2 && d->xu && d->xb && d->h && d->q4 &&
                    d->pp && d->att && d->ff && (d->enc || !d->n_enc) && (d->up_enc || !d->n_up),
                "s3gen_encode: null descriptor field");
    CBX_REQUIRE(d->B >= 1 && d->N >= 1 && d->n_enc >= 0 && d->n_up >= 0, "s3gen_encode: bad shape");
    EncCtx c{d, stream};
    const long N = d->N, T2 = 2 * N, MN = (long)d->B * N, M2 = (long)d->B * T2;
    int rc;
    if ((rc = cbx_embed_f32(d->ids, d->emb, nullptr, nullptr, d->x0, MN, 512, 512, 1.0f, 1, stream))) return rc;
    if ((rc = c.embed(d->x0, d->xa, d->e_w, d->e_b, d->e_lnw, d->e_lnb, MN))) return rc;
    // PreLookaheadLayer (upsample_encoder.py:81-96): right-pad 3 conv k4 -> leaky_relu -> left-pad 2 conv k3 -> + x
    if ((rc = c.conv(d->xa, d->pl1_w, d->y1, d->pl1_b, nullptr, d->lens, N, N, 4, 0, 1, CBX_ACT_LRELU, 0.01f))) return rc;
    if ((rc = c.conv(d->y1, d->pl2_w, d->x2, d->pl2_b, d->xa, nullptr, N, N, 3, 2, 1, CBX_ACT_NONE, 0.0f))) return rc;
    for (int i = 0; i < d->n_enc; ++i)
        if ((rc = c.conformer(d->enc[i], d->x2, N, d->pe, d->lens))) return rc;
    // Upsample1D (upsample_encoder.py:59-63): nearest x2, left-pad 4, conv k5 -- fused in the A-operand address map
    if ((rc = c.conv(d->x2, d->up_w, d->xu, d->up_b, nullptr, nullptr, N, T2, 5, 4, 2, CBX_ACT_NONE, 0.0f))) return rc;
    if ((rc = c.embed(d->xu, d->xb, d->u_w, d->u_b, d->u_lnw, d->u_lnb, M2))) return rc;
    for (int i = 0; i < d->n_up; ++i)
        if ((rc = c.conformer(d->up_enc[i], d->xb, T2, d->pe2, d->lens2))) return rc;
    if ((rc = c.ln(d->xb, d->xb, d->after_w, d->after_b, M2, 1e-5f, 1.0f))) return rc;
    cbx_gemm_t g{};  // encoder_proj (flow.py:168-169)
    g.A = d->xb, g.W = d->proj_w, g.C = d->mu, g.bias = d->proj_b;
    g.M = (int)M2, g.N = 80, g.K = 512, g.Cin = 512, g.taps = 1, g.dil = 1, g.stride = 1, g.up = 1, g.nz1 = 1, g.nz2 = 1, g.alpha = 1.0f;
    g.lda = 512, g.ldw = 512, g.ldc = 80, g.precision = d->precision;
    return cbx_gemm_f32(&g, stream);
}

// ------------------------------------------------------------------------------------------------------------------ HiFT F0 predictor + source
extern "C" int cbx_hift_f0_source(const cbx_hift_f0_t* d, void* stream) {
    CBX_REQUIRE(d && d->mel && d->cls_w && d->cls_b && d->src_w && d->phase && d->noise && d->buf0 && d->buf1 && d->f0 && d->s && d->cum,
                "hift_f0_source: null descriptor field");
    CBX_REQUIRE(d->B >= 1 && d->T >= 1, "hift_f0_source: bad shape");
    const long T = d->T;
    const float* x = d->mel;
    int cin = 80, rc;
    float* bufs[2] = {d->buf0, d->buf1};
    for (int n = 0; n < 5; ++n) {
        CBX_REQUIRE(d->f0_w[n] && d->f0_b[n], "hift_f0_source: null condnet weight %d", n);
        cbx_gemm_t g{};
        g.A = x, g.W = d->f0_w[n], g.C = bufs[n & 1], g.bias = d->f0_b[n], g.lens = d->lens;
        g.M = (int)T, g.N = 512, g.K = 3 * cin, g.Cin = cin, g.taps = 3, g.dil = 1, g.stride = 1, g.pad_left = 1, g.up = 1, g.Tin = (int)T, g.nz1 = d->B, g.nz2 = 1;
        g.act1 = CBX_ACT_ELU, g.alpha = 1.0f, g.lda = cin, g.a_s1 = T * cin, g.ldw = 3L * cin, g.ldc = 512, g.c_s1 = T * 512, g.precision = 1;
        if ((rc = cbx_gemm_f32(&g, stream))) return rc;
        x = bufs[n & 1], cin = 512;
    }
    cbx_gemm_t g{};  // classifier + abs
    g.A = x, g.W = d->cls_w, g.C = d->f0, g.bias = d->cls_b;
    g.M = (int)(d->B * T), g.N = 1, g.K = 512, g.Cin = 512, g.taps = 1, g.dil = 1, g.stride = 1, g.up = 1, g.nz1 = 1, g.nz2 = 1;
    g.act1 = CBX_ACT_ABS, g.alpha = 1.0f, g.lda = 512, g.ldw = 512, g.ldc = 1, g.precision = 1;
    if ((rc = cbx_gemm_f32(&g, stream))) return rc;
    return cbx_hift_source_f32(d->f0, d->phase, d->noise, d->src_w, d->src_b, d->s, d->cum, d->B, (int)T, 480, 24000.0f, stream);
}
