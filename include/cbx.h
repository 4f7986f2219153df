/* cbx.h -- C ABI of libcbx_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the Chatterbox hot path
 * generate() = T3.inference -> S3Gen.flow_inference -> HiFT.inference.
 *
 * The reference (resemble-ai/chatterbox) has no FFI: its numerical layer is PyTorch ATen + HF transformers +
 * diffusers called from Python.  Each entry point below replaces the ATen/HF op group that the cited reference
 * lines dispatch; INTEGRATION.md shows the ctypes binding.  Conventions:
 *   - plain C types only; every pointer is a DEVICE pointer valid on `stream` (a hipStream_t passed as void*);
 *   - all tensors fp32 row-major "channel-last" (rows = time/tokens, columns = channels) unless stated;
 *   - the callee never allocates, never synchronises, and is hipGraph-capturable;
 *   - return 0 on success, else a negative CBX_E* / positive hipError_t; cbx_last_error() gives text.
 */
#ifndef CBX_H
#define CBX_H
#ifdef __cplusplus
extern "C" {
#endif

#define CBX_ABI_VERSION 16 /* 2: cbx_gemm_t.precision, cbx_flash_attn_split_f32; 4: packed decode GEMV operands, RMSNorm / residual folded into cbx_gemv_f32;
                              5: precision 16 (f16x3) + cbx_set_range_flag; 6: LayerNorm folded into the GEMM A operand (ln_stats / ln_w / ln_b, cbx_row_stats_f32);
                              7: plane-format operands (cbx_gemm_planes, cbx_split_planes_f32, cbx_layernorm_planes_f32, cbx_flash_attn_split_po),
                                 per-device range flag, cbx_gemm_ln_fusable; 8: cbx_gemm_pl_t.PT (transposed column range), cbx_set_decode_attn_workspace;
                              9: 12- and 4-column decode GEMV tiles (cbx_gemv_t.half_tile = 12 / 4), cbx_t3_step_t.qkv_tile, d_ksplit = 1;
                              10: no process-wide state on the decode path -- cbx_decode_attn_t / cbx_decode_attn_rope (geometry and the split-context
                                  workspace per call), cbx_gemv_t.flags, cbx_t3_step_t.da_* / gemv_flags; the cbx_set_* setters remain as TEST HOOKS of
                                  the positional entry points only;
                              11: cbx_gemv_t.col_tiles / ssq_out (column-tile / split-K form of the RMSNorm-folded decode GEMV), cbx_decode_attn_t.qkv_nparts /
                                  qkv_part_stride / qkv_ssq / rms_dim / rms_eps (the attention adds the q/k/v partial sums and applies rstd),
                                  cbx_t3_step_t.qkv_ksplit / qkv_ct / head_ct / qkv_ssq, cbx_t3_prefill;
                              12: stage-level seams of the flow and the vocoder: cbx_planes_t, cbx_s3gen_encode, cbx_cfm_solve, cbx_hift_f0_source, cbx_hift_decode;
                              13: per-call launch geometry of the plane-format kernels (cbx_gemm_pl_t.tile, cbx_flash_attn_planes_v, cbx_cfm_t.gemm_tile /
                                  attn_version: no process-wide state on the flow path either) incl. the CO-RESIDENT forms of the throughput schedule
                                  (one workgroup per CU that leaves half of the register file and 64 KiB of LDS to another stream), CBX_GEMV_SHALLOW;
                              14: the batch-1 decode path of the GPT-2 backbones (Turbo / Nano): cbx_gemv_row_f32, cbx_decode_attn_parts; the token loop in C: cbx_t3_loop_*;
                              15: cbx_flash_attn_kv_f32 (K / V head strides: attention over the KV cache in the prefill);
                              16: cbx_gpt2_prefill (the prefill of the GPT-2 backbones as one call, optionally behind a cached conditioning prefix) */
#define CBX_EINVAL (-22)

/* activations usable in GEMM / elementwise epilogues */
enum { CBX_ACT_NONE = 0, CBX_ACT_SILU = 1, CBX_ACT_GELU_ERF = 2, CBX_ACT_GELU_TANH = 3, CBX_ACT_MISH = 4,
       CBX_ACT_LRELU = 5, CBX_ACT_ELU = 6, CBX_ACT_TANH = 7, CBX_ACT_SNAKE = 8, CBX_ACT_ABS = 9 };

int cbx_abi_version(void);
const char* cbx_last_error(void);

/* ---- implicit-GEMM linear / conv1d / batched matmul on fp32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32) ----
 * C[z][m][n] = beta*C + alpha*( act1( sum_{tap,c} A[z][ (m*stride + tap*dil - pad_left)/up ][c] * W[z][n][tap*Cin+c]
 *                                      + bias[n] ) + R[z][m][n] ),      C2 = act2(C)   (optional)
 * Replaces F.linear / F.conv1d / F.conv_transpose1d (phase-packed weights) / torch.matmul on:
 *   HF LlamaAttention/LlamaMLP projections (models/t3/t3.py:326-333,378-384), CFM estimator convs and
 *   transformer-block linears (models/s3gen/decoder.py:243-333, matcha/transformer.py:243-316), conformer
 *   encoder linears/convs (transformer/upsample_encoder.py:237-304), HiFT convs (hifigan.py:412-444).
 * z = z1*nz2 + z2 (two-level batch: e.g. utterance row, attention head).
 * K is walked in float4 units: K % 4 == 0, except with w_kn where A only has to be readable and finite up to the next
 * multiple of 4 (W rows >= K read as zero).
 */
typedef struct cbx_gemm_t {
    const float* A; const float* W; float* C;
    const float* bias;        /* [N] or NULL */
    const float* R;           /* residual, or NULL */
    float* C2;                /* second output act2(C), or NULL */
    const float* act1_param;  /* per-column parameter of act1 (snake alpha) or NULL */
    const float* act2_param;
    const int* lens;          /* per-z1 number of valid INPUT rows (rows >= lens read as 0), or NULL */
    int M, N, K;              /* output rows per batch, output columns, K = taps*Cin */
    int Cin, taps, dil, stride, pad_left, up, Tin;
    int nz1, nz2;
    int w_kn;                 /* 0: W is [N][K] (torch Linear layout); 1: W is [K][N] (e.g. P @ V) */
    int swiglu;               /* 1: packed [32 gate | 32 up] column groups -> C[m][N/2] = silu(gate)*up */
    int act1, act2;
    float act1_slope, act2_slope, alpha, beta;
    long lda, a_s1, a_s2;     /* strides in floats */
    long ldw, w_s1, w_s2;
    long ldc, c_s1, c_s2;
    long ldr, r_s1, r_s2;
    long ldc2, c2_s1, c2_s2;
    int precision;            /* 0 = exact; 1 = exact fp32 MFMA
                                 (bitwise an fmaf chain); 3 = "bf16x3", 6 = "bf16x6": every fp32 operand split into 2 / 3
                                 bf16 planes and the product rebuilt from 3 / 6 bf16-MFMA plane products with fp32
                                 accumulation (rel. error ~4e-6 / ~1e-7; fp32 MFMA ~2.5e-7); 16 = "f16x3": two fp16 planes
                                 a = h + l/2048, three fp16-MFMA products in two fp32 accumulators (rel. error ~1e-7 at the cost of
                                 bf16x3; operands must satisfy |a| <= 65504, see cbx_set_range_flag).  Shapes the split kernel does
                                 not serve (w_kn, swiglu, M <= 32, conv Cin % 32 != 0) run exact. */
    int reserved0;
    /* ABI v6: LayerNorm folded into the A operand of a Linear (precision 16, taps == 1, nz == 1, no lens):
     * A'[m][k] = (A[m][k] - mean[m]) * rstd[m] * ln_w[k] + ln_b[k] is applied on the way to the matrix cores, so the standalone
     * F.layer_norm pass (one read + one write of the activation) becomes a statistics pass (cbx_row_stats_f32: one read). */
    const float* ln_stats;    /* [M][2] = {mean, rstd} per row, or NULL */
    const float* ln_w;        /* [K] */
    const float* ln_b;        /* [K] */
} cbx_gemm_t;
int cbx_gemm_f32(const cbx_gemm_t* p, void* stream);
/* Device word that every precision-16 launch (GEMM, flash attention, plane-format producers) ORs a 1 into when it meets an operand outside
 * the fp16 range (its result is then not meaningful and the caller repeats the computation at precision 6, which has the fp32 exponent
 * range).  NULL (the default) = not reported.  One word PER DEVICE: the call registers the word for the calling thread's current device and
 * a launch reports into the word of the device it runs on; the word must stay allocated while such launches are in flight. */
int cbx_set_range_flag(int* dev_flag);
/* stats[r] = {mean, rstd = 1/sqrt(var + eps)} of row r of x (C = 256: the CFM transformer blocks), computed exactly as cbx_layernorm_f32
 * computes them; feeds cbx_gemm_t.ln_stats.  Replaces the statistics half of F.layer_norm (matcha/transformer.py:243-316 norm1 / norm3). */
int cbx_row_stats_f32(const float* x, float* stats, long rows, int C, long ldx, float eps, void* stream);

/* Would cbx_gemm_f32 accept ln_stats for a precision-16 Linear with M rows, K columns and row stride lda (floats)?  0 when a tuning knob
 * (cbx_set_split_tile) or a > 2 GiB operand rules the folded form out: the caller then runs
 * cbx_layernorm_f32 + a plain Linear (F.layer_norm + F.linear, matcha/transformer.py:243-316). */
int cbx_gemm_ln_fusable(long M, int K, long lda);

/* ---- PLANE-FORMAT operands (ABI v7): the f16x3 arithmetic without per-tile conversion ----
 * A "planes" tensor stores an fp32 tensor X as two fp16 planes X = h + l / 2048, h = RNE16(X), l = RNE16(2048 (X - h)): element (row, c)
 * of plane h at base[row*ld + c], of plane l at base[row*ld + lo + c] (halves) -- the same 4 bytes per element as fp32, 22 significand
 * bits (the f16x3 mode of cbx_gemm_t.precision; |X| <= 65504, producers raise the cbx_set_range_flag word otherwise).
 * Constant weights are split once at load, activations are WRITTEN in plane format by their producer (the epilogues below), and the
 * consumer's K loop is a plain fp16 MFMA loop fed by direct global -> LDS loads.
 *
 * cbx_gemm_planes: C[z][m][n] and / or P[z][m][n] = act( sum_{tap,c} A[z][m*stride + tap*dil - pad_left][c] * W[n][tap*Cin + c] + bias[n] )
 *                                                      + R[z][m][n]      (act: none, GELU (erf) or SiLU; act_param / alpha reserved: NULL / 0 or 1)
 * A, W planes (W in [N][K] layout); C fp32 and / or P planes (either may be NULL); R fp32 residual.  Rows outside [0, min(Tin, lens[z]))
 * read as zero.  K % 32 == 0, Cin % 32 == 0.  Replaces the F.linear / F.conv1d calls of the CFM estimator: CausalConv1d / ResnetBlock1D
 * (models/s3gen/decoder.py:49-98, matcha/decoder.py:56-61), BasicTransformerBlock projections and FeedForward (matcha/transformer.py:243-316),
 * final_proj (decoder.py:331-333). */
typedef struct cbx_gemm_pl_t {
    const void* A; const void* W;   /* fp16 planes */
    float* C;                       /* fp32 output or NULL */
    void* P;                        /* planes output or NULL */
    const float* bias;              /* [N] or NULL */
    const float* R;                 /* fp32 residual or NULL */
    const float* act_param;         /* per-column activation parameter or NULL */
    const int* lens;                /* per-z number of valid INPUT rows or NULL */
    int M, N, K;                    /* output rows per batch, columns, K = taps * Cin */
    int Cin, taps, dil, stride, pad_left, Tin, nz1;
    int act; float act_slope, alpha;
    long lda, a_lo, a_s1;           /* halves */
    long ldw, w_lo, w_s1;           /* halves; w_s1 = per-batch stride of W (0: shared) */
    long ldc, c_s1;                 /* floats */
    long ldr, r_s1;                 /* floats */
    long ldp, p_lo, p_s1;           /* halves */
    int reserved0;                  /* 0 (diagnostic switches of a -DCBX_DIAG build) */
    /* ABI v8 -- transposed plane output for a column range (the V^T operand of cbx_flash_attn_planes produced by the SAME launch as the
     * q | k projection: to_q / to_k / to_v of matcha/transformer.py:243-316 as ONE Linear over [Wq; Wk; Wv]).  PT != NULL: output columns
     * n >= pt_n0 are not written to C / P; element (m, n) goes to PT[(m / pt_T) * pt_zs + (n - pt_n0) * pt_ld + (m % pt_T)] (h plane; the
     * l plane pt_lo halves further), i.e. one (N - pt_n0) x pt_T matrix per group of pt_T rows.  Needs P for the columns below pt_n0,
     * pt_n0 % 256 == 0, pt_T % 4 == 0, M % 4 == 0, no C / R / activation, nz1 == 1. */
    void* PT; int pt_n0, pt_T; long pt_ld, pt_lo, pt_zs;   /* halves */
    /* ABI v13 -- launch geometry of THIS call: 0 = the library's measured choice (or the cbx_set_planes_tile test hook); n > 0 = form n of the tile menu
     * (gemm_planes.hip); CBX_PL_TILE_CORESIDENT (-1) = the measured choice among the forms that leave room on the CU: ONE workgroup of 8 waves and <= 120
     * VGPRs per CU (form 8 where K % 64 == 0: 128 KiB of LDS, 32 KiB left; form 17 otherwise: 96 KiB), so that the workgroups of a latency-bound kernel chain on ANOTHER stream (the T3 decode step of the next
     * batch) stay co-resident instead of waiting for these to retire (profiles/r05_overlap_*); round 6: the two wide Linears of a transformer block (ff1 + GELU, q | k | v) keep
     * their loader-wave forms even then (measured: profiles/r06_{s,t,v}_*_ab.log).  Forms 41 / 42 (round 6) = forms 32 / 35 with the DEFERRED epilogue: a finished tile is folded
     * and the rest of its epilogue is issued between the MFMA groups of the workgroup's next tile.  Same arithmetic in every form. */
    int tile;
    /* ABI v13 -- LayerNorm of the FINISHED row, produced by the epilogue (N == 256 only: the attention out-projection, ff2 and the 1 x 1 residual conv of the CFM
     * estimator, whose consumer is nn.LayerNorm -> Linear, matcha/transformer.py:243-316 norm3 / the next block's norm1).  ln_w != NULL: besides C (the fp32
     * row incl. bias and residual) the launch writes LNP[m][n] = (C[m][n] - mean_m) * rstd_m * ln_w[n] + ln_b[n] in plane format (row stride ld_lnp, plane
     * offset lnp_lo, batch stride lnp_s1, halves), mean / variance over the 256 columns in fp32 (two passes, as cbx_layernorm_f32), eps ln_eps.  It replaces the
     * cbx_layernorm_planes_f32 launch between the two Linears.  Needs N == 256, K % 32 == 0, no activation, no P / PT; runs on the row-spanning 64 x 256 tile. */
    const float* ln_w; const float* ln_b; void* LNP; long ld_lnp, lnp_lo, lnp_s1; float ln_eps;
} cbx_gemm_pl_t;
#define CBX_PL_TILE_CORESIDENT (-1)
/* ABI v13: a stream ATTRIBUTE (like its priority): launches on a co-resident stream that would otherwise put several workgroups of one kernel on a CU
 * (LayerNorm, the split-operand GEMM of the encoder / vocoder) reserve enough LDS that at most one or two fit, so half of every SIMD's register file stays free
 * for another stream's kernels.  No effect on results.  Mark a stream once, before using it (ChatterboxEngine marks its flow + vocoder stream). */
int cbx_set_stream_coresident(void* stream, int on);
int cbx_gemm_planes(const cbx_gemm_pl_t* p, void* stream);
/* tuning knob: tile shape of cbx_gemm_planes (0 = automatic; see gemm_planes.hip) */
int cbx_set_planes_tile(int t);
/* tuning knob: 1 (default) = persistent workgroups walking several output tiles each (DMA runs across tile boundaries), 0 = one tile per workgroup */
int cbx_set_planes_persist(int on);
/* x (rows, C) fp32 -> planes (weights at load; estimator inputs).  C, ldx, ldp, p_lo multiples of 4. */
int cbx_split_planes_f32(const float* x, void* planes, long rows, int C, long ldx, long ldp, long p_lo, void* stream);
/* cbx_layernorm_f32 (C = 256, LayerNorm form) whose result is written in plane format: nn.LayerNorm (+ Mish + time bias) feeding a conv /
 * Linear (decoder.py:49-63, matcha/transformer.py:243-316 norm1 / norm3). */
int cbx_layernorm_planes_f32(const float* x, void* planes, const float* w, const float* b, const float* post_add, long rows, int C,
                             long ldx, long ldp, long p_lo, float eps, int act, float out_scale, void* stream);

/* ---- skinny-M weight-streaming GEMM for decode (M = 2*B rows <= 64), HBM-roofline kernel ----
 * out[ks][m][n] = sum_{k in slice ks} x[m][k] * W[n][k]  (+ bias on slice 0);  ksplit > 1 leaves partial sums that
 * cbx_add_rmsnorm_f32 reduces in fixed order.  swiglu: W is the packed [32 gate | 32 up] image, N = #features,
 * out[m][f] = silu(gate_f) * up_f.  Replaces the q_len == 1 HF Llama projections + speech head (t3.py:378-386). */
typedef struct cbx_gemv_t {
    const float* x; const float* W; const float* bias; float* out;
    int M, N, K;
    int ksplit;      /* K slices across workgroups (grid.y) */
    int nw;          /* waves per workgroup sharing one 16-column tile: 4 or 8 */
    int swiglu;
    int act;         /* CBX_ACT_* applied after the bias (ksplit == 1 only): gelu_new of GPT-2's c_fc */
    long ldx, ldw, ldo, part_stride;
    /* ABI v4 */
    int w_packed;    /* W is the lane-ordered packed image written by cbx_pack_gemv_weight_f32 (ldw ignored; swiglu: tiles
                        2f = gate, 2f+1 = up of feature tile f) */
    int x_packed;    /* x is in the same packed layout (rows padded to 16, ldx ignored); needs w_packed */
    int half_tile;   /* (was reserved0) narrow output tiles: 0 = 16 columns per workgroup; 1 or 8 = 8 columns (W is the 8-row-tile packed
                        image, cbx_pack_gemv_weight_f32 with swiglu = 8): twice the workgroups for projections with few output tiles
                        (N = 1024: 128 instead of 64); 12 / 4 = 12 / 4 columns (images packed with swiglu = 12 / 4): N = 3072 resp.
                        N = 1024 on exactly 256 workgroups.  Same per-column arithmetic as the 16-column form (bit-identical results) */
    int out_packed;  /* out (and res) use the packed operand layout of the CONSUMING gemv (its K = this N; N % 32 == 0); with ksplit > 1
                        the partial images are part_stride floats apart */
    const float* norm_w; /* [K] or NULL: LlamaRMSNorm(x) folded in (x * norm_w feeds the MFMAs, rstd applied in the epilogue);
                            needs w_packed, x_packed, ksplit == 1 */
    const float* res;    /* or NULL: out = res + x W^T, res in the same layout as out (in place allowed); ksplit == 1 */
    float eps;           /* RMSNorm epsilon */
    int n_xpart;         /* 0, 2 or 4: the x operand is x + sum_j xpart[j] (split-K partial images of the producing projection, packed
                            layout, reduced in fixed order on the way to the MFMA); needs norm_w, M <= 16, nw == 8 */
    const float* xpart;  /* [n_xpart] images, xpart_stride floats apart */
    long xpart_stride;
    float* x_out;        /* or NULL: receives x + sum_j xpart[j] (packed; must not alias x: other workgroups still read it).  Since ABI v11 rows >= M of
                          * a packed image (x_packed operands, xpart, x_out) are NEITHER READ NOR WRITTEN: lanes past M fetch row 0 and are zeroed before
                          * the MFMA, x_out stores are guarded -- a host that reuses a buffer keeps whatever its pad rows held (the engines zero theirs) */
    int w_bf16;          /* W is a cbx_pack_gemv_weight_bf16 image (opt-in: weights rounded to bf16, half the streamed bytes; M <= 16) */
    int flags;           /* (was reserved1) ABI v10, CBX_GEMV_* bits below; 0 = the plain launch */
    const float* ln_cw;  /* or NULL: LayerNorm instead of RMSNorm (GPT-2 ln_1 / ln_2 / ln_f): with norm_w = LN weight w, ln_cw[n] = */
    const float* ln_cb;  /* sum_k w[k] W[n][k] and ln_cb[n] = sum_k b[k] W[n][k] + bias[n] (constants of the layer, computed at load): */
                         /* out[m][n] = rstd[m] (sum_k x w W - mean[m] ln_cw[n]) + ln_cb[n], then `act` */
    /* ABI v11 -- column-tile / split-K form of the RMSNorm-folded packed GEMV (norm_w, packed fp32 operands in the 16-column image, M <= 16, no
     * bias / residual / activation; xpart / x_out as above).  col_tiles = 1 .. 4: a workgroup owns that many 16-column tiles, which share every x
     * register (activation : weight bytes per workgroup = 1 : col_tiles instead of 1 : 1 -- the decode GEMVs are bound by the bytes a CU moves,
     * profiles/r04_decode_launch_timeline.txt), and 1 / ksplit of K on 8 waves (K % (256 ksplit) == 0).  ksplit == 1: out = RMSNorm(x) W^T as
     * before.  ksplit > 1: out[ks] (part_stride floats apart) holds the UN-normalised partial sums of K slice ks and ssq_out[ks * 16 + m] the
     * slice's sum of squares of row m: the consumer adds the partials in fixed order and applies rstd = rsqrt(sum_ks ssq / K + eps)
     * (cbx_decode_attn_t.qkv_nparts does).  0 = the one-tile form above. */
    int col_tiles;
    float* ssq_out;
} cbx_gemv_t;
/* Packed GEMV weight layout (decode path; the weights are constants, so they are laid out once for the MFMA lane order):
 *   dst[(((tile * (K/32) + kb) * 2 + h) * 64 + lane) * 4 + s] = src[tile*16 + (lane & 15)][kb*32 + (lane >> 4)*8 + h*4 + s]
 * i.e. each (16-row tile, 32-deep K block) is 2 KiB of contiguous memory in exactly the order the 64 lanes of a wave load it
 * (two 1-KiB instructions).  Rows are zero-padded to a multiple of 16.  swiglu = 1: src holds [gate (F rows); up (F rows)] and
 * the tiles are interleaved gate/up per 16 features.  dst must hold ceil(N/16)*16*K floats.
 * swiglu = 8 / 12 / 4: the narrow-tile image (tr-row tiles, 4 * tr lanes per half block: dst[(((tile * (K/32) + kb) * 2 + h) * 4 * tr + q * tr + c) * 4 + s]
 * = src[tile * tr + c][kb*32 + q*8 + h*4 + s]) for cbx_gemv_t.half_tile; dst holds ceil(N/tr)*tr*K floats. */
int cbx_pack_gemv_weight_f32(const float* src, float* dst, int N, int K, long ld_src, int swiglu, void* stream);
/* the same image with the weights rounded (RNE) to bf16: [tile][K/32][lanes][8 bf16] (cbx_gemv_t.w_bf16); dst holds half the bytes */
int cbx_pack_gemv_weight_bf16(const float* src, void* dst, int N, int K, long ld_src, int swiglu, void* stream);
int cbx_gemv_f32(const cbx_gemv_t* p, void* stream);
/* cbx_gemv_t.flags (ABI v10: per launch, set by the caller -- the engines carry them in their own tune, nothing process-wide):
 * CBX_GEMV_PRE_EPI: the launch requests the operands of its epilogue (residual element, bias, LayerNorm-fold constants) together with its first
 *   weight batch instead of after the reduction -- one dependent memory round trip less; same values added in the same order.
 * CBX_GEMV_DEEP: an 8-wave plain packed GEMV whose waves own >= 256 of K (the down projection with ksplit = 1) requests 8 K blocks per batch
 *   instead of 4 -- half the dependent load batches.  Both: results unchanged bit for bit (tests/test_zz_abi_v9_gpu.py, on hardware). */
#define CBX_GEMV_PRE_EPI 1
#define CBX_GEMV_DEEP 2
/* CBX_GEMV_SHALLOW (ABI v13): the RMSNorm-folded SwiGLU launch (gate | up of the decode step) requests 2 K blocks per batch instead of 4: <= 128 VGPRs instead of
 * 162, so that its workgroups fit on a CU beside a workgroup of ANOTHER stream that leaves half of the register file free (the throughput schedule of
 * ChatterboxEngine.synthesize_pipelined: T3 of batch k + 1 beside flow + vocoder of batch k).  Same products, same order: bit-identical. */
#define CBX_GEMV_SHALLOW 4
/* TEST HOOKS (default 0): OR the bit into the flags of EVERY cbx_gemv_f32 launch of the process. */
int cbx_set_gemv_deep_batches(int on);
int cbx_set_gemv_epilogue_prefetch(int on);
/* x += sum_k part[k] (fixed order), h = RMSNorm(x) * w : residual add + split-K reduce + LlamaRMSNorm in one pass (the residual /
 * input_layernorm / post_attention_layernorm steps of HF LlamaDecoderLayer inside T3.inference's loop, t3.py:378-386) */
int cbx_add_rmsnorm_f32(float* x, const float* part, int ksplit, long part_stride, long ldp, const float* w, float* h,
                        int rows, int C, long ldx, long ldh, float eps, void* stream);
/* same with LayerNorm (rms = 0, bias b) for the GPT-2 backbone of Turbo/Nano (HF GPT2Block ln_1 / ln_2 / ln_f, driven by
 * T3.inference_turbo, t3.py:392-468) */
int cbx_add_norm_f32(float* x, const float* part, int ksplit, long part_stride, long ldp, const float* w, const float* b,
                     float* h, int rows, int C, long ldx, long ldh, float eps, int rms, void* stream);

/* ---- normalisation (wavefront reductions, one wave per row) ----
 * LayerNorm / RMSNorm over the last dim (C <= 4096, C % 4 == 0):
 *   y = act( (x-mean)*rstd*w + b ) [+ post_add[c]] [* rowmask]            (nn.LayerNorm, HF LlamaRMSNorm)
 * Replaces F.layer_norm (+Mish of CausalBlock1D, decoder.py:49-63) and LlamaRMSNorm. */
int cbx_layernorm_f32(const float* x, float* y, const float* w, const float* b, const float* post_add,
                      long rows, int C, long ldx, long ldy, float eps, int rms, int act, float out_scale,
                      void* stream);

/* ---- attention ----
 * Flash-style fp32 attention, head_dim 64: O = softmax(scale*Q K^T + mask) V, one (z1,head) per grid.y slice.
 * q/k/v/o are addressed as base + z1*s_b + t*s_t + head*64 (+ kv: s_kb/s_kt). key_lens[z1] masks keys >= len
 * (diffusers additive -1e10 bias, decoder.py:26-34,285-286); causal!=0 adds j<=i+causal_off (HF sdpa is_causal).
 * Replaces F.scaled_dot_product_attention in diffusers AttnProcessor2_0 and HF sdpa_attention_forward. */
int cbx_flash_attn_f32(const float* q, const float* k, const float* v, float* o, const int* key_lens,
                       int nz1, int n_heads, int Tq, int Tk, long q_sb, long q_st, long k_sb, long k_st,
                       long v_sb, long v_st, long o_sb, long o_st, float scale, int causal, void* stream);
/* ABI v15: the same with HEAD strides for K and V (k_sh / v_sh floats between the heads of one token; cbx_flash_attn_f32 = 64): keys and values read where
 * DynamicCache keeps them -- T3's KV cache [row][head][max_ctx][64] has k_st = 64, k_sh = max_ctx * 64.  Serves the prefill of the TEXT positions of T3.inference
 * when the 34 conditioning positions of a voice are already cached (t3.py:303-335: the prompt [cond | text | BOS BOS]; HF sdpa is_causal with Tq < Tk: query i sees
 * keys j <= i + (Tk - Tq)). */
int cbx_flash_attn_kv_f32(const float* q, const float* k, const float* v, float* o, const int* key_lens, int nz1, int n_heads, int Tq, int Tk,
                          long q_sb, long q_st, long k_sb, long k_st, long k_sh, long v_sb, long v_st, long v_sh, long o_sb, long o_st, float scale,
                          int causal, void* stream);
/* Same contract on the bf16 / fp16 matrix cores with split fp32 operands: precision 3 ("bf16x3", rel. error ~4e-6 per
 * contraction), 6 ("bf16x6", fp32-level) or 16 ("f16x3", fp32-level, fp16 operand range), see cbx_gemm_t.precision.
 * 5.3x / 2.7x / 5.3x fewer matrix-core cycles. */
int cbx_flash_attn_split_f32(const float* q, const float* k, const float* v, float* o, const int* key_lens,
                             int nz1, int n_heads, int Tq, int Tk, long q_sb, long q_st, long k_sb, long k_st,
                             long v_sb, long v_st, long o_sb, long o_st, float scale, int causal, int precision,
                             void* stream);

/* cbx_flash_attn_split_f32 at precision 16 whose output is written in plane format (o_sb / o_st / o_lo in halves): the attention result
 * feeds only the to_out projection (matcha/transformer.py:275-283 via diffusers Attention), a cbx_gemm_planes. */
int cbx_flash_attn_split_po(const float* q, const float* k, const float* v, void* o_planes, const int* key_lens,
                            int nz1, int n_heads, int Tq, int Tk, long q_sb, long q_st, long k_sb, long k_st,
                            long v_sb, long v_st, long o_sb, long o_st, long o_lo, float scale, int causal, void* stream);

/* (cbx_mlp_planes -- the fused feed-forward launch of ABI v7-v12 -- is gone: measured equal at B = 8 and 18-27 % behind at batch 1, no configuration wanted it) */
/* Flash attention (head_dim 64, f16x3 arithmetic) on plane-format operands: q, k [token][d] planes as a cbx_gemm_planes P output holds them,
 * vt = V^T [d][token] planes (the v projection computed with swapped operands: A = W_v, W = the activations; row stride vt_sd >= Tk rounded
 * up to 8, tails finite), o [token][d] planes.  All strides / plane offsets in halves; heads are 64 columns (q, k, o) or 64 rows (vt) apart.
 * No operand conversion, no VALU staging: K and V^T tiles are DMA'd global -> LDS.  Replaces diffusers Attention / F.scaled_dot_product_attention
 * inside BasicTransformerBlock (matcha/transformer.py:243-316). */
int cbx_flash_attn_planes(const void* q, const void* k, const void* vt, void* o, const int* key_lens, int nz1, int n_heads, int Tq,
                          int Tk, long q_sb, long q_st, long q_lo, long k_sb, long k_st, long k_lo, long vt_sb, long vt_sd,
                          long vt_lo, long o_sb, long o_st, long o_lo, float scale, int causal, void* stream);
/* ABI v13: the same with the kernel version chosen PER CALL (0 = the library's automatic choice: version 4, or its 128-query twin (the kernel of version 5) where the grid
 * of 256-query workgroups would fill at most half of the chip -- batch 1; else 1 .. 6 as cbx_set_attn_planes_version).  5 = the CO-RESIDENT form:
 * the free-running loop of version 4 on 4-wave workgroups of 128 queries, one per CU (96 KiB of LDS, one wave of ~200 VGPRs per SIMD): 3/5 of the register
 * file stay free for another stream's workgroups; bit-identical to version 4; +8 % per launch when alone (profiles/r05_overlap_*). */
int cbx_flash_attn_planes_v(const void* q, const void* k, const void* vt, void* o, const int* key_lens, int nz1, int n_heads, int Tq,
                            int Tk, long q_sb, long q_st, long q_lo, long k_sb, long k_st, long k_lo, long vt_sb, long vt_sd,
                            long vt_lo, long o_sb, long o_st, long o_lo, float scale, int causal, int version, void* stream);
/* tuning knob (A/B hook, no engine calls it): kernel version of cbx_flash_attn_planes (0 = automatic; 4 = free-running loop, every wave meets the others once per
 * key tile; 2 = two wave groups alternating matrix / vector blocks; 1 = one group; 3 = version 2 with wave priorities; 6 = version 4 with its DMAs issued between
 * the softmax and the PV product: measured equal, profiles/r06_y_*) */
int cbx_set_attn_planes_version(int v);

/* Single-query decode attention over a KV cache (HF DynamicCache + sdpa, q_len == 1; t3.py:378-384).
 * cache layout [row][head][pos][64]; ctx_lens[row] = number of valid positions (including the new token). */
int cbx_decode_attn_f32(const float* q, const float* kc, const float* vc, float* o, const int* ctx_lens,
                        int rows, int n_heads, long q_ld, long o_ld, long cache_row_stride, long cache_head_stride,
                        float scale, void* stream);

/* Fused per-token attention of the decode loop: HF apply_rotary_pos_emb on q,k of the fused qkv row (head_dim 64,
 * rotate_half form; cos_t == sin_t == NULL skips the rotation: GPT-2) + DynamicCache append at positions[row] + sdpa over
 * positions [0, positions[row]] (t3.py:378-384, 438-446).
 * o_packed: o is written in the packed operand layout of the o-projection GEMV (cbx_gemv_t.x_packed; o_ld ignored). */
int cbx_decode_attn_rope_f32(const float* qkv, const int* positions, const float* cos_t, const float* sin_t, float* kc,
                             float* vc, float* o, int rows, int n_heads, long ld_qkv, long o_ld, int o_packed,
                             long cache_row_stride, long cache_head_stride, float scale, void* stream);
/* ABI v10: the same op with everything it depends on IN the call -- the form the engines use (no process-wide state: two engines, two streams
 * or a hipGraph captured earlier never see each other's geometry or share a workspace).
 *   unroll   : key rows in flight per 16-lane group and step: 0 (= 4), 4, 8 or 16.
 *   pipeline : bit 0 = software-pipelined K / V stream (the rows of the next step are requested before the current step is multiplied; two
 *              register sets; 4 or 8 rows per step); bit 1 = non-temporal K / V loads (4 rows per step); bit 2 = speculative first step
 *              (positions 0 .. 63 of every (row, head) are requested before positions[row] has arrived; needs cache_head_stride >= 64 * 64).
 *              Every combination gives the results of pipeline = 0 bit for bit (hardware test: tests/test_zz_abi_v9_gpu.py).
 *   split_ws / split_cnt / split_pairs : CALLER-OWNED workspace of the split-context form -- 66 * 8 floats per (row, head) and one int per
 *              (row, head), zeroed once by the caller, re-armed by every launch -- or NULL.  With rows * n_heads < 128 <= split_pairs the
 *              context of a (row, head) that is >= split_min (0 = 512) positions long is walked by up to 8 workgroups, merged in split order
 *              (deterministic).  One workspace serves launches that are ordered on one stream; concurrent launches need their own. */
typedef struct cbx_decode_attn_t {
    const float* qkv; const int* positions; const float *cos_t, *sin_t;
    float *kc, *vc, *o;
    int rows, n_heads; long ld_qkv, o_ld; int o_packed;
    long cache_row_stride, cache_head_stride; float scale;
    int unroll, pipeline, split_min;
    float* split_ws; int* split_cnt; long split_pairs;
    /* ABI v11: qkv_nparts = 2 .. 4: `qkv` holds that many UN-normalised split-K partial sums of the fused q/k/v row (cbx_gemv_t.col_tiles with
     * ksplit > 1), qkv_part_stride floats apart; the kernel adds them in fixed order and multiplies by rstd[row] = rsqrt(sum_p qkv_ssq[p * 16 + row]
     * / rms_dim + rms_eps) (LlamaRMSNorm of the projection's input, folded through the contraction); rows <= 16.  0 / 1: qkv is the finished row. */
    int qkv_nparts; long qkv_part_stride; const float* qkv_ssq; int rms_dim; float rms_eps;
} cbx_decode_attn_t;
int cbx_decode_attn_rope(const cbx_decode_attn_t* p, void* stream);

/* ---- few-row decode (ABI v14): 1 .. 4 activation rows, weights in the checkpoint's row-major layout ----
 * cbx_gemv_row_f32:  out[m][n] = act( x'[m] . W[n][:] + bias[n] ) + res[m][n],  m < M <= 4, n < N, where the operand x' is
 *     x                                   (ln_w == NULL, parts == NULL),
 *     LayerNorm(x) * ln_w + ln_b          (ln_w, ln_b; eps; two-pass variance: F.layer_norm), or
 *     the attention output of the row     (parts: the n_parts split-context records per head that cbx_decode_attn_parts left, merged in
 *                                          slice order; K == 64 n_heads; x unused).
 * A wave owns rows_per_wave (0 = automatic: ~1024 waves) output columns over the whole of K and requests all of its weights up front: no LDS
 * reduction, no partial images.  K in {256, 768, 1024, 3072, 4096}; W [N][ldw] fp32, 16-byte aligned; res may alias out.
 * Replaces HF Conv1D / nn.Linear at q_len == 1 + GPT2Block ln_1 / ln_2 / ln_f inside T3.inference_turbo's loop (models/t3/t3.py:435-460) for
 * batches of at most 4 rows (the engines use it up to 2: beyond, the 16-row MFMA tile of cbx_gemv_f32 is faster). */
#define CBX_ATTN_PART_REC 68 /* floats per (row, head, slice) record: {running max, sum, -, -, 64 numerators} */
typedef struct cbx_gemv_row_t {
    const float* x; const float* W; const float* bias; const float* res; float* out;
    const float *ln_w, *ln_b; float eps;
    const float* parts; int n_parts, n_heads;
    int N, K; long ldw;
    int act;            /* CBX_ACT_* after the bias */
    int rows_per_wave;  /* 0 = automatic; 1, 2, 3, 4, 8 (K >= 3072 or parts: 1, 2) */
    int M;              /* activation rows, 1 .. 4 (0 = 1) */
    long ldx, ldo, ldr; /* M > 1: floats between consecutive rows of x, out, res */
    long parts_row_stride; /* M > 1: floats between the records of consecutive rows (n_heads * n_parts * CBX_ATTN_PART_REC when dense) */
} cbx_gemv_row_t;
int cbx_gemv_row_f32(const cbx_gemv_row_t* p, void* stream);
/* Decode attention of ONE new token per row over a small (row, head) grid, split over n_splits workgroups per (row, head) by 16-position
 * chunks (chunk c belongs to slice c % n_splits -- independent of the context length, so the cache stream starts before positions[] has
 * arrived).  Appends the token's k / v (RoPE'd when cos_t / sin_t are given; GPT-2: NULL) to the caches at positions[row] and leaves
 * parts[((row * n_heads + head) * n_splits + slice) * CBX_ATTN_PART_REC] for the consumer to merge (cbx_gemv_row_t.parts): no ticket, no
 * fence, no last-arriver.  chunks = 16-position chunks in flight per workgroup (2, 4 (= 0) or 8).  The caches hold >= max_ctx positions
 * behind every (row, head).  Replaces the q_len == 1 attention of HF GPT2Attention / LlamaAttention (t3.py:435-460 via sdpa). */
typedef struct cbx_attn_parts_t {
    const float* qkv; const int* positions; const float *cos_t, *sin_t;
    float *kc, *vc, *parts;
    int rows, n_heads, n_splits, chunks, max_ctx;
    long ld_qkv, cache_row_stride, cache_head_stride;
    float scale;
} cbx_attn_parts_t;
int cbx_decode_attn_parts(const cbx_attn_parts_t* p, void* stream);
/* tuning knob: tile shape of the split-bf16 GEMM (0 = automatic; 64, 12864, 128, 1282) */
int cbx_set_split_tile(int t);
/* TEST HOOKS of the positional cbx_decode_attn_rope_f32 (process-wide; the engines pass cbx_decode_attn_t instead): unroll / pipeline /
 * split_min as the fields above, and the workspace it splits through, registered for the calling
 * thread's current device (single-stream use only). */
int cbx_set_decode_attn_unroll(int u);
int cbx_set_decode_attn_pipeline(int on);
int cbx_set_decode_attn_workspace(float* ws, int* zeroed_counters, long max_pairs);
int cbx_set_decode_attn_split_min(int min_ctx);

/* Row softmax over materialised scores with optional relative-position term and key mask
 * (RelPositionMultiHeadedAttention.forward, transformer/attention.py:249-330):
 *   p[z][i][j] = softmax_j( scale*(ac[z][i][j] + bd[z][i][T-1-i+j]) ), keys j >= key_lens[z1] -> 0. */
int cbx_softmax_relpos_f32(const float* ac, const float* bd, float* p, const int* key_lens, int nz1, int nz2,
                           int Tq, int Tk, long ld_ac, long ld_bd, long ld_p, long zs_ac, long zs_bd, long zs_p,
                           float scale, void* stream);

/* The same attention WITHOUT materialised scores (flash form, exact fp32 MFMA; written after the GPU budget of round 3 was spent: verified on
 * the SIMT emulator against the materialised path, not yet timed):
 *   o[z][i][h] = sum_j softmax_j( scale*(qu[z][i][h] . k[z][j][h] + qv[z][i][h] . pp[T-1-i+j][h]) ) v[z][j][h],  keys j >= key_lens[z] masked.
 * qu = q + pos_bias_u, qv = q + pos_bias_v, k, v: [z][token][head][64] views that share one batch stride q_sb and token stride q_st (the
 * fused q | q | k | v projection); pp = linear_pos(pos_emb): (2T-1) rows of n_heads*64, row stride pp_st; o: [z][token][head][64].
 * Memory O(T) instead of 16 T^2 floats per head; the position term costs as many MFMAs as q k^T (each 32-row block of pp (q+v)^T serves
 * two key blocks).  Replaces RelPositionMultiHeadedAttention.forward (transformer/attention.py:249-330) for long utterances (60 s VC). */
int cbx_flash_relpos_f32(const float* qu, const float* qv, const float* k, const float* v, const float* pp, float* o, const int* key_lens,
                         int nz1, int n_heads, int T, long q_sb, long q_st, long pp_st, long o_sb, long o_st, float scale, void* stream);

/* ---- elementwise / glue ---- */
/* y[r][c] = act(x[r][c]) (per-column param for snake), 2-D strided: the activations that cannot ride a GEMM / LayerNorm epilogue
 * (Mish of the ResNet time MLPs, matcha/decoder.py:49,58; Snake at a ResBlock entry, hifigan.py:34-60,146-150). */
int cbx_act_f32(const float* x, float* y, const float* param, long rows, int C, long ldx, long ldy, int act,
                float slope, void* stream);
/* generic 2-D strided copy / scale-add: y = a*x + b*y (skip-connection concat of the CFM up block, decoder.py:294,316-317; conformer
 * macaron residuals, transformer/encoder_layer.py:193-229) */
int cbx_axpby_f32(const float* x, float* y, long rows, int C, long ldx, long ldy, float a, float b, void* stream);
/* out[r][:] = table[ids[r]][:] * scale [+ table2[ids2[r]][:]]   (nn.Embedding gathers: speech_emb + LearnedPositionEmbeddings, t3.py:116-119,
 * 370-371; flow.input_embedding, flow.py:106,166; negative ids give zeros).
 * flags bit 1 (value 2): out is the packed operand image of a decode GEMV (cbx_gemv_t.x_packed, K = C % 32 == 0; ld_out ignored;
 * rows of the last 16-row tile that are not written keep their previous contents). */
int cbx_embed_f32(const long long* ids, const float* table, const float* table2, const int* ids2, float* out,
                  long rows, int C, long ld_out, float scale, int flags, void* stream);
/* RoPE (HF apply_rotary_pos_emb, rotate_half form, as called by LlamaAttention in T3's prefill, t3.py:326-333) on q,k rows of a fused
 * qkv buffer + KV-cache append.
 * positions[r] gives the absolute position of row r; cos/sin tables are [max_pos][64] (cat(freqs,freqs)). */
int cbx_rope_kv_f32(float* qkv, const int* positions, const float* cos_t, const float* sin_t, float* kc, float* vc,
                    const int* cache_rows, long n_rows, int n_heads, long ld_qkv, long cache_row_stride,
                    long cache_head_stride, void* stream);
/* CFM Euler + CFG update (flow_matching.py:125-141): x += dt*((1+w)*v[b] - w*v[B+b]) written to both row sets. */
int cbx_cfm_euler_f32(float* xin, const float* v, int B, long T, int C, long ld_x, long ld_v, long xs_b, long vs_b,
                      float dt, float w, int cfg, void* stream);

/* ---- T3 sampler (t3.py:339-368 + HF logits processors): one workgroup per utterance ---- */
#define CBX_SAMPLER_NPARAMS 8
typedef struct cbx_sampler_t {
    const float* logits;       /* [2*B][ld] rows b (cond) and B+b (uncond) when cfg, else [B][ld] */
    long ld; int V; int B; int cfg;
    float cfg_weight, temperature, min_p, top_p, rep_penalty;
    int top_k;                 /* 0 = off */
    int order;                 /* 0: T3.inference (penalty,temperature,min-p,top-p); 1: inference_turbo
                                  (temperature,top-k,top-p,penalty)  (t3.py:339-356 vs 396-404) */
    int ban_token;             /* probability forced to 0 (EOS ban for fixed-length runs) or -1 */
    int eos_token;
    int ban_from;              /* ids >= ban_from (EOS included) get probability 0; 0 = off.  Synthetic random-weight
                                  fixed-length runs use 6561 so that only valid S3 tokens are emitted.  Bans act AFTER the
                                  processors (like zeroing softmax entries); if no probability mass is left (the arg-max is
                                  banned and min-p / top-k pruned the rest) the result is the allowed id with the largest
                                  CFG-combined raw logit, lowest id on ties */
    unsigned char* seen;       /* [B][V] 0/1 map of generated ids (repetition penalty) */
    const float* uniforms;     /* [B][max_steps] U[0,1): the injected RNG of torch.multinomial */
    int max_steps;
    int* step;                 /* [B] step index, incremented */
    long long* out_tokens;     /* [B][max_steps] */
    int* done;                 /* [B] set when EOS sampled; finished utterances are skipped */
    int* n_generated;          /* [B] */
    long long* next_ids;       /* [rows] id to embed at the next decode step (both CFG rows), or NULL */
    int* next_pos_ids;         /* [rows] learned speech-position index of the next input (= step+1), or NULL */
    int* positions;            /* [rows] RoPE / cache position of the next input, incremented, or NULL */
    int* ctx_lens;             /* [rows] context length of the next decode step, incremented, or NULL */
    const float* dev_params;   /* ABI v4: NULL, or [B][CBX_SAMPLER_NPARAMS] floats in DEVICE memory that override, per utterance,
                                  {cfg_weight, temperature, min_p, top_p, rep_penalty, top_k, ban_token, ban_from}: a new request
                                  with other settings then needs no re-capture of the decode hipGraph */
} cbx_sampler_t;
int cbx_t3_sample(const cbx_sampler_t* p, void* stream);

/* ---- stage-level entry point: ONE token step of T3.inference's loop for every row (t3.py:338-386) ----
 * Sequences the kernel-level entry points above on `stream` (embedding gather, n_layers x 5 launches, head, sampler): no allocation,
 * no synchronisation, hipGraph-capturable.  rows <= 16 (the packed-operand path); all weights are cbx_pack_gemv_weight_f32 images
 * (wgu with swiglu = 1), all buffers caller-owned device memory. */
typedef struct cbx_t3_layer_t {
    const float *ln1, *ln2;               /* [dim] RMSNorm weights (input_layernorm, post_attention_layernorm) */
    const float *wqkv, *wo, *wgu, *wd;    /* packed images: [3*dim][dim], [dim][dim], gate|up [2*ffn][dim], [dim][ffn] */
} cbx_t3_layer_t;
typedef struct cbx_t3_step_t {
    int n_layers, rows, dim, ffn, n_heads, vocab;
    int o_nw, gu_nw, d_nw, d_ksplit;      /* launch geometry (T3Engine.tune): 8 / 8 / 16 / 2; d_ksplit = 1 (ABI v9): the down projection adds
                                             the residual itself (no partial images, no fold in the next q/k/v GEMV) */
    int half_tiles, w_bf16;               /* cbx_gemv_t.half_tile of the wo / wd images (0, 1 = 8, 4); 1: all weight images are bf16 */
    float eps, attn_scale;
    const cbx_t3_layer_t* layers;         /* HOST array [n_layers] */
    const float *speech_emb, *speech_pos, *final_norm, *head; /* embeddings [V][dim], [P][dim]; tfmr.norm; packed head [ceil16(vocab)][dim] */
    const float *cos_t, *sin_t;           /* RoPE tables [max_pos][64] */
    float *kc, *vc;                       /* KV cache [n_layers][rows][n_heads][max_ctx][64] */
    long kv_row_stride, kv_head_stride;   /* floats: n_heads*max_ctx*64, max_ctx*64 */
    const long long* next_ids;            /* [rows] token to embed (written by the sampler) */
    const int *next_pos_ids, *positions;  /* [rows] learned position index; RoPE / cache position */
    float *x_a, *x_b;                     /* packed residual images [ceil16(rows)][dim] (ping-pong), pad rows zero */
    float *qkv, *att, *g, *pd;            /* [rows][3*dim]; packed [ceil16(rows)][dim]; packed [..][ffn]; partial images [d_ksplit][ceil16(rows)][dim] */
    float* logits;                        /* [rows][ld_logits] */
    long ld_logits;
    const cbx_sampler_t* sampler;         /* sampler descriptor (host struct) run at the end of the step, or NULL */
    int qkv_tile;                         /* ABI v9: cbx_gemv_t.half_tile of the wqkv images (0 or 12) */
    /* ABI v10: the rest of the step's geometry (was process-wide): cbx_decode_attn_t.unroll / pipeline / split_min / split_* of every attention
     * launch, cbx_gemv_t.flags of every GEMV launch */
    int da_unroll, da_pipeline, da_split_min, gemv_flags;
    float* da_ws; int* da_cnt; long da_pairs;
    /* ABI v11: qkv_ksplit = 2 or 4 (dim % (256 qkv_ksplit) == 0; with qkv_ct = column tiles per workgroup, qkv_tile = 0): the q/k/v projection in the split-K column-tile form --
     * qkv then holds [qkv_ksplit][rows][3*dim] partial sums and qkv_ssq [qkv_ksplit][16] sums of squares, which the attention launch folds;
     * head_ct > 0: the speech head in the column-tile form (ksplit 1).  0 = the one-tile forms. */
    int qkv_ksplit, qkv_ct, head_ct;
    float* qkv_ssq;
} cbx_t3_step_t;
int cbx_t3_decode_step(const cbx_t3_step_t* d, void* stream);

/* ---- handle-level entry points: the TOKEN LOOP of T3.inference in C (t3.py:338-386 incl. its EOS test `:366`; SURVEY.md 8b "cbx_t3_generate") ----
 * cbx_t3_loop_create deep-copies the step descriptor (its host arrays: layers, sampler) and captures ONE cbx_t3_decode_step in a hipGraph (thread-local
 * capture on a stream of the library's own -- `stream` may be the legacy default stream --; nothing is allocated on the device); cbx_t3_loop_run replays it up to n_steps times on `stream`.  poll_every > 0: after every
 * poll_every steps the per-utterance done flags (cbx_sampler_t.done) are fetched and the loop ends early once every utterance has sampled its EOS --
 * the reference's per-token host sync, amortised; poll_every = 0: all n_steps are enqueued without any synchronisation (asynchronous use).
 * *steps_run = steps enqueued.  The caller owns every device buffer of the descriptor and keeps it alive until cbx_t3_loop_destroy.  One handle per
 * (descriptor, stream of launches); not thread-safe.  With cbx_t3_prefill a C host runs the whole device side of T3.inference without Python. */
typedef struct cbx_t3_loop cbx_t3_loop_t;
int cbx_t3_loop_create(const cbx_t3_step_t* step, void* stream, cbx_t3_loop_t** out);
int cbx_t3_loop_run(cbx_t3_loop_t* h, int n_steps, int poll_every, void* stream, int* steps_run);
int cbx_t3_loop_destroy(cbx_t3_loop_t* h);

/* ---- stage-level entry point: the PREFILL of T3.inference for every row (t3.py:303-335 -> t3_hf_backend.py:71-111: HF LlamaModel over the S prompt
 * positions, KV cache filled) ----
 * x (rows * S, dim) holds the input embeddings (prepare_input_embeds + the second BOS, t3.py:102-130, 305-313; rows right-padded to S) and returns the last
 * layer's residual stream; the caller applies tfmr.norm + speech_head to the positions it needs (cbx_layernorm_f32 + cbx_gemm_f32).  `layers`: the
 * cbx_t3_layer_t of the decode step with ROW-MAJOR weights (torch Linear layout; wgu = the [32 gate | 32 up]-interleaved image, cbx_gemm_t.swiglu).
 * positions / cache_rows [rows * S]: RoPE position and KV-cache row of every prompt position.  precision: cbx_gemm_t.precision of the plain projections and
 * of the attention (0 / 1 = exact fp32 MFMA, the parity path; 6 = bf16x6).  Sequences kernel-level entry points only: no allocation, no synchronisation,
 * hipGraph-capturable; results bit-identical to issuing the same launches one by one. */
typedef struct cbx_t3_prefill_t {
    int n_layers, rows, S, dim, ffn, n_heads, precision;
    float eps, attn_scale;
    const cbx_t3_layer_t* layers;         /* HOST array [n_layers], row-major weights */
    float* x;                             /* [rows * S][dim] in / out */
    float *h, *qkv, *att, *g;             /* workspaces: [rows * S][dim], [..][3 * dim], [..][dim], [..][ffn] */
    const int *positions, *cache_rows;    /* [rows * S] */
    const float *cos_t, *sin_t;           /* RoPE tables [max_pos][64] */
    float *kc, *vc;                       /* KV cache [n_layers][rows][n_heads][max_ctx][64] */
    long kv_layer_stride, kv_row_stride, kv_head_stride;  /* floats */
} cbx_t3_prefill_t;
int cbx_t3_prefill(const cbx_t3_prefill_t* d, void* stream);

/* ---- ABI v16: the PREFILL of T3.inference_turbo for every row (t3.py:392-468 -> HF GPT2Model over [speaker | prompt tokens | text | start-speech], absolute
 * position embeddings already added to x; KV cache filled) ----
 * x (rows * S, dim) in / out as cbx_t3_prefill_t; per layer ln_1 -> c_attn (+ bias) -> K / V append at positions[] of cache row cache_rows[] -> causal attention ->
 * attention c_proj (+ bias + residual) -> ln_2 -> c_fc (+ bias + gelu_new) -> mlp c_proj (+ bias + residual); weights ROW-MAJOR (N, K) (the checkpoint's Conv1D
 * weights transposed), ffn = 4 * dim.  prefix > 0: the first `prefix` positions of every row are ALREADY in the KV cache (the speaker + prompt-token positions of a
 * voice see only themselves under the causal mask and carry absolute positions: computed once per voice); x then holds the remaining S positions of every row,
 * positions[] start at `prefix`, and the attention reads keys / values [prefix | S] from the cache (cbx_flash_attn_kv_f32).  Sequences kernel-level entry points
 * only (exact fp32): no allocation, no synchronisation; results bit-identical to issuing the same launches one by one. */
typedef struct cbx_gpt2_layer_t {
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *wqkv, *bqkv, *wo, *bo, *wfc, *bfc, *wpr, *bpr;
} cbx_gpt2_layer_t;
typedef struct cbx_gpt2_prefill_t {
    int n_layers, rows, S, prefix, dim, n_heads;
    float eps, attn_scale;
    const cbx_gpt2_layer_t* layers;       /* HOST array [n_layers] */
    float* x;                             /* [rows * S][dim] in / out */
    float *h, *qkv, *att, *g;             /* workspaces: [rows * S][dim], [..][3 * dim], [..][dim], [..][4 * dim] */
    const int *positions, *cache_rows;    /* [rows * S] */
    float *kc, *vc;                       /* KV cache [n_layers][rows][n_heads][max_ctx][64] */
    long kv_layer_stride, kv_row_stride, kv_head_stride;  /* floats */
} cbx_gpt2_prefill_t;
int cbx_gpt2_prefill(const cbx_gpt2_prefill_t* d, void* stream);

/* ---- stage-level entry points of the S3Gen flow decoder and of the HiFT vocoder (ABI v12) ----
 * A plane-format operand (see PLANE-FORMAT operands above) as the stage descriptors carry it: base of the h plane with any column offset applied, row
 * stride and plane offset in halves. */
typedef struct cbx_planes_t { void* p; long ld, lo; } cbx_planes_t;

/* BasicTransformerBlock of the CFM estimator (matcha/transformer.py:243-316): norm1 -> to_q | to_k | to_v -> attention -> to_out + residual -> norm3 ->
 * FeedForward (GELU) + residual.  wqkv: planes of the (1536, 256) row block [Wq; Wk; Wv]; wo (256, 512); w1 (1024, 256); w2 (256, 1024). */
typedef struct cbx_cfm_tblock_t {
    const float *n1_w, *n1_b, *n3_w, *n3_b, *bo, *b1, *b2;
    cbx_planes_t wqkv, wo, w1, w2;
} cbx_cfm_tblock_t;
/* One down / mid / up stage of ConditionalDecoder (decoder.py:243-333): CausalResnetBlock1D (c1, n1 + Mish + time bias, c2, n2 + Mish, 1x1 res_conv), n_tb
 * transformer blocks and -- down and up stage only -- the trailing CausalConv1d (tail.p == NULL otherwise).  Conv weights: planes of the tap-major
 * (256, taps * cin) image. */
typedef struct cbx_cfm_stage_t {
    cbx_planes_t c1, c2, res, tail;
    const float *c1_b, *n1_w, *n1_b, *c2_b, *n2_w, *n2_b, *res_b, *tail_b;
    int cin, n_tb;
    const cbx_cfm_tblock_t* tb;           /* HOST array [n_tb] */
} cbx_cfm_stage_t;
/* CausalConditionalCFM.solve_euler (models/s3gen/flow_matching.py:78-145, 196-233; meanflow: 196-233 with cfg = 0) around ConditionalDecoder.forward on
 * plane-format operands (the f16x3 numerics, cbx_gemm_t.precision 16): n_steps x [x columns of xinP re-split (steps > 0), estimator -> v, Euler (+ CFG) update
 * of xin].  xin (rows, T, 320) fp32 = [x | mu | spk | cond] per position (rows = 2 B with CFG: the second B rows hold [x | 0 | 0 | 0]); xinP its plane image,
 * split by the caller before the call; on return xin[:B, :, :80] is the mel.  tbias [n_steps][n_stages][256]: every ResNet's time-MLP output per step (a
 * function of the schedule only; the caller computes it once).  dt: HOST array [n_steps] of t_span[k + 1] - t_span[k].  lens [rows]: valid positions per row.
 * Needs T even, rows * T > 32, (rows * T + 512) * 4096 < 2^31.  Sequences kernel-level entry points only (results bit-identical to issuing them one by one):
 * no allocation, no synchronisation, hipGraph-capturable. */
typedef struct cbx_cfm_t {
    int n_stages, rows, B, n_steps, cfg, fused_qkv, fused_ln;    /* fused_ln (ABI v13; the slot of v12's fused_mlp): 1 = norm3 from the attention
                                                                  * out-projection's epilogue (cbx_gemm_pl_t.ln_w), 2 = also the next block's norm1
                                                                  * from ff2's; 0 = LayerNorm launches of their own */
    long T;
    float cfg_rate;
    const float* dt;                      /* HOST [n_steps] */
    const cbx_cfm_stage_t* stages;        /* HOST [n_stages]: down, mid ..., up */
    cbx_planes_t fin_c, fin_proj;         /* final_block conv (256, 3 * 256), final_proj (80, 256) */
    const float *fin_c_b, *fin_n_w, *fin_n_b, *fin_proj_b;
    const float* tbias;
    const int* lens;
    float* xin;
    cbx_planes_t xinP;                    /* (rows * T, 320) */
    float *ra, *rb, *x, *v;               /* workspaces (rows, T, 256) x 3, estimator output (rows, T, 80) */
    cbx_planes_t aP, hP, qkP, attP, ffP, xP, yP, catP;   /* plane workspaces over rows * T rows: 256, 256, 1024, 512, 1024, 256, 256, 512 columns */
    cbx_planes_t vtP;                     /* V^T: (rows * 512) rows x (T rounded up to 8) columns, the padding zero */
    int gemm_tile, attn_version;          /* ABI v13: cbx_gemm_pl_t.tile of every plane GEMM / version of every attention launch (0, 0 = the defaults;
                                           * CBX_PL_TILE_CORESIDENT, 5 = the co-resident forms of the throughput schedule) */
} cbx_cfm_t;
int cbx_cfm_solve(const cbx_cfm_t* d, void* stream);

/* ConformerEncoderLayer of the S3Gen token encoder (transformer/encoder_layer.py:160-236 with RelPositionMultiHeadedAttention, attention.py:249-330):
 * w4 / b4 = the fused projection [q + pos_bias_u | q + pos_bias_v | k | v] (2048, 512); wpos = linear_pos (512, 512, no bias). */
typedef struct cbx_conformer_t {
    const float *ln_mha_w, *ln_mha_b, *w4, *b4, *wpos, *wo, *bo, *ln_ff_w, *ln_ff_b, *w1, *b1, *w2, *b2;
} cbx_conformer_t;
/* UpsampleConformerEncoder.forward + encoder_proj (transformer/upsample_encoder.py:237-304, flow.py:161-169) for B rows of N tokens: input_embedding
 * gather (ids < 0 = padded position: zero vector), embed (Linear + LayerNorm * sqrt(512)), PreLookaheadLayer, n_enc conformer layers, Upsample1D (nearest x2
 * + conv k5, fused in the conv's address map), up_embed, n_up conformer layers over 2 N positions, after_norm, encoder_proj -> mu (B, 2 N, 80).  The rel-pos
 * attention runs in its flash form (cbx_flash_relpos_f32).  pe / pe2: EspnetRelPositionalEncoding tables of N resp. 2 N positions ((2 T - 1) x 512, row r <->
 * relative position T - 1 - r), constants of the length.  precision: cbx_gemm_t.precision of every Linear / conv.  Workspaces: x0, xa, y1, x2 (B N x 512);
 * xu, xb, h, att (B 2N x 512); q4, ff (B 2N x 2048); pp (4 N x 512).  Same contract as cbx_cfm_solve. */
typedef struct cbx_s3enc_t {
    int B, N, n_enc, n_up, precision;
    const long long* ids;                 /* [B * N] */
    const int *lens, *lens2;              /* [B]: valid tokens per row, twice that */
    const float *emb, *e_w, *e_b, *e_lnw, *e_lnb, *u_w, *u_b, *u_lnw, *u_lnb, *pl1_w, *pl1_b, *pl2_w, *pl2_b, *up_w, *up_b, *after_w, *after_b, *proj_w, *proj_b;
    const cbx_conformer_t *enc, *up_enc;  /* HOST arrays [n_enc], [n_up] */
    const float *pe, *pe2;
    float *x0, *xa, *y1, *x2, *xu, *xb, *h, *q4, *pp, *att, *ff;
    float* mu;
} cbx_s3enc_t;
int cbx_s3gen_encode(const cbx_s3enc_t* d, void* stream);

/* The front half of HiFTGenerator.inference (hifigan.py:462-469): ConvRNNF0Predictor (f0_predictor.py:52-55: 5 x Conv1d k3 + ELU, Linear, abs; always exact
 * fp32 -- its output is integrated into a phase over ~10^5 samples) -> f0 (B, T), then f0_upsamp + SourceModuleHnNSF (hifigan.py:201-231, 267-283) with the
 * caller's initial phases (B, 9) and noise (B, 9, 480 T) -> s (B, 480 T).  buf0 / buf1: (B, T, 512) workspaces; cum: (B, 9, T) doubles.  lens: NULL or [B]. */
typedef struct cbx_hift_f0_t {
    int B;
    long T;
    const float* mel;                     /* (B, T, 80) */
    const int* lens;
    const float *f0_w[5], *f0_b[5], *cls_w, *cls_b, *src_w;
    float src_b;
    const float *phase, *noise;
    float *buf0, *buf1, *f0, *s;
    double* cum;
} cbx_hift_f0_t;
int cbx_hift_f0_source(const cbx_hift_f0_t* d, void* stream);

/* ResBlock of HiFT (hifigan.py:118-161): three (Snake, dilated conv, Snake, conv, + x) rounds; conv weights tap-major (C, k * C), weight_norm folded */
typedef struct cbx_hift_resblock_t {
    const float *c1_w[3], *c1_b[3], *c2_w[3], *c2_b[3], *a1[3], *a2[3];
} cbx_hift_resblock_t;
/* HiFTGenerator.decode (hifigan.py:412-444) for B rows: STFT of the source s, conv_pre, 3 x [leaky ReLU, ConvTranspose1d (phase-packed 3-tap form, cbx_gemm_t),
 * + source_resblock(source_down(STFT)), mean of 3 ResBlocks], conv_post, iSTFT (+ S3Gen's trim_fade when fade).  mel (B, T, 80) channel-last, s (B, 480 T),
 * wav (B, 480 T).  lens: NULL or [5][B] = valid rows of {mel, stage 1, stage 2, stage 3 = STFT frames, source samples} = {n, 8 n, 40 n, 120 n + 1, 480 n}.
 * precision: cbx_gemm_t.precision of every conv.  Workspaces: spec, post (B, 120 T + 1, 32); x0 (B, T, 512); xs .. nxt[1]: (B, 120 T + 1, 64) floats each
 * (the widest stage; stages 1 and 2 use a prefix).  Same contract as cbx_cfm_solve. */
typedef struct cbx_hift_t {
    int B, precision, fade;
    long T;
    const float *mel, *s;
    float* wav;
    const int* lens;
    const float *conv_pre_w, *conv_pre_b, *ups_w[3], *ups_b[3], *src_down_w[3], *src_down_b[3], *conv_post_w, *conv_post_b;
    cbx_hift_resblock_t src_rb[3], rb[9];
    float *spec, *post, *x0, *xs, *t1, *xa, *xb, *an, *si, *sa, *acc, *a0, *nxt[2];
} cbx_hift_t;
int cbx_hift_decode(const cbx_hift_t* d, void* stream);

/* ---- HiFT source + (i)STFT (hifigan.py:201-231,267-283,396-410) ---- */
int cbx_hift_source_f32(const float* f0, const float* phase, const float* noise, const float* lin_w, float lin_b,
                        float* s, double* frame_cum, int B, int T, int up, float sr, void* stream);
/* sample_lens[b] (or NULL): per-row signal length; the centre-reflect padding mirrors at that row's own end */
int cbx_hift_stft_f32(const float* s, float* spec, const int* sample_lens, int B, long L, long ld_spec, void* stream);
/* x[b][frame][0..8] log-magnitude, [9..17] phase pre-sin (conv_post output); fade_n>0 applies S3Gen trim_fade. */
int cbx_hift_istft_f32(const float* x, float* wav, int B, long frames, long ldx, float clamp, int fade_n,
                       void* stream);

/* ---- voice-prompt / voice-conversion front-end ----
 * Contractions (framed DFT as a GEMM over overlapping waveform rows, mel filterbanks, Conv1d/Conv2d-as-Toeplitz, attention, LSTM
 * projections) use cbx_gemm_f32 / cbx_flash_attn_f32 / cbx_gemv_f32; these are the remaining element-wise / reduction passes. */
/* depthwise Conv1d, channel-last: y[b][t][c] = sum_k w[c][k] x[b][t+k-pad_left][c] (+ x[b][t][c]); rows >= lens[b] are zero.
 * S3TokenizerV2 FSMN memory block (third-party s3tokenizer.model_v2, Conv1d k31 groups=C; parity unpinned). */
int cbx_dwconv1d_f32(const float* x, const float* w, float* y, const int* lens, int B, int T, int C, int taps, int pad_left,
                     long ldx, long ldy, long x_sb, long y_sb, int add_input, void* stream);
/* torch.nn.LSTM cell, gates (i,f,g,o) = pre + hh (voice_encoder.py:139-163: nn.LSTM(40, 256, 3 layers)) */
int cbx_lstm_cell_f32(const float* pre, const float* hh, float* c, float* h, int B, int H, long ld_pre, long ld_hh, long ldc,
                      long ldh, void* stream);
/* y = act(x * scale[c] + shift[c]): eval BatchNorm + ReLU that PRECEDES a conv in CAMPPlus (xvector.py:136-151,262-266) */
int cbx_affine_act_f32(const float* x, float* y, const float* scale, const float* shift, long rows, int C, long ldx, long ldy,
                       int act, void* stream);
/* spec row [re(0..F-1) | im(0..F-1)] -> |.|^2 (mode 0: s3tokenizer.py:161, voice_encoder/melspec.py:35-39) or
 * sqrt(|.|^2 + eps) (mode 1: s3gen/utils/mel.py:80) */
int cbx_cplx_power_f32(const float* spec, float* out, long rows, int F, long ld_spec, long ld_out, int mode, float eps, void* stream);
#define CBX_UN_LOG_CLAMP 1     /* log(max(x, a))                 utils/mel.py:18-19 */
#define CBX_UN_LOG10_CLAMP 2   /* log10(max(x, a))               s3tokenizer.py:165 */
#define CBX_UN_FLOOR_AFFINE 3  /* (max(x, *dev_scalar - a) + b) / b   s3tokenizer.py:166-167 */
#define CBX_UN_AFFINE 4        /* a x + b */
int cbx_unary_f32(const float* x, float* y, long rows, int C, long ldx, long ldy, int op, float a, float b, const float* dev_scalar,
                  void* stream);
/* out[0] = max over a 2-D strided block (log_spec.max() of s3tokenizer.py:166, kept on the device as CBX_UN_FLOOR_AFFINE's dev_scalar) */
int cbx_reduce_max_f32(const float* x, float* out, long rows, int C, long ldx, void* stream);
/* CAMLayer context (xvector.py:204-231): ctx[s][c] = mean_t x + mean over segment s (avg_pool1d ceil_mode), s = t / seg_len */
int cbx_seg_context_f32(const float* x, float* ctx, int T, int C, int seg_len, long ldx, long ldc, void* stream);
/* y[t][c] *= sigmoid(m[t / seg_len][c])  (xvector.py:209-213) */
int cbx_seg_gate_mul_f32(float* y, const float* m, int T, int C, int seg_len, long ldy, long ldm, void* stream);
/* StatsPool (xvector.py:153-165): out = [mean_t | unbiased std_t] */
int cbx_stats_pool_f32(const float* x, float* out, int T, int C, long ldx, void* stream);
/* FSQ codebook index of S3TokenizerV2 (third-party; restated: tanh * 0.999 -> round -> +1 -> base-3 digits) */
int cbx_fsq_index(const float* h, long long* idx, long rows, long ldh, void* stream);

#ifdef __cplusplus
}
#endif
#endif
