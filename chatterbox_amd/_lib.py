"""ctypes binding of libcbx_hip.so (the C ABI declared in include/cbx.h).

The product path has NO fallback: if the shared library is missing or an entry point is absent, importing this
module raises.  torch is imported first so that the already-loaded ROCm runtime (libamdhip64.so.7) is the one
our library binds to -- pointers and streams are then interchangeable with torch's.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede CDLL: shares the HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CBX_LIB_PATH") or os.path.join(_HERE, "libcbx_hip.so")  # CBX_LIB_PATH: developer builds (scripts/diag_gemm.sh)

c_f = ctypes.c_void_p
c_int, c_long, c_float = ctypes.c_int, ctypes.c_long, ctypes.c_float


ABI_VERSION = 16  # include/cbx.h CBX_ABI_VERSION


class GemmParams(ctypes.Structure):
    _fields_ = [
        ("A", c_f), ("W", c_f), ("C", c_f), ("bias", c_f), ("R", c_f), ("C2", c_f),
        ("act1_param", c_f), ("act2_param", c_f), ("lens", c_f),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("Cin", c_int), ("taps", c_int), ("dil", c_int), ("stride", c_int), ("pad_left", c_int), ("up", c_int),
        ("Tin", c_int), ("nz1", c_int), ("nz2", c_int), ("w_kn", c_int), ("swiglu", c_int),
        ("act1", c_int), ("act2", c_int),
        ("act1_slope", c_float), ("act2_slope", c_float), ("alpha", c_float), ("beta", c_float),
        ("lda", c_long), ("a_s1", c_long), ("a_s2", c_long),
        ("ldw", c_long), ("w_s1", c_long), ("w_s2", c_long),
        ("ldc", c_long), ("c_s1", c_long), ("c_s2", c_long),
        ("ldr", c_long), ("r_s1", c_long), ("r_s2", c_long),
        ("ldc2", c_long), ("c2_s1", c_long), ("c2_s2", c_long),
        ("precision", c_int), ("reserved0", c_int), ("ln_stats", c_f), ("ln_w", c_f), ("ln_b", c_f),
    ]


class GemmPlParams(ctypes.Structure):
    _fields_ = [
        ("A", c_f), ("W", c_f), ("C", c_f), ("P", c_f), ("bias", c_f), ("R", c_f), ("act_param", c_f), ("lens", c_f),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("Cin", c_int), ("taps", c_int), ("dil", c_int), ("stride", c_int), ("pad_left", c_int), ("Tin", c_int), ("nz1", c_int),
        ("act", c_int), ("act_slope", c_float), ("alpha", c_float),
        ("lda", c_long), ("a_lo", c_long), ("a_s1", c_long), ("ldw", c_long), ("w_lo", c_long), ("w_s1", c_long),
        ("ldc", c_long), ("c_s1", c_long), ("ldr", c_long), ("r_s1", c_long), ("ldp", c_long), ("p_lo", c_long), ("p_s1", c_long), ("reserved0", c_int),
        ("PT", c_f), ("pt_n0", c_int), ("pt_T", c_int), ("pt_ld", c_long), ("pt_lo", c_long), ("pt_zs", c_long),  # ABI v8
        ("tile", c_int),  # ABI v13: this call's tile form (0 automatic, -1 = PL_TILE_CORESIDENT)
        ("ln_w", c_f), ("ln_b", c_f), ("LNP", c_f), ("ld_lnp", c_long), ("lnp_lo", c_long), ("lnp_s1", c_long), ("ln_eps", c_float),  # ABI v13: LayerNorm epilogue
    ]


class GemvParams(ctypes.Structure):
    _fields_ = [("x", c_f), ("W", c_f), ("bias", c_f), ("out", c_f), ("M", c_int), ("N", c_int), ("K", c_int),
                ("ksplit", c_int), ("nw", c_int), ("swiglu", c_int), ("act", c_int), ("ldx", c_long), ("ldw", c_long), ("ldo", c_long),
                ("part_stride", c_long), ("w_packed", c_int), ("x_packed", c_int), ("half_tile", c_int), ("out_packed", c_int),
                ("norm_w", c_f), ("res", c_f), ("eps", c_float), ("n_xpart", c_int), ("xpart", c_f), ("xpart_stride", c_long), ("x_out", c_f), ("w_bf16", c_int), ("flags", c_int), ("ln_cw", c_f), ("ln_cb", c_f),
                ("col_tiles", c_int), ("ssq_out", c_f)]  # ABI v11


GEMV_PRE_EPI, GEMV_DEEP, GEMV_SHALLOW = 1, 2, 4  # cbx_gemv_t.flags
PL_TILE_CORESIDENT, ATTN_PL_CORESIDENT = -1, 5  # cbx_gemm_pl_t.tile / cbx_flash_attn_planes_v version: the co-resident forms (ABI v13)


class DecodeAttnParams(ctypes.Structure):  # cbx_decode_attn_t (ABI v10)
    _fields_ = [("qkv", c_f), ("positions", c_f), ("cos_t", c_f), ("sin_t", c_f), ("kc", c_f), ("vc", c_f), ("o", c_f),
                ("rows", c_int), ("n_heads", c_int), ("ld_qkv", c_long), ("o_ld", c_long), ("o_packed", c_int),
                ("cache_row_stride", c_long), ("cache_head_stride", c_long), ("scale", c_float),
                ("unroll", c_int), ("pipeline", c_int), ("split_min", c_int), ("split_ws", c_f), ("split_cnt", c_f), ("split_pairs", c_long),
                ("qkv_nparts", c_int), ("qkv_part_stride", c_long), ("qkv_ssq", c_f), ("rms_dim", c_int), ("rms_eps", c_float)]  # ABI v11


class GemvRowParams(ctypes.Structure):  # cbx_gemv_row_t (ABI v14)
    _fields_ = [("x", c_f), ("W", c_f), ("bias", c_f), ("res", c_f), ("out", c_f), ("ln_w", c_f), ("ln_b", c_f), ("eps", c_float),
                ("parts", c_f), ("n_parts", c_int), ("n_heads", c_int), ("N", c_int), ("K", c_int), ("ldw", c_long), ("act", c_int),
                ("rows_per_wave", c_int), ("M", c_int), ("ldx", c_long), ("ldo", c_long), ("ldr", c_long), ("parts_row_stride", c_long)]


class AttnPartsParams(ctypes.Structure):  # cbx_attn_parts_t (ABI v14)
    _fields_ = [("qkv", c_f), ("positions", c_f), ("cos_t", c_f), ("sin_t", c_f), ("kc", c_f), ("vc", c_f), ("parts", c_f),
                ("rows", c_int), ("n_heads", c_int), ("n_splits", c_int), ("chunks", c_int), ("max_ctx", c_int),
                ("ld_qkv", c_long), ("cache_row_stride", c_long), ("cache_head_stride", c_long), ("scale", c_float)]


ATTN_PART_REC = 68  # CBX_ATTN_PART_REC


class SamplerParams(ctypes.Structure):
    _fields_ = [
        ("logits", c_f), ("ld", c_long), ("V", c_int), ("B", c_int), ("cfg", c_int),
        ("cfg_weight", c_float), ("temperature", c_float), ("min_p", c_float), ("top_p", c_float),
        ("rep_penalty", c_float), ("top_k", c_int), ("order", c_int), ("ban_token", c_int), ("eos_token", c_int), ("ban_from", c_int),
        ("seen", c_f), ("uniforms", c_f), ("max_steps", c_int), ("step", c_f), ("out_tokens", c_f),
        ("done", c_f), ("n_generated", c_f), ("next_ids", c_f), ("next_pos_ids", c_f), ("positions", c_f),
        ("ctx_lens", c_f), ("dev_params", c_f),
    ]


class T3Layer(ctypes.Structure):
    _fields_ = [("ln1", c_f), ("ln2", c_f), ("wqkv", c_f), ("wo", c_f), ("wgu", c_f), ("wd", c_f)]


class T3Step(ctypes.Structure):
    _fields_ = [("n_layers", c_int), ("rows", c_int), ("dim", c_int), ("ffn", c_int), ("n_heads", c_int), ("vocab", c_int),
                ("o_nw", c_int), ("gu_nw", c_int), ("d_nw", c_int), ("d_ksplit", c_int), ("half_tiles", c_int), ("w_bf16", c_int),
                ("eps", c_float), ("attn_scale", c_float),
                ("layers", ctypes.POINTER(T3Layer)), ("speech_emb", c_f), ("speech_pos", c_f), ("final_norm", c_f), ("head", c_f),
                ("cos_t", c_f), ("sin_t", c_f), ("kc", c_f), ("vc", c_f), ("kv_row_stride", c_long), ("kv_head_stride", c_long),
                ("next_ids", c_f), ("next_pos_ids", c_f), ("positions", c_f), ("x_a", c_f), ("x_b", c_f), ("qkv", c_f), ("att", c_f),
                ("g", c_f), ("pd", c_f), ("logits", c_f), ("ld_logits", c_long), ("sampler", ctypes.POINTER(SamplerParams)),
                ("qkv_tile", c_int),  # ABI v9
                ("da_unroll", c_int), ("da_pipeline", c_int), ("da_split_min", c_int), ("gemv_flags", c_int),  # ABI v10
                ("da_ws", c_f), ("da_cnt", c_f), ("da_pairs", c_long),
                ("qkv_ksplit", c_int), ("qkv_ct", c_int), ("head_ct", c_int), ("qkv_ssq", c_f)]  # ABI v11


class T3Prefill(ctypes.Structure):  # cbx_t3_prefill_t
    _fields_ = [("n_layers", c_int), ("rows", c_int), ("S", c_int), ("dim", c_int), ("ffn", c_int), ("n_heads", c_int), ("precision", c_int),
                ("eps", c_float), ("attn_scale", c_float), ("layers", ctypes.POINTER(T3Layer)), ("x", c_f), ("h", c_f), ("qkv", c_f), ("att", c_f), ("g", c_f),
                ("positions", c_f), ("cache_rows", c_f), ("cos_t", c_f), ("sin_t", c_f), ("kc", c_f), ("vc", c_f),
                ("kv_layer_stride", c_long), ("kv_row_stride", c_long), ("kv_head_stride", c_long)]


class Gpt2Layer(ctypes.Structure):  # cbx_gpt2_layer_t (ABI v16)
    _fields_ = [(k, c_f) for k in ("ln1_w", "ln1_b", "ln2_w", "ln2_b", "wqkv", "bqkv", "wo", "bo", "wfc", "bfc", "wpr", "bpr")]


class Gpt2Prefill(ctypes.Structure):  # cbx_gpt2_prefill_t (ABI v16)
    _fields_ = [("n_layers", c_int), ("rows", c_int), ("S", c_int), ("prefix", c_int), ("dim", c_int), ("n_heads", c_int), ("eps", c_float), ("attn_scale", c_float),
                ("layers", ctypes.POINTER(Gpt2Layer)), ("x", c_f), ("h", c_f), ("qkv", c_f), ("att", c_f), ("g", c_f), ("positions", c_f), ("cache_rows", c_f),
                ("kc", c_f), ("vc", c_f), ("kv_layer_stride", c_long), ("kv_row_stride", c_long), ("kv_head_stride", c_long)]


class PlanesRef(ctypes.Structure):  # cbx_planes_t (ABI v12)
    _fields_ = [("p", c_f), ("ld", c_long), ("lo", c_long)]


class CfmTBlock(ctypes.Structure):  # cbx_cfm_tblock_t
    _fields_ = [(k, c_f) for k in ("n1_w", "n1_b", "n3_w", "n3_b", "bo", "b1", "b2")] + [(k, PlanesRef) for k in ("wqkv", "wo", "w1", "w2")]


class CfmStage(ctypes.Structure):  # cbx_cfm_stage_t
    _fields_ = ([(k, PlanesRef) for k in ("c1", "c2", "res", "tail")]
                + [(k, c_f) for k in ("c1_b", "n1_w", "n1_b", "c2_b", "n2_w", "n2_b", "res_b", "tail_b")]
                + [("cin", c_int), ("n_tb", c_int), ("tb", ctypes.POINTER(CfmTBlock))])


class CfmSolve(ctypes.Structure):  # cbx_cfm_t
    _fields_ = ([(k, c_int) for k in ("n_stages", "rows", "B", "n_steps", "cfg", "fused_qkv", "fused_ln")]
                + [("T", c_long), ("cfg_rate", c_float), ("dt", ctypes.POINTER(c_float)), ("stages", ctypes.POINTER(CfmStage)),
                   ("fin_c", PlanesRef), ("fin_proj", PlanesRef)]
                + [(k, c_f) for k in ("fin_c_b", "fin_n_w", "fin_n_b", "fin_proj_b", "tbias", "lens", "xin")]
                + [("xinP", PlanesRef)] + [(k, c_f) for k in ("ra", "rb", "x", "v")]
                + [(k, PlanesRef) for k in ("aP", "hP", "qkP", "attP", "ffP", "xP", "yP", "catP", "vtP")]
                + [("gemm_tile", c_int), ("attn_version", c_int)])  # ABI v13


class Conformer(ctypes.Structure):  # cbx_conformer_t
    _fields_ = [(k, c_f) for k in ("ln_mha_w", "ln_mha_b", "w4", "b4", "wpos", "wo", "bo", "ln_ff_w", "ln_ff_b", "w1", "b1", "w2", "b2")]


class S3Encode(ctypes.Structure):  # cbx_s3enc_t
    _fields_ = ([(k, c_int) for k in ("B", "N", "n_enc", "n_up", "precision")] + [(k, c_f) for k in ("ids", "lens", "lens2")]
                + [(k, c_f) for k in ("emb", "e_w", "e_b", "e_lnw", "e_lnb", "u_w", "u_b", "u_lnw", "u_lnb", "pl1_w", "pl1_b", "pl2_w", "pl2_b", "up_w", "up_b",
                                      "after_w", "after_b", "proj_w", "proj_b")]
                + [("enc", ctypes.POINTER(Conformer)), ("up_enc", ctypes.POINTER(Conformer)), ("pe", c_f), ("pe2", c_f)]
                + [(k, c_f) for k in ("x0", "xa", "y1", "x2", "xu", "xb", "h", "q4", "pp", "att", "ff", "mu")])


class HiftF0(ctypes.Structure):  # cbx_hift_f0_t
    _fields_ = [("B", c_int), ("T", c_long), ("mel", c_f), ("lens", c_f), ("f0_w", c_f * 5), ("f0_b", c_f * 5), ("cls_w", c_f), ("cls_b", c_f), ("src_w", c_f),
                ("src_b", c_float), ("phase", c_f), ("noise", c_f), ("buf0", c_f), ("buf1", c_f), ("f0", c_f), ("s", c_f), ("cum", c_f)]


class HiftResblock(ctypes.Structure):  # cbx_hift_resblock_t
    _fields_ = [(k, c_f * 3) for k in ("c1_w", "c1_b", "c2_w", "c2_b", "a1", "a2")]


class HiftDecode(ctypes.Structure):  # cbx_hift_t
    _fields_ = ([("B", c_int), ("precision", c_int), ("fade", c_int), ("T", c_long), ("mel", c_f), ("s", c_f), ("wav", c_f), ("lens", c_f),
                 ("conv_pre_w", c_f), ("conv_pre_b", c_f), ("ups_w", c_f * 3), ("ups_b", c_f * 3), ("src_down_w", c_f * 3), ("src_down_b", c_f * 3),
                 ("conv_post_w", c_f), ("conv_post_b", c_f), ("src_rb", HiftResblock * 3), ("rb", HiftResblock * 9)]
                + [(k, c_f) for k in ("spec", "post", "x0", "xs", "t1", "xa", "xb", "an", "si", "sa", "acc", "a0")] + [("nxt", c_f * 2)])


_SIGS = {
    "cbx_abi_version": ([], c_int),
    "cbx_last_error": ([], ctypes.c_char_p),
    "cbx_gemm_f32": ([ctypes.POINTER(GemmParams), c_f], c_int),
    "cbx_gemv_f32": ([ctypes.POINTER(GemvParams), c_f], c_int),
    "cbx_gemv_row_f32": ([ctypes.POINTER(GemvRowParams), c_f], c_int),
    "cbx_decode_attn_parts": ([ctypes.POINTER(AttnPartsParams), c_f], c_int),
    "cbx_set_gemv_deep_batches": ([c_int], c_int),
    "cbx_set_gemv_epilogue_prefetch": ([c_int], c_int),
    "cbx_pack_gemv_weight_f32": ([c_f, c_f, c_int, c_int, c_long, c_int, c_f], c_int),
    "cbx_pack_gemv_weight_bf16": ([c_f, c_f, c_int, c_int, c_long, c_int, c_f], c_int),
    "cbx_add_norm_f32": ([c_f, c_f, c_int, c_long, c_long, c_f, c_f, c_f, c_int, c_int, c_long, c_long, c_float, c_int, c_f], c_int),
    "cbx_add_rmsnorm_f32": ([c_f, c_f, c_int, c_long, c_long, c_f, c_f, c_int, c_int, c_long, c_long, c_float, c_f], c_int),
    "cbx_layernorm_f32": ([c_f, c_f, c_f, c_f, c_f, c_long, c_int, c_long, c_long, c_float, c_int, c_int, c_float, c_f], c_int),
    "cbx_flash_relpos_f32": ([c_f] * 7 + [c_int] * 3 + [c_long] * 5 + [c_float, c_f], c_int),
    "cbx_flash_attn_f32": ([c_f, c_f, c_f, c_f, c_f, c_int, c_int, c_int, c_int] + [c_long] * 8 + [c_float, c_int, c_f], c_int),
    "cbx_flash_attn_kv_f32": ([c_f, c_f, c_f, c_f, c_f, c_int, c_int, c_int, c_int] + [c_long] * 10 + [c_float, c_int, c_f], c_int),
    "cbx_flash_attn_split_f32": ([c_f, c_f, c_f, c_f, c_f, c_int, c_int, c_int, c_int] + [c_long] * 8 + [c_float, c_int, c_int, c_f], c_int),
    "cbx_decode_attn_f32": ([c_f, c_f, c_f, c_f, c_f, c_int, c_int, c_long, c_long, c_long, c_long, c_float, c_f], c_int),
    "cbx_decode_attn_rope_f32": ([c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_int, c_int, c_long, c_long, c_int, c_long, c_long, c_float, c_f], c_int),
    "cbx_decode_attn_rope": ([ctypes.POINTER(DecodeAttnParams), c_f], c_int),
    "cbx_set_decode_attn_unroll": ([c_int], c_int),
    "cbx_set_decode_attn_pipeline": ([c_int], c_int),
    "cbx_set_decode_attn_workspace": ([c_f, c_f, c_long], c_int),
    "cbx_set_decode_attn_split_min": ([c_int], c_int),
    "cbx_set_split_tile": ([c_int], c_int),
    "cbx_set_range_flag": ([c_f], c_int),
    "cbx_gemm_ln_fusable": ([c_long, c_int, c_long], c_int),
    "cbx_gemm_planes": ([ctypes.POINTER(GemmPlParams), c_f], c_int),
    "cbx_set_planes_tile": ([c_int], c_int),
    "cbx_set_planes_persist": ([c_int], c_int),
    "cbx_split_planes_f32": ([c_f, c_f, c_long, c_int, c_long, c_long, c_long, c_f], c_int),
    "cbx_layernorm_planes_f32": ([c_f, c_f, c_f, c_f, c_f, c_long, c_int, c_long, c_long, c_long, c_float, c_int, c_float, c_f], c_int),
    "cbx_flash_attn_split_po": ([c_f, c_f, c_f, c_f, c_f, c_int, c_int, c_int, c_int] + [c_long] * 9 + [c_float, c_int, c_f], c_int),
    "cbx_flash_attn_planes": ([c_f, c_f, c_f, c_f, c_f, c_int, c_int, c_int, c_int] + [c_long] * 12 + [c_float, c_int, c_f], c_int),
    "cbx_set_stream_coresident": ([c_f, c_int], c_int),
    "cbx_flash_attn_planes_v": ([c_f, c_f, c_f, c_f, c_f, c_int, c_int, c_int, c_int] + [c_long] * 12 + [c_float, c_int, c_int, c_f], c_int),
    "cbx_set_attn_planes_version": ([c_int], c_int),
    "cbx_row_stats_f32": ([c_f, c_f, c_long, c_int, c_long, c_float, c_f], c_int),
    "cbx_softmax_relpos_f32": ([c_f, c_f, c_f, c_f, c_int, c_int, c_int, c_int] + [c_long] * 6 + [c_float, c_f], c_int),
    "cbx_act_f32": ([c_f, c_f, c_f, c_long, c_int, c_long, c_long, c_int, c_float, c_f], c_int),
    "cbx_axpby_f32": ([c_f, c_f, c_long, c_int, c_long, c_long, c_float, c_float, c_f], c_int),
    "cbx_embed_f32": ([c_f, c_f, c_f, c_f, c_f, c_long, c_int, c_long, c_float, c_int, c_f], c_int),
    "cbx_rope_kv_f32": ([c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_long, c_int, c_long, c_long, c_long, c_f], c_int),
    "cbx_cfm_euler_f32": ([c_f, c_f, c_int, c_long, c_int, c_long, c_long, c_long, c_long, c_float, c_float, c_int, c_f], c_int),
    "cbx_dwconv1d_f32": ([c_f, c_f, c_f, c_f, c_int, c_int, c_int, c_int, c_int, c_long, c_long, c_long, c_long, c_int, c_f], c_int),
    "cbx_lstm_cell_f32": ([c_f, c_f, c_f, c_f, c_int, c_int, c_long, c_long, c_long, c_long, c_f], c_int),
    "cbx_affine_act_f32": ([c_f, c_f, c_f, c_f, c_long, c_int, c_long, c_long, c_int, c_f], c_int),
    "cbx_cplx_power_f32": ([c_f, c_f, c_long, c_int, c_long, c_long, c_int, c_float, c_f], c_int),
    "cbx_unary_f32": ([c_f, c_f, c_long, c_int, c_long, c_long, c_int, c_float, c_float, c_f, c_f], c_int),
    "cbx_reduce_max_f32": ([c_f, c_f, c_long, c_int, c_long, c_f], c_int),
    "cbx_seg_context_f32": ([c_f, c_f, c_int, c_int, c_int, c_long, c_long, c_f], c_int),
    "cbx_seg_gate_mul_f32": ([c_f, c_f, c_int, c_int, c_int, c_long, c_long, c_f], c_int),
    "cbx_stats_pool_f32": ([c_f, c_f, c_int, c_int, c_long, c_f], c_int),
    "cbx_fsq_index": ([c_f, c_f, c_long, c_long, c_f], c_int),
    "cbx_t3_decode_step": ([ctypes.POINTER(T3Step), c_f], c_int),
    "cbx_t3_prefill": ([ctypes.POINTER(T3Prefill), c_f], c_int),
    "cbx_gpt2_prefill": ([ctypes.POINTER(Gpt2Prefill), c_f], c_int),
    "cbx_t3_loop_create": ([ctypes.POINTER(T3Step), c_f, ctypes.POINTER(c_f)], c_int),
    "cbx_t3_loop_run": ([c_f, c_int, c_int, c_f, ctypes.POINTER(c_int)], c_int),
    "cbx_t3_loop_destroy": ([c_f], c_int),
    "cbx_t3_sample": ([ctypes.POINTER(SamplerParams), c_f], c_int),
    "cbx_cfm_solve": ([ctypes.POINTER(CfmSolve), c_f], c_int),
    "cbx_s3gen_encode": ([ctypes.POINTER(S3Encode), c_f], c_int),
    "cbx_hift_f0_source": ([ctypes.POINTER(HiftF0), c_f], c_int),
    "cbx_hift_decode": ([ctypes.POINTER(HiftDecode), c_f], c_int),
    "cbx_hift_source_f32": ([c_f, c_f, c_f, c_f, c_float, c_f, c_f, c_int, c_int, c_int, c_float, c_f], c_int),
    "cbx_hift_stft_f32": ([c_f, c_f, c_f, c_int, c_long, c_long, c_f], c_int),
    "cbx_hift_istft_f32": ([c_f, c_f, c_int, c_long, c_long, c_float, c_int, c_f], c_int),
}


class CbxError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m chatterbox_amd.build` (hipcc --offload-arch=gfx950). "
            "chatterbox_amd has no CPU / eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (args, res) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"libcbx_hip.so does not export {name} (stale build?)") from e
        fn.argtypes = args
        fn.restype = res
    if lib.cbx_abi_version() != ABI_VERSION:
        raise ImportError("libcbx_hip.so ABI version mismatch")
    return lib


lib = _load()


def check(rc, what=""):
    if rc != 0:
        raise CbxError(f"{what}: rc={rc}: {lib.cbx_last_error().decode(errors='replace')}")
