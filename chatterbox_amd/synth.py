"""Seeded synthetic checkpoints and inputs (there is no network: no pretrained weights).

`t3_state_dict()` / `s3gen_state_dict()` emit tensors under the *reference's own
state-dict key layout* (verified against the reference
constructors by tests/golden/make_golden.py), so the very same dict loads into
the unmodified reference modules (`load_state_dict(strict=True)` modulo the
out-of-scope `tokenizer.*` / `speaker_encoder.*` prefixes) and into our loader.

Every tensor is drawn from its own CPU generator seeded by crc32(key) ^ seed, so
a 2-layer model is a strict prefix of the 30-layer one and generation order
does not matter.  The synthetic-input recipe follows SURVEY.md section 8(d).
"""
import math
import zlib

import torch

# ----------------------------------------------------------------------------- primitives


def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def _normal(key, shape, std, seed, mean=0.0):
    return torch.randn(shape, generator=_gen(key, seed), dtype=torch.float32) * std + mean


def _uniform(key, shape, bound, seed):
    return (torch.rand(shape, generator=_gen(key, seed), dtype=torch.float32) * 2 - 1) * bound


def _linear(sd, name, out_f, in_f, seed, bias=True, gain=1.0):
    b = gain / math.sqrt(in_f)
    sd[name + ".weight"] = _uniform(name + ".weight", (out_f, in_f), b * math.sqrt(3.0), seed)
    if bias:
        sd[name + ".bias"] = _uniform(name + ".bias", (out_f,), 0.1, seed)


def _norm(sd, name, dim, seed, bias=True):
    sd[name + ".weight"] = 1.0 + _normal(name + ".weight", (dim,), 0.1, seed)
    if bias:
        sd[name + ".bias"] = _normal(name + ".bias", (dim,), 0.1, seed)


def _conv(sd, name, out_c, in_c, k, seed, gain=1.0):
    b = gain / math.sqrt(in_c * k)
    sd[name + ".weight"] = _uniform(name + ".weight", (out_c, in_c, k), b * math.sqrt(3.0), seed)
    sd[name + ".bias"] = _uniform(name + ".bias", (out_c,), 0.1, seed)


def _wn_conv(sd, name, dim0, dim1, k, seed, gain=1.0, fan_in=None):
    """weight_norm parametrization keys: original0 = g (dim0,1,1), original1 = v (dim0,dim1,k)."""
    fan_in = fan_in or dim1 * k
    b = gain / math.sqrt(fan_in)
    v = _uniform(name + ".v", (dim0, dim1, k), b * math.sqrt(3.0), seed)
    g = v.flatten(1).norm(dim=1).view(dim0, 1, 1) * (1.0 + _normal(name + ".g", (dim0, 1, 1), 0.1, seed))
    sd[name + ".parametrizations.weight.original0"] = g
    sd[name + ".parametrizations.weight.original1"] = v


# ----------------------------------------------------------------------------- T3 (Llama backbone)

T3_MTL = dict(dim=1024, inter=4096, n_layers=30, n_heads=16, head_dim=64, text_vocab=2454,
              speech_vocab=8194, max_text_pos=2048 + 2, max_speech_pos=4096 + 4, spk_dim=256,
              n_query=32, perceiver_heads=4)


def t3_state_dict(n_layers=30, seed=0, text_vocab=2454):
    """Keys of `T3(T3Config.multilingual())` (reference src/chatterbox/models/t3/t3.py:49-86)."""
    c = dict(T3_MTL, n_layers=n_layers, text_vocab=text_vocab)
    d, f = c["dim"], c["inter"]
    sd = {}
    sd["tfmr.embed_tokens.weight"] = _normal("tfmr.embed_tokens.weight", (8, d), 0.02, seed)
    for i in range(n_layers):
        p = f"tfmr.layers.{i}."
        for nm, (o, k) in dict(q_proj=(d, d), k_proj=(d, d), v_proj=(d, d), o_proj=(d, d)).items():
            sd[p + f"self_attn.{nm}.weight"] = _normal(p + f"self_attn.{nm}.weight", (o, k), 0.03, seed)
        for nm, (o, k) in dict(gate_proj=(f, d), up_proj=(f, d), down_proj=(d, f)).items():
            sd[p + f"mlp.{nm}.weight"] = _normal(p + f"mlp.{nm}.weight", (o, k), 0.03, seed)
        _norm(sd, p + "input_layernorm", d, seed, bias=False)
        _norm(sd, p + "post_attention_layernorm", d, seed, bias=False)
    _norm(sd, "tfmr.norm", d, seed, bias=False)
    _linear(sd, "cond_enc.spkr_enc", d, c["spk_dim"], seed)
    sd["cond_enc.emotion_adv_fc.weight"] = _normal("cond_enc.emotion_adv_fc.weight", (d, 1), 1.0, seed)
    sd["cond_enc.perceiver.pre_attention_query"] = _uniform("cond_enc.perceiver.pre_attention_query",
                                                            (1, c["n_query"], d), 0.3, seed)
    _norm(sd, "cond_enc.perceiver.attn.norm", d, seed)
    for nm in ("to_q", "to_k", "to_v", "proj_out"):
        _linear(sd, f"cond_enc.perceiver.attn.{nm}", d, d, seed)
    sd["text_emb.weight"] = _normal("text_emb.weight", (c["text_vocab"], d), 1.0, seed)
    sd["speech_emb.weight"] = _normal("speech_emb.weight", (c["speech_vocab"], d), 1.0, seed)
    sd["text_pos_emb.emb.weight"] = _normal("text_pos_emb.emb.weight", (c["max_text_pos"], d), 0.02, seed)
    sd["speech_pos_emb.emb.weight"] = _normal("speech_pos_emb.emb.weight", (c["max_speech_pos"], d), 0.02, seed)
    sd["text_head.weight"] = _normal("text_head.weight", (c["text_vocab"], d), 0.05, seed)
    # a sharper-than-default head so that sampling is not uniform over 8194 ids
    sd["speech_head.weight"] = _normal("speech_head.weight", (c["speech_vocab"], d), 0.08, seed)
    return sd


def t3_turbo_state_dict(n_layers=24, d=1024, seed=0, include_wte=False):
    """Keys of the Turbo/Nano `T3` (GPT2-medium d=1024/24L, GPT2-small d=768/12L; reference tts_turbo.py:153-167).
    HF Conv1D weights are [in, out].  `tfmr.wte` exists in the checkpoint files and is deleted after loading."""
    sd = {}
    sd["tfmr.wpe.weight"] = _normal("tfmr.wpe.weight", (8196, d), 0.02, seed)
    if include_wte:
        sd["tfmr.wte.weight"] = _normal("tfmr.wte.weight", (50276, d), 0.02, seed)
    for i in range(n_layers):
        p = f"tfmr.h.{i}."
        _norm(sd, p + "ln_1", d, seed)
        _norm(sd, p + "ln_2", d, seed)
        for nm, (i_f, o_f) in dict(**{"attn.c_attn": (d, 3 * d), "attn.c_proj": (d, d), "mlp.c_fc": (d, 4 * d), "mlp.c_proj": (4 * d, d)}).items():
            sd[p + nm + ".weight"] = _normal(p + nm + ".weight", (i_f, o_f), 0.03, seed)
            sd[p + nm + ".bias"] = _uniform(p + nm + ".bias", (o_f,), 0.05, seed)
    _norm(sd, "tfmr.ln_f", d, seed)
    _linear(sd, "cond_enc.spkr_enc", d, 256, seed)
    sd["text_emb.weight"] = _normal("text_emb.weight", (50276, d), 1.0, seed)
    sd["speech_emb.weight"] = _normal("speech_emb.weight", (6563, d), 1.0, seed)
    sd["text_head.weight"] = _normal("text_head.weight", (50276, d), 0.05, seed)
    sd["speech_head.weight"] = _normal("speech_head.weight", (6563, d), 0.08, seed)
    sd["speech_head.bias"] = _uniform("speech_head.bias", (6563,), 0.1, seed)
    return sd


def turbo_text_tokens(n=64, seed=1):
    """GPT-2 BPE ids uniform in [0, 50257) (no SOT/EOT are added on the Turbo path, tts_turbo.py:295-296)."""
    return torch.randint(0, 50257, (n,), generator=torch.Generator().manual_seed(1500 + seed)).long()


# ----------------------------------------------------------------------------- S3Gen (flow + HiFT)


def _conformer_layer(sd, p, seed):
    for nm in ("linear_q", "linear_k", "linear_v", "linear_out"):
        _linear(sd, p + "self_attn." + nm, 512, 512, seed)
    _linear(sd, p + "self_attn.linear_pos", 512, 512, seed, bias=False)
    sd[p + "self_attn.pos_bias_u"] = _uniform(p + "self_attn.pos_bias_u", (8, 64), 0.3, seed)
    sd[p + "self_attn.pos_bias_v"] = _uniform(p + "self_attn.pos_bias_v", (8, 64), 0.3, seed)
    _linear(sd, p + "feed_forward.w_1", 2048, 512, seed)
    _linear(sd, p + "feed_forward.w_2", 512, 2048, seed)
    _norm(sd, p + "norm_ff", 512, seed)
    _norm(sd, p + "norm_mha", 512, seed)


def _cfm_stage(sd, p, cin, seed, tail):
    _linear(sd, p + "0.mlp.1", 256, 1024, seed)
    _conv(sd, p + "0.block1.block.0", 256, cin, 3, seed)
    _norm(sd, p + "0.block1.block.2", 256, seed)
    _conv(sd, p + "0.block2.block.0", 256, 256, 3, seed)
    _norm(sd, p + "0.block2.block.2", 256, seed)
    _conv(sd, p + "0.res_conv", 256, cin, 1, seed)
    for j in range(4):
        q = p + f"1.{j}."
        _norm(sd, q + "norm1", 256, seed)
        for nm in ("to_q", "to_k", "to_v"):
            _linear(sd, q + "attn1." + nm, 512, 256, seed, bias=False)
        _linear(sd, q + "attn1.to_out.0", 256, 512, seed)
        _norm(sd, q + "norm3", 256, seed)
        _linear(sd, q + "ff.net.0.proj", 1024, 256, seed)
        _linear(sd, q + "ff.net.2", 256, 1024, seed)
    if tail:
        _conv(sd, p + "2", 256, 256, 3, seed)


def _hift_resblock(sd, p, ch, k, seed):
    for j in range(3):
        _wn_conv(sd, p + f"convs1.{j}", ch, ch, k, seed)
        sd[p + f"convs1.{j}.bias"] = _uniform(p + f"convs1.{j}.bias", (ch,), 0.05, seed)
        _wn_conv(sd, p + f"convs2.{j}", ch, ch, k, seed)
        sd[p + f"convs2.{j}.bias"] = _uniform(p + f"convs2.{j}.bias", (ch,), 0.05, seed)
        sd[p + f"activations1.{j}.alpha"] = 1.0 + _normal(p + f"activations1.{j}.alpha", (ch,), 0.2, seed)
        sd[p + f"activations2.{j}.alpha"] = 1.0 + _normal(p + f"activations2.{j}.alpha", (ch,), 0.2, seed)


def s3gen_state_dict(seed=0, meanflow=False, n_mid=12, n_enc=6, n_up_enc=4):
    """Keys of `S3Token2Wav()` minus tokenizer.* / speaker_encoder.* (reference s3gen.py:53-258)."""
    sd = {}
    sd["flow.input_embedding.weight"] = _normal("flow.input_embedding.weight", (6561, 512), 1.0, seed)
    _linear(sd, "flow.spk_embed_affine_layer", 80, 192, seed)
    e = "flow.encoder."
    _linear(sd, e + "embed.out.0", 512, 512, seed)
    _norm(sd, e + "embed.out.1", 512, seed)
    _norm(sd, e + "after_norm", 512, seed)
    _conv(sd, e + "pre_lookahead_layer.conv1", 512, 512, 4, seed)
    _conv(sd, e + "pre_lookahead_layer.conv2", 512, 512, 3, seed)
    for i in range(n_enc):
        _conformer_layer(sd, e + f"encoders.{i}.", seed)
    _conv(sd, e + "up_layer.conv", 512, 512, 5, seed)
    _linear(sd, e + "up_embed.out.0", 512, 512, seed)
    _norm(sd, e + "up_embed.out.1", 512, seed)
    for i in range(n_up_enc):
        _conformer_layer(sd, e + f"up_encoders.{i}.", seed)
    _linear(sd, "flow.encoder_proj", 80, 512, seed)
    d = "flow.decoder.estimator."
    _linear(sd, d + "time_mlp.linear_1", 1024, 320, seed)
    _linear(sd, d + "time_mlp.linear_2", 1024, 1024, seed)
    if meanflow:
        w = torch.zeros(1024, 2048)
        w[:, :1024] = torch.eye(1024)
        sd[d + "time_embed_mixer.weight"] = w + _normal(d + "time_embed_mixer.weight", (1024, 2048), 0.01, seed)
    _cfm_stage(sd, d + "down_blocks.0.", 320, seed, tail=True)
    for i in range(n_mid):
        _cfm_stage(sd, d + f"mid_blocks.{i}.", 256, seed, tail=False)
    _cfm_stage(sd, d + "up_blocks.0.", 512, seed, tail=True)
    _conv(sd, d + "final_block.block.0", 256, 256, 3, seed)
    _norm(sd, d + "final_block.block.2", 256, seed)
    _conv(sd, d + "final_proj", 80, 256, 1, seed)

    h = "mel2wav."
    sd[h + "m_source.l_linear.weight"] = _uniform(h + "m_source.l_linear.weight", (1, 9), 1.5, seed)
    sd[h + "m_source.l_linear.bias"] = _uniform(h + "m_source.l_linear.bias", (1,), 0.1, seed)
    _wn_conv(sd, h + "conv_pre", 512, 80, 7, seed)
    sd[h + "conv_pre.bias"] = _uniform(h + "conv_pre.bias", (512,), 0.05, seed)
    for i, (cin, cout, k) in enumerate(((512, 256, 16), (256, 128, 11), (128, 64, 7))):
        # ConvTranspose1d weight is (Cin, Cout, k); weight_norm(dim=0) -> g per *input* channel
        _wn_conv(sd, h + f"ups.{i}", cin, cout, k, seed, fan_in=cin * k // (8, 5, 3)[i])
        sd[h + f"ups.{i}.bias"] = _uniform(h + f"ups.{i}.bias", (cout,), 0.05, seed)
    for i, (cout, k) in enumerate(((256, 30), (128, 6), (64, 1))):
        _conv(sd, h + f"source_downs.{i}", cout, 18, k, seed)
    for i, (ch, k) in enumerate(((256, 7), (128, 7), (64, 11))):
        _hift_resblock(sd, h + f"source_resblocks.{i}.", ch, k, seed)
    for i in range(9):
        _hift_resblock(sd, h + f"resblocks.{i}.", (256, 128, 64)[i // 3], (3, 7, 11)[i % 3], seed)
    # small conv_post so exp(.) stays far from the 1e2 clip and the +-0.99 clamp rarely fires
    _wn_conv(sd, h + "conv_post", 18, 64, 7, seed, gain=0.15)
    sd[h + "conv_post.bias"] = _uniform(h + "conv_post.bias", (18,), 0.05, seed) - 2.5
    f = h + "f0_predictor."
    for j, cin in zip((0, 2, 4, 6, 8), (80, 512, 512, 512, 512)):
        _wn_conv(sd, f + f"condnet.{j}", 512, cin, 3, seed)
        sd[f + f"condnet.{j}.bias"] = _uniform(f + f"condnet.{j}.bias", (512,), 0.05, seed)
    # scaled so that |f0| straddles the 10 Hz voiced threshold (both SineGen branches exercised)
    sd[f + "classifier.weight"] = _uniform(f + "classifier.weight", (1, 512), 12.0, seed)
    sd[f + "classifier.bias"] = torch.tensor([60.0])
    return sd


# ----------------------------------------------------------------------------- prompt-analysis networks


def _bnorm(sd, name, c, seed, affine=True):
    """eval-mode BatchNorm buffers (+ affine parameters)."""
    if affine:
        sd[name + ".weight"] = 1.0 + _normal(name + ".weight", (c,), 0.1, seed)
        sd[name + ".bias"] = _normal(name + ".bias", (c,), 0.1, seed)
    sd[name + ".running_mean"] = _normal(name + ".running_mean", (c,), 0.1, seed)
    sd[name + ".running_var"] = 1.0 + _normal(name + ".running_var", (c,), 0.1, seed).abs()
    sd[name + ".num_batches_tracked"] = torch.tensor(0)


CAMPPLUS_BLOCKS = ((12, 3, 1), (24, 3, 2), (16, 3, 2))  # (layers, kernel, dilation): reference xvector.py:377-379


def campplus_state_dict(seed=0, prefix="speaker_encoder."):
    """Keys of `CAMPPlus()` (reference models/s3gen/xvector.py:340-415), stored under `speaker_encoder.` in the S3Gen checkpoint."""
    sd = {}

    def conv2(name, co, ci, k):
        sd[name + ".weight"] = _normal(name + ".weight", (co, ci, k, k), math.sqrt(2.0 / (ci * k * k)), seed)

    def conv1(name, co, ci, k, bias=False):
        sd[name + ".weight"] = _normal(name + ".weight", (co, ci, k), math.sqrt(2.0 / (ci * k)), seed)
        if bias:
            sd[name + ".bias"] = _uniform(name + ".bias", (co,), 0.1, seed)

    conv2("head.conv1", 32, 1, 3)
    _bnorm(sd, "head.bn1", 32, seed)
    for layer in ("head.layer1", "head.layer2"):
        for j in (0, 1):
            q = f"{layer}.{j}"
            conv2(q + ".conv1", 32, 32, 3)
            _bnorm(sd, q + ".bn1", 32, seed)
            conv2(q + ".conv2", 32, 32, 3)
            _bnorm(sd, q + ".bn2", 32, seed)
            if j == 0:  # stride 2 -> projection shortcut
                conv2(q + ".shortcut.0", 32, 32, 1)
                _bnorm(sd, q + ".shortcut.1", 32, seed)
    conv2("head.conv2", 32, 32, 3)
    _bnorm(sd, "head.bn2", 32, seed)
    conv1("xvector.tdnn.linear", 128, 320, 5)
    _bnorm(sd, "xvector.tdnn.nonlinear.batchnorm", 128, seed)
    ch = 128
    for bi, (n_layers, k, _dil) in enumerate(CAMPPLUS_BLOCKS):
        for li in range(n_layers):
            q = f"xvector.block{bi + 1}.tdnnd{li + 1}"
            cin = ch + li * 32
            _bnorm(sd, q + ".nonlinear1.batchnorm", cin, seed)
            conv1(q + ".linear1", 128, cin, 1)
            _bnorm(sd, q + ".nonlinear2.batchnorm", 128, seed)
            conv1(q + ".cam_layer.linear_local", 32, 128, k)
            conv1(q + ".cam_layer.linear1", 64, 128, 1, bias=True)
            conv1(q + ".cam_layer.linear2", 32, 64, 1, bias=True)
        ch += n_layers * 32
        _bnorm(sd, f"xvector.transit{bi + 1}.nonlinear.batchnorm", ch, seed)
        conv1(f"xvector.transit{bi + 1}.linear", ch // 2, ch, 1)
        ch //= 2
    _bnorm(sd, "xvector.out_nonlinear.batchnorm", ch, seed)
    conv1("xvector.dense.linear", 192, 2 * ch, 1)
    _bnorm(sd, "xvector.dense.nonlinear.batchnorm", 192, seed, affine=False)
    return {prefix + k: v for k, v in sd.items()}


def voice_encoder_state_dict(seed=0):
    """Keys of `VoiceEncoder()` (reference models/voice_encoder/voice_encoder.py:139-154; ve.safetensors)."""
    sd = {"similarity_weight": torch.tensor([10.0]), "similarity_bias": torch.tensor([-5.0])}
    for l, cin in enumerate((40, 256, 256)):
        b = 1.0 / math.sqrt(256)
        # inputs are power mels (large dynamic range): keep layer 0's input weights small so the gates do not saturate
        sd[f"lstm.weight_ih_l{l}"] = _uniform(f"lstm.weight_ih_l{l}", (1024, cin), b * (0.05 if l == 0 else 1.0), seed)
        sd[f"lstm.weight_hh_l{l}"] = _uniform(f"lstm.weight_hh_l{l}", (1024, 256), b, seed)
        sd[f"lstm.bias_ih_l{l}"] = _uniform(f"lstm.bias_ih_l{l}", (1024,), b, seed)
        sd[f"lstm.bias_hh_l{l}"] = _uniform(f"lstm.bias_hh_l{l}", (1024,), b, seed)
    _linear(sd, "proj", 256, 256, seed)
    return sd


def s3tokenizer_state_dict(seed=0, prefix="tokenizer.", n_layer=6):
    """Keys of the third-party `S3TokenizerV2` as stored under `tokenizer.` in the S3Gen checkpoint (assumed from the published
    s3tokenizer package: AudioEncoderV2 + FSQ; parity unpinned)."""
    sd = {}
    D = 1280
    _conv(sd, "encoder.conv1", D, 128, 3, seed)
    _conv(sd, "encoder.conv2", D, D, 3, seed)
    for i in range(n_layer):
        q = f"encoder.blocks.{i}."
        _norm(sd, q + "attn_ln", D, seed)
        _linear(sd, q + "attn.query", D, D, seed)
        _linear(sd, q + "attn.key", D, D, seed, bias=False)
        _linear(sd, q + "attn.value", D, D, seed)
        _linear(sd, q + "attn.out", D, D, seed)
        sd[q + "attn.fsmn_block.weight"] = _uniform(q + "attn.fsmn_block.weight", (D, 1, 31), 0.1, seed)
        _norm(sd, q + "mlp_ln", D, seed)
        _linear(sd, q + "mlp.0", 4 * D, D, seed)
        _linear(sd, q + "mlp.2", D, 4 * D, seed)
    _linear(sd, "quantizer._codebook.project_down", 8, D, seed, gain=3.0)
    return {prefix + k: v for k, v in sd.items()}


def prompt_wav(seconds=6.0, sr=24000, seed=9):
    """A deterministic speech-like test signal: a few drifting harmonics with an amplitude envelope + a little noise, |x| < 1."""
    n = int(seconds * sr)
    t = torch.arange(n, dtype=torch.float64) / sr
    g = torch.Generator().manual_seed(9000 + seed)
    f0 = 120.0 + 40.0 * torch.sin(2 * math.pi * 0.7 * t) + 15.0 * torch.sin(2 * math.pi * 2.3 * t)
    ph = 2 * math.pi * torch.cumsum(f0, 0) / sr
    x = sum((0.5 / h) * torch.sin(h * ph + float(torch.rand(1, generator=g)) * 6.28) for h in range(1, 9))
    env = 0.5 * (1 + torch.sin(2 * math.pi * 1.1 * t - 1.0)).clamp(min=0.05)
    x = x * env + 0.01 * torch.randn(n, generator=g, dtype=torch.float64)
    return (0.6 * x / x.abs().max()).float()


# ----------------------------------------------------------------------------- synthetic inputs


def text_tokens(n=64, seed=1, vocab=2454):
    """`n` ids uniform in [1, vocab) excluding SOT 255 / EOT 0, then SOT/EOT padded (mtl_tts.py:319-322)."""
    g = torch.Generator().manual_seed(1000 + seed)
    t = torch.randint(1, vocab - 1, (n,), generator=g)
    t = torch.where(t >= 255, t + 1, t)
    return torch.cat([torch.tensor([255]), t, torch.tensor([0])]).long()


def t3_cond(seed=2, prompt_len=150, emotion=0.5):
    g = torch.Generator().manual_seed(2000 + seed)
    spk = torch.randn(1, 256, generator=g)
    spk = spk / spk.norm()
    toks = torch.randint(0, 6561, (1, prompt_len), generator=g)
    return dict(speaker_emb=spk, cond_prompt_speech_tokens=toks, emotion_adv=emotion * torch.ones(1, 1, 1))


def s3gen_ref(seed=4, n_prompt_tokens=250):
    g = torch.Generator().manual_seed(4000 + seed)
    tok = torch.randint(0, 6561, (1, n_prompt_tokens), generator=g)
    feat = (torch.randn(1, 2 * n_prompt_tokens, 80, generator=g) * 2 - 5).clamp(-11.5, 2.0)
    emb = torch.randn(1, 192, generator=g)
    return dict(prompt_token=tok, prompt_token_len=torch.tensor([n_prompt_tokens]), prompt_feat=feat,
                prompt_feat_len=None, embedding=emb)


def speech_tokens(n=250, seed=1):
    g = torch.Generator().manual_seed(5000 + seed)
    return torch.randint(0, 6561, (n,), generator=g)


def randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(6000 + seed))


def rand(shape, seed):
    return torch.rand(shape, generator=torch.Generator().manual_seed(7000 + seed))
