"""TEST INFRASTRUCTURE ONLY (the oracle) -- never imported by the product path.

CPU fp32 restatement of the reference's hot path generate() = T3.inference ->
S3Gen.flow_inference -> HiFT.inference, written functionally over the
reference's own state-dict keys.  Each function cites the reference lines it
follows (paths relative to /root/reference/src/chatterbox).  The third-party
arithmetic the reference delegates to (HF transformers Llama / logits
processors, diffusers Attention/GELU) is restated from the published
algorithms.

Pinning: the reference ships no tests or golden vectors, so this
restatement is pinned against outputs of the *reference itself* executed in the
authoring container (tests/golden/make_golden.py imports the unmodified
reference modules through oracle/ref_import.py, loads the same synthetic
state-dict and injected noise, and commits small fixtures under tests/golden/).
tests/test_oracle_golden.py re-checks this file against those fixtures anywhere.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
"""
import math

import torch
import torch.nn.functional as F

# =============================================================================
# T3 (Llama-520M backbone)         models/t3/t3.py, llama_configs.py:1-33
# =============================================================================

START_SPEECH, STOP_SPEECH = 6561, 6562


def llama3_inv_freq(head_dim=64, theta=500000.0, factor=8.0, low=1.0, high=4.0, orig=8192):
    """HF ROPE_INIT_FUNCTIONS['llama3'] with the reference's rope_scaling (llama_configs.py:22-30)."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    low_wl, high_wl = orig / low, orig / high
    wl = 2 * math.pi / inv
    scaled = torch.where(wl > low_wl, inv / factor, inv)
    smooth = (orig / wl - low) / (high - low)
    mid = (1 - smooth) * scaled / factor + smooth * scaled
    is_mid = ~(wl < high_wl) & ~(wl > low_wl)
    return torch.where(is_mid, mid, scaled)


def rope_cos_sin(positions, inv_freq):
    fr = positions.float()[:, None] * inv_freq[None, :]
    emb = torch.cat([fr, fr], dim=-1)
    return emb.cos(), emb.sin()


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def rms_norm(x, w, eps=1e-5):
    v = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(v + eps))


def llama_forward(sd, x, n_layers, past=None, n_heads=16):
    """HF LlamaModel forward on inputs_embeds (third-party; invoked from t3_hf_backend.py:93-100).

    x: (B, S, 1024).  past: list of (k, v) each (B, H, ctx, 64) or None.  Returns final-norm hidden, new past.
    Decode steps (S == 1 with a cache) attend to the whole cache unmasked."""
    B, S, D = x.shape
    hd = D // n_heads
    ctx0 = 0 if past is None else past[0][0].shape[2]
    pos = torch.arange(ctx0, ctx0 + S)
    cos, sin = rope_cos_sin(pos, llama3_inv_freq(hd))
    new_past = []
    for i in range(n_layers):
        p = f"tfmr.layers.{i}."
        h = rms_norm(x, sd[p + "input_layernorm.weight"])
        q = F.linear(h, sd[p + "self_attn.q_proj.weight"]).view(B, S, n_heads, hd).transpose(1, 2)
        k = F.linear(h, sd[p + "self_attn.k_proj.weight"]).view(B, S, n_heads, hd).transpose(1, 2)
        v = F.linear(h, sd[p + "self_attn.v_proj.weight"]).view(B, S, n_heads, hd).transpose(1, 2)
        q = q * cos + _rot_half(q) * sin
        k = k * cos + _rot_half(k) * sin
        if past is not None:
            k = torch.cat([past[i][0], k], dim=2)
            v = torch.cat([past[i][1], v], dim=2)
        new_past.append((k, v))
        a = F.scaled_dot_product_attention(q, k, v, is_causal=(S > 1 and past is None))
        a = a.transpose(1, 2).reshape(B, S, D)
        x = x + F.linear(a, sd[p + "self_attn.o_proj.weight"])
        h = rms_norm(x, sd[p + "post_attention_layernorm.weight"])
        g = F.silu(F.linear(h, sd[p + "mlp.gate_proj.weight"])) * F.linear(h, sd[p + "mlp.up_proj.weight"])
        x = x + F.linear(g, sd[p + "mlp.down_proj.weight"])
    return rms_norm(x, sd["tfmr.norm.weight"]), new_past


def perceiver(sd, h):
    """Perceiver.forward + AttentionBlock2.forward (t3/modules/perceiver.py:156-170,200-212)."""
    p = "cond_enc.perceiver."

    def attn_block(x1, x2):
        n1 = F.layer_norm(x1, (1024,), sd[p + "attn.norm.weight"], sd[p + "attn.norm.bias"])
        n2 = F.layer_norm(x2, (1024,), sd[p + "attn.norm.weight"], sd[p + "attn.norm.bias"])
        q = F.linear(n1, sd[p + "attn.to_q.weight"], sd[p + "attn.to_q.bias"])
        k = F.linear(n2, sd[p + "attn.to_k.weight"], sd[p + "attn.to_k.bias"])
        v = F.linear(n2, sd[p + "attn.to_v.weight"], sd[p + "attn.to_v.bias"])
        B = q.shape[0]
        sp = lambda t: t.view(B, t.shape[1], 4, 256).permute(0, 2, 1, 3)
        o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v))
        o = o.permute(0, 2, 1, 3).reshape(B, -1, 1024)
        return x1 + F.linear(o, sd[p + "attn.proj_out.weight"], sd[p + "attn.proj_out.bias"])

    q0 = sd[p + "pre_attention_query"].expand(h.shape[0], -1, -1)
    pre = attn_block(q0, h)
    return attn_block(pre, pre)


def t3_cond_embeds(sd, speaker_emb, prompt_tokens, emotion_adv):
    """T3.prepare_conditioning + T3CondEnc.forward (t3.py:92-100, cond_enc.py:64-97) -> (1, 34, 1024)."""
    spk = F.linear(speaker_emb.view(-1, 256), sd["cond_enc.spkr_enc.weight"], sd["cond_enc.spkr_enc.bias"])[:, None]
    pe = sd["speech_emb.weight"][prompt_tokens] + sd["speech_pos_emb.emb.weight"][: prompt_tokens.shape[1]][None]
    per = perceiver(sd, pe)
    emo = F.linear(emotion_adv.view(-1, 1, 1), sd["cond_enc.emotion_adv_fc.weight"])
    return torch.cat([spk, per, emo], dim=1)


def t3_prefill_embeds(sd, cond, text_tokens):
    """prepare_input_embeds + the second BOS (t3.py:102-130, 305-313).  text_tokens (2, Tt) incl. SOT/EOT;
    row 1 is the CFG-unconditional row (token embeddings zeroed, positional kept)."""
    te = sd["text_emb.weight"][text_tokens].clone()
    te[1].zero_()
    te = te + sd["text_pos_emb.emb.weight"][: text_tokens.shape[1]][None]
    bos = (sd["speech_emb.weight"][START_SPEECH] + sd["speech_pos_emb.emb.weight"][0])[None, None].expand(2, 1, -1)
    return torch.cat([cond.expand(2, -1, -1), te, bos, bos], dim=1)


def process_logits(cond, uncond, generated, cfg_weight=0.5, temperature=0.8, min_p=0.05, top_p=1.0,
                   repetition_penalty=1.2):
    """The per-step logit pipeline (t3.py:339-356) with HF processor semantics.
    cond/uncond: (V,), generated: 1-D LongTensor of ids so far (incl. BOS).  Returns filtered logits (V,)."""
    l = cond + cfg_weight * (cond - uncond)
    ids = torch.unique(generated)
    s = l[ids]
    l = l.clone()
    l[ids] = torch.where(s < 0, s * repetition_penalty, s / repetition_penalty)
    if temperature != 1.0:
        l = l / temperature
    pr = torch.softmax(l, -1)
    l = l.masked_fill(pr < min_p * pr.max(), float("-inf"))
    if top_p < 1.0:
        sl, si = torch.sort(l, descending=False)
        cp = sl.softmax(-1).cumsum(-1)
        rm = cp <= (1 - top_p)
        rm[-1] = False
        l = l.masked_fill(torch.zeros_like(rm).scatter(0, si, rm), float("-inf"))
    return l


def sample_inverse_cdf(probs, u):
    """Deterministic stand-in for torch.multinomial(probs, 1) (t3.py:360) given an injected uniform u in [0,1):
    the first index whose inclusive cumulative probability exceeds u * sum(probs)."""
    c = probs.double().cumsum(-1)
    idx = int(torch.searchsorted(c, torch.tensor(float(u) * float(c[-1]), dtype=torch.float64), right=True))
    nz = torch.nonzero(probs > 0).flatten()
    return min(idx, int(nz[-1]))


def t3_inference(sd, n_layers, cond_in, text_tokens, max_new_tokens, uniforms, cfg_weight=0.5, temperature=0.8,
                 min_p=0.05, top_p=1.0, repetition_penalty=1.2, ban_eos=False, forced_tokens=None,
                 return_logits=False):
    """T3.inference (t3.py:226-390) for ONE utterance (2 CFG rows).  `uniforms[i]` replaces the RNG of step i.
    forced_tokens: teacher forcing (sampled ids are replaced), used for logits parity."""
    cond = t3_cond_embeds(sd, cond_in["speaker_emb"], cond_in["cond_prompt_speech_tokens"], cond_in["emotion_adv"])
    emb = t3_prefill_embeds(sd, cond, text_tokens)
    hid, past = llama_forward(sd, emb, n_layers)
    logits = F.linear(hid[:, -1], sd["speech_head.weight"])
    generated = [START_SPEECH]
    out, all_logits = [], []
    for i in range(max_new_tokens):
        if return_logits:
            all_logits.append(logits.clone())
        c, u = logits[0], logits[1]
        l = process_logits(c, u, torch.tensor(generated), cfg_weight, temperature, min_p, top_p, repetition_penalty)
        pr = torch.softmax(l, -1)
        if ban_eos:  # fixed-length synthetic runs: EOS probability zeroed, remaining mass renormalised
            pr[STOP_SPEECH] = 0.0
        tok = sample_inverse_cdf(pr, uniforms[i])
        if forced_tokens is not None:
            tok = int(forced_tokens[i])
        out.append(tok)
        generated.append(tok)
        if tok == STOP_SPEECH:
            break
        e = (sd["speech_emb.weight"][tok] + sd["speech_pos_emb.emb.weight"][i + 1])[None, None].expand(2, 1, -1)
        hid, past = llama_forward(sd, e, n_layers, past)
        logits = F.linear(hid[:, -1], sd["speech_head.weight"])
    res = torch.tensor(out, dtype=torch.long)
    return (res, torch.stack(all_logits)) if return_logits else res


# =============================================================================
# S3Gen: conformer encoder          s3gen/transformer/upsample_encoder.py
# =============================================================================


def _rel_pos_table(T, d=512):
    """EspnetRelPositionalEncoding.position_encoding (embedding.py:224-294): rows r=0..2T-2 <-> rel pos T-1-r."""
    pos = torch.arange(T - 1, -T, -1, dtype=torch.float32)[:, None]
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(2 * T - 1, d)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def _conformer_layer(sd, p, x, pos_emb, key_mask):
    """ConformerEncoderLayer.forward + RelPositionMultiHeadedAttention.forward
    (encoder_layer.py:160-236, attention.py:249-330); no macaron, no conv module."""
    B, T, _ = x.shape
    h = F.layer_norm(x, (512,), sd[p + "norm_mha.weight"], sd[p + "norm_mha.bias"], 1e-12)
    a = p + "self_attn."
    q = F.linear(h, sd[a + "linear_q.weight"], sd[a + "linear_q.bias"]).view(B, T, 8, 64)
    k = F.linear(h, sd[a + "linear_k.weight"], sd[a + "linear_k.bias"]).view(B, T, 8, 64).transpose(1, 2)
    v = F.linear(h, sd[a + "linear_v.weight"], sd[a + "linear_v.bias"]).view(B, T, 8, 64).transpose(1, 2)
    pp = F.linear(pos_emb, sd[a + "linear_pos.weight"]).view(1, -1, 8, 64).transpose(1, 2)  # (1,8,2T-1,64)
    qu = (q + sd[a + "pos_bias_u"]).transpose(1, 2)
    qv = (q + sd[a + "pos_bias_v"]).transpose(1, 2)
    ac = qu @ k.transpose(-2, -1)
    bd = qv @ pp.transpose(-2, -1)  # (B,8,T,2T-1)
    idx = (T - 1 - torch.arange(T)[:, None] + torch.arange(T)[None, :])  # rel_shift: bd[i, T-1-i+j]
    bd = torch.gather(bd, 3, idx[None, None].expand(B, 8, T, T))
    sc = (ac + bd) / 8.0
    m = ~key_mask[:, None, None, :]
    at = torch.softmax(sc.masked_fill(m, float("-inf")), -1).masked_fill(m, 0.0)
    o = (at @ v).transpose(1, 2).reshape(B, T, 512)
    x = x + F.linear(o, sd[a + "linear_out.weight"], sd[a + "linear_out.bias"])
    h = F.layer_norm(x, (512,), sd[p + "norm_ff.weight"], sd[p + "norm_ff.bias"], 1e-12)
    f = p + "feed_forward."
    h = F.linear(F.silu(F.linear(h, sd[f + "w_1.weight"], sd[f + "w_1.bias"])), sd[f + "w_2.weight"], sd[f + "w_2.bias"])
    return x + h


def _count(sd, prefix):
    n = 0
    while f"{prefix}{n}.norm_ff.weight" in sd:
        n += 1
    return n


def encoder_forward(sd, xs, lens):
    """UpsampleConformerEncoder.forward (upsample_encoder.py:237-304).  xs (B,T,512) already masked, lens (B,)."""
    e = "flow.encoder."
    B, T, _ = xs.shape
    km = torch.arange(T)[None, :] < lens[:, None]

    def embed(pfx, x):
        x = F.layer_norm(F.linear(x, sd[pfx + "out.0.weight"], sd[pfx + "out.0.bias"]), (512,),
                         sd[pfx + "out.1.weight"], sd[pfx + "out.1.bias"], 1e-5)
        return x * math.sqrt(512.0), _rel_pos_table(x.shape[1])

    x, pe = embed(e + "embed.", xs)
    # PreLookaheadLayer (upsample_encoder.py:81-96)
    y = x.transpose(1, 2)
    y = F.leaky_relu(F.conv1d(F.pad(y, (0, 3)), sd[e + "pre_lookahead_layer.conv1.weight"], sd[e + "pre_lookahead_layer.conv1.bias"]))
    y = F.conv1d(F.pad(y, (2, 0)), sd[e + "pre_lookahead_layer.conv2.weight"], sd[e + "pre_lookahead_layer.conv2.bias"])
    x = y.transpose(1, 2) + x
    for i in range(_count(sd, e + "encoders.")):
        x = _conformer_layer(sd, e + f"encoders.{i}.", x, pe, km)
    # Upsample1D (upsample_encoder.py:59-63): nearest x2, left pad 4, conv k5
    y = x.transpose(1, 2).repeat_interleave(2, dim=2)
    y = F.conv1d(F.pad(y, (4, 0)), sd[e + "up_layer.conv.weight"], sd[e + "up_layer.conv.bias"])
    x = y.transpose(1, 2)
    km2 = torch.arange(2 * T)[None, :] < (2 * lens)[:, None]
    x, pe = embed(e + "up_embed.", x)
    for i in range(_count(sd, e + "up_encoders.")):
        x = _conformer_layer(sd, e + f"up_encoders.{i}.", x, pe, km2)
    return F.layer_norm(x, (512,), sd[e + "after_norm.weight"], sd[e + "after_norm.bias"], 1e-5), km2


# =============================================================================
# S3Gen: CFM estimator + Euler      s3gen/decoder.py, flow_matching.py, matcha/*
# =============================================================================


def _sinusoidal(t, dim=320, scale=1000.0):
    half = dim // 2
    f = torch.exp(torch.arange(half).float() * -(math.log(10000) / (half - 1)))
    e = scale * t[:, None] * f[None]
    return torch.cat([e.sin(), e.cos()], -1)


def _causal_block(sd, p, x, mask):
    """CausalBlock1D (decoder.py:49-63): (x*m) -> causal conv k3 -> LN(C) -> Mish -> *m.  x (B,C,T)."""
    y = F.conv1d(F.pad(x * mask, (2, 0)), sd[p + "block.0.weight"], sd[p + "block.0.bias"])
    y = F.layer_norm(y.transpose(1, 2), (y.shape[1],), sd[p + "block.2.weight"], sd[p + "block.2.bias"]).transpose(1, 2)
    return F.mish(y) * mask


def _resnet(sd, p, x, mask, temb):
    """ResnetBlock1D.forward with causal blocks (matcha/decoder.py:56-61, decoder.py:66-70)."""
    h = _causal_block(sd, p + "block1.", x, mask)
    h = h + F.linear(F.mish(temb), sd[p + "mlp.1.weight"], sd[p + "mlp.1.bias"])[:, :, None]
    h = _causal_block(sd, p + "block2.", h, mask)
    return h + F.conv1d(x * mask, sd[p + "res_conv.weight"], sd[p + "res_conv.bias"])


def _tblock(sd, p, x, bias):
    """BasicTransformerBlock.forward (matcha/transformer.py:243-316) with diffusers Attention/GELU semantics."""
    B, T, _ = x.shape
    h = F.layer_norm(x, (256,), sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    sp = lambda t: t.view(B, T, 8, 64).transpose(1, 2)
    q, k, v = (sp(F.linear(h, sd[p + f"attn1.to_{c}.weight"])) for c in "qkv")
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=bias[:, None])  # bias (B,1,T) -> (B,1,1,T)
    o = o.transpose(1, 2).reshape(B, T, 512)
    x = x + F.linear(o, sd[p + "attn1.to_out.0.weight"], sd[p + "attn1.to_out.0.bias"])
    h = F.layer_norm(x, (256,), sd[p + "norm3.weight"], sd[p + "norm3.bias"])
    h = F.gelu(F.linear(h, sd[p + "ff.net.0.proj.weight"], sd[p + "ff.net.0.proj.bias"]))
    return x + F.linear(h, sd[p + "ff.net.2.weight"], sd[p + "ff.net.2.bias"])


def estimator_forward(sd, x, mask, mu, t, spks, cond, r=None):
    """ConditionalDecoder.forward (decoder.py:243-333).  x, mu, cond (B,80,T); mask (B,1,T); t (B,); spks (B,80)."""
    d = "flow.decoder.estimator."

    def time_mlp(tt):
        e = _sinusoidal(tt)
        e = F.silu(F.linear(e, sd[d + "time_mlp.linear_1.weight"], sd[d + "time_mlp.linear_1.bias"]))
        return F.linear(e, sd[d + "time_mlp.linear_2.weight"], sd[d + "time_mlp.linear_2.bias"])

    temb = time_mlp(t)
    if r is not None:
        temb = F.linear(torch.cat([temb, time_mlp(r)], 1), sd[d + "time_embed_mixer.weight"])
    T = x.shape[2]
    h = torch.cat([x, mu, spks[:, :, None].expand(-1, -1, T), cond], 1)
    bias = (1.0 - mask) * -1.0e10  # mask_to_bias (decoder.py:26-34)

    def stage(p, h):
        h = _resnet(sd, p + "0.", h, mask, temb).transpose(1, 2)
        for j in range(4):
            h = _tblock(sd, p + f"1.{j}.", h, bias)
        return h.transpose(1, 2)

    h = stage(d + "down_blocks.0.", h)
    skip = h
    h = F.conv1d(F.pad(h * mask, (2, 0)), sd[d + "down_blocks.0.2.weight"], sd[d + "down_blocks.0.2.bias"])
    n_mid = 0
    while f"{d}mid_blocks.{n_mid}.0.mlp.1.weight" in sd:
        h = stage(d + f"mid_blocks.{n_mid}.", h)
        n_mid += 1
    h = stage(d + "up_blocks.0.", torch.cat([h, skip], 1))
    h = F.conv1d(F.pad(h * mask, (2, 0)), sd[d + "up_blocks.0.2.weight"], sd[d + "up_blocks.0.2.bias"])
    h = _causal_block(sd, d + "final_block.", h, mask)
    return F.conv1d(h * mask, sd[d + "final_proj.weight"], sd[d + "final_proj.bias"]) * mask


def cfm_solve(sd, z, mu, mask, spks, cond, n_timesteps=10, cfg_rate=0.7, meanflow=False):
    """CausalConditionalCFM.forward + solve_euler / basic_euler (flow_matching.py:78-145,196-246).  z injected."""
    t_span = torch.linspace(0, 1, n_timesteps + 1)
    if not meanflow:
        t_span = 1 - torch.cos(t_span * 0.5 * math.pi)
    x = z
    B = mu.shape[0]
    for t, r in zip(t_span[:-1], t_span[1:]):
        if meanflow:
            dxdt = estimator_forward(sd, x, mask, mu, t[None].expand(B), spks, cond, r=r[None].expand(B))
        else:
            z0 = torch.zeros_like(mu)
            out = estimator_forward(sd, torch.cat([x, x]), torch.cat([mask, mask]), torch.cat([mu, z0]),
                                    t[None].expand(2 * B), torch.cat([spks, torch.zeros_like(spks)]),
                                    torch.cat([cond, z0]))
            dxdt = (1.0 + cfg_rate) * out[:B] - cfg_rate * out[B:]
        x = x + (r - t) * dxdt
    return x


def flow_inference(sd, tokens, token_lens, ref, z, n_timesteps=10, meanflow=False, hold_back=None):
    """CausalMaskedDiffWithXvec.inference (flow.py:131-198), finalize=True.  tokens (B,N) padded, token_lens (B,);
    ref: dict prompt_token (1,P), prompt_feat (1,2P,80), embedding (1,192); z (B,80,2P+2N) injected noise.
    Returns mel (B,80,2N) (padded region beyond 2*token_lens[b] is not meaningful).
    hold_back (B,) int or None: chunked synthesis -- the last hold_back[b] mel frames of utterance b (the encoder's 3-token lookahead,
    flow.py:170-171 `finalize=False`; that branch of the reference raises a shape error, so the semantics are restated here: the
    frames are masked out of the CFM exactly like padding) are not generated."""
    B = tokens.shape[0]
    emb = F.normalize(ref["embedding"].float().view(1, -1), dim=1)
    spk = F.linear(emb, sd["flow.spk_embed_affine_layer.weight"], sd["flow.spk_embed_affine_layer.bias"]).expand(B, -1)
    P = ref["prompt_token"].shape[1]
    tok = torch.cat([ref["prompt_token"].expand(B, -1), tokens], 1)
    lens = P + token_lens
    m = (torch.arange(tok.shape[1])[None] < lens[:, None]).float()[:, :, None]
    x = sd["flow.input_embedding.weight"][tok.long()] * m
    h, hm = encoder_forward(sd, x, lens)
    mu = F.linear(h, sd["flow.encoder_proj.weight"], sd["flow.encoder_proj.bias"]).transpose(1, 2)
    T = mu.shape[2]
    cond = torch.zeros(B, 80, T)
    cond[:, :, : 2 * P] = ref["prompt_feat"].float().transpose(1, 2)
    mask = hm.float()[:, None, :]
    if hold_back is not None:
        keep = 2 * lens - torch.as_tensor(hold_back)
        mask = mask * (torch.arange(T)[None] < keep[:, None]).float()[:, None, :]
    mel = cfm_solve(sd, z, mu, mask, spk, cond, n_timesteps, meanflow=meanflow)
    return mel[:, :, 2 * P:]


# =============================================================================
# HiFT vocoder                      s3gen/hifigan.py, f0_predictor.py
# =============================================================================


def _wn(sd, p):
    """Fold the weight_norm parametrization: w = g * v / ||v|| (norm over all dims but 0)."""
    if p + ".weight" in sd:
        return sd[p + ".weight"]
    g, v = sd[p + ".parametrizations.weight.original0"], sd[p + ".parametrizations.weight.original1"]
    return g * v / v.flatten(1).norm(dim=1).view(-1, 1, 1)


def f0_predict(sd, mel):
    """ConvRNNF0Predictor.forward (f0_predictor.py:52-55).  mel (B,80,T) -> f0 (B,T)."""
    p = "mel2wav.f0_predictor."
    x = mel
    for j in (0, 2, 4, 6, 8):
        x = F.elu(F.conv1d(x, _wn(sd, p + f"condnet.{j}"), sd[p + f"condnet.{j}.bias"], padding=1))
    return torch.abs(F.linear(x.transpose(1, 2), sd[p + "classifier.weight"], sd[p + "classifier.bias"]).squeeze(-1))


def source_module(sd, f0, phase, noise, sr=24000, up=480):
    """f0_upsamp + SourceModuleHnNSF/SineGen (hifigan.py:201-231,267-283,467-469).
    f0 (B,T); phase (B,9,1) U(-pi,pi) with [:,0]=0 injected; noise (B,9,480T) N(0,1) injected.  -> s (B,1,480T)"""
    f0u = f0[:, None, :].repeat_interleave(up, dim=2)  # nearest upsample
    mult = torch.arange(1, 10, dtype=torch.float32)[None, :, None]
    F_mat = f0u * mult / sr
    theta = 2 * math.pi * (torch.cumsum(F_mat, dim=-1) % 1)  # CPU cumsum accumulates in double, rounds to fp32
    sine = 0.1 * torch.sin(theta + phase)
    uv = (f0u > 10).float()
    sine = sine * uv + (uv * 0.003 + (1 - uv) * 0.1 / 3) * noise
    merged = torch.tanh(F.linear(sine.transpose(1, 2), sd["mel2wav.m_source.l_linear.weight"], sd["mel2wav.m_source.l_linear.bias"]))
    return merged.transpose(1, 2)


def _snake(x, alpha):
    a = alpha[None, :, None]
    return x + (1.0 / (a + 1e-9)) * torch.sin(x * a) ** 2


def _resblock(sd, p, x, k):
    for j, dil in enumerate((1, 3, 5)):
        xt = _snake(x, sd[p + f"activations1.{j}.alpha"])
        xt = F.conv1d(xt, _wn(sd, p + f"convs1.{j}"), sd[p + f"convs1.{j}.bias"], dilation=dil, padding=(k * dil - dil) // 2)
        xt = _snake(xt, sd[p + f"activations2.{j}.alpha"])
        xt = F.conv1d(xt, _wn(sd, p + f"convs2.{j}"), sd[p + f"convs2.{j}.bias"], padding=(k - 1) // 2)
        x = xt + x
    return x


def hift_decode(sd, mel, s):
    """HiFTGenerator.decode (hifigan.py:412-444).  mel (B,80,T), s (B,1,480T) -> wav (B,480T)."""
    h = "mel2wav."
    win = torch.hann_window(16, periodic=True)
    spec = torch.stft(s.squeeze(1), 16, 4, 16, window=win, return_complex=True)
    s_stft = torch.cat([spec.real, spec.imag], 1)
    x = F.conv1d(mel, _wn(sd, h + "conv_pre"), sd[h + "conv_pre.bias"], padding=3)
    for i, (u, k) in enumerate(((8, 16), (5, 11), (3, 7))):
        x = F.leaky_relu(x, 0.1)
        x = F.conv_transpose1d(x, _wn(sd, h + f"ups.{i}"), sd[h + f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        if i == 2:
            x = F.pad(x, (1, 0), mode="reflect")
        st, pd = ((15, 7), (3, 1), (1, 0))[i]
        si = F.conv1d(s_stft, sd[h + f"source_downs.{i}.weight"], sd[h + f"source_downs.{i}.bias"], stride=st, padding=pd)
        si = _resblock(sd, h + f"source_resblocks.{i}.", si, (7, 7, 11)[i])
        x = x + si
        xs = None
        for j, kk in enumerate((3, 7, 11)):
            r = _resblock(sd, h + f"resblocks.{i * 3 + j}.", x, kk)
            xs = r if xs is None else xs + r
        x = xs / 3
    x = F.leaky_relu(x)
    x = F.conv1d(x, _wn(sd, h + "conv_post"), sd[h + "conv_post.bias"], padding=3)
    mag = torch.exp(x[:, :9]).clip(max=1e2)
    ph = torch.sin(x[:, 9:])
    wav = torch.istft(torch.complex(mag * torch.cos(ph), mag * torch.sin(ph)), 16, 4, 16, window=win)
    return wav.clamp(-0.99, 0.99)


def hift_inference(sd, mel, phase, noise, cache_source=None):
    """HiFTGenerator.inference (hifigan.py:462-474) with injected SineGen phase/noise.  Returns (wav, source).
    cache_source (B,1,L): the source of an earlier chunk overrides the first L samples (hifigan.py:470-472)."""
    f0 = f0_predict(sd, mel)
    s = source_module(sd, f0, phase, noise)
    if cache_source is not None and cache_source.shape[2]:
        s = s.clone()
        s[:, :, : cache_source.shape[2]] = cache_source
    return hift_decode(sd, mel, s), s


def trim_fade(wav, sr=24000):
    """S3Token2Wav trim_fade (s3gen.py:255-258,360): first 20 ms silenced, next 20 ms raised-cosine fade-in."""
    n = sr // 50
    tf = torch.zeros(2 * n)
    tf[n:] = (torch.cos(torch.linspace(math.pi, 0, n)) + 1) / 2
    wav = wav.clone()
    wav[:, : 2 * n] *= tf
    return wav


# =============================================================================
# end-to-end + parity metrics
# =============================================================================


def s3gen_inference(sd, tokens, token_lens, ref, z, phase, noise, n_timesteps=10):
    """S3Token2Wav.inference (s3gen.py:330-362): flow -> HiFT -> trim_fade.  Returns (wav, mel)."""
    mel = flow_inference(sd, tokens, token_lens, ref, z, n_timesteps)
    wav, _ = hift_inference(sd, mel, phase, noise)
    return trim_fade(wav), mel


def slaney_mel_filter(sr=24000, n_fft=1920, n_mels=80, fmin=0.0, fmax=8000.0):
    """librosa.filters.mel (slaney) as used by s3gen/utils/mel.py:56."""
    def h2m(f):
        f = torch.as_tensor(f, dtype=torch.float64)
        lin = f / (200.0 / 3)
        return torch.where(f >= 1000.0, 15.0 + torch.log(f.clamp(min=1e-10) / 1000.0) / (math.log(6.4) / 27.0), lin)

    def m2h(m):
        lin = m * (200.0 / 3)
        return torch.where(m >= 15.0, 1000.0 * torch.exp((math.log(6.4) / 27.0) * (m - 15.0)), lin)

    ff = torch.linspace(0, sr / 2.0, 1 + n_fft // 2, dtype=torch.float64)
    mf = m2h(torch.linspace(float(h2m(fmin)), float(h2m(fmax)), n_mels + 2, dtype=torch.float64))
    fd = mf[1:] - mf[:-1]
    ramps = mf[:, None] - ff[None, :]
    w = torch.clamp(torch.minimum(-ramps[:-2] / fd[:-1, None], ramps[2:] / fd[1:, None]), min=0)
    return (w * (2.0 / (mf[2:] - mf[:-2]))[:, None]).float()


def log_mel(wav):
    """mel_spectrogram (s3gen/utils/mel.py:36-85) -- the mel-L1 parity metric.  wav (B,L) -> (B,80,frames)."""
    y = F.pad(wav[:, None], (720, 720), mode="reflect").squeeze(1)
    sp = torch.stft(y, 1920, hop_length=480, win_length=1920, window=torch.hann_window(1920), center=False,
                    return_complex=True)
    mag = torch.sqrt(sp.real ** 2 + sp.imag ** 2 + 1e-9)
    return torch.log(torch.clamp(slaney_mel_filter() @ mag, min=1e-5))


# =============================================================================
# Turbo / Nano: GPT-2 backbone T3 + meanflow S3Gen       models/t3/t3.py:392-468, tts_turbo.py:153-167
# =============================================================================


def gpt2_forward(sd, x, n_layers, n_heads, past=None):
    """HF GPT2Model on inputs_embeds (third-party; llama_configs.py:35-103): learned wpe added to EVERY input at its
    absolute position, pre-LN blocks, Conv1D weights stored [in, out], gelu_new, final ln_f."""
    B, S, D = x.shape
    hd = D // n_heads
    p0 = 0 if past is None else past[0][0].shape[2]
    x = x + sd["tfmr.wpe.weight"][p0:p0 + S][None]
    new_past = []
    for i in range(n_layers):
        p = f"tfmr.h.{i}."
        h = F.layer_norm(x, (D,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], 1e-5)
        qkv = h @ sd[p + "attn.c_attn.weight"] + sd[p + "attn.c_attn.bias"]
        q, k, v = (t.view(B, S, n_heads, hd).transpose(1, 2) for t in qkv.split(D, dim=2))
        if past is not None:
            k, v = torch.cat([past[i][0], k], 2), torch.cat([past[i][1], v], 2)
        new_past.append((k, v))
        a = F.scaled_dot_product_attention(q, k, v, is_causal=(S > 1 and past is None))
        a = a.transpose(1, 2).reshape(B, S, D)
        x = x + a @ sd[p + "attn.c_proj.weight"] + sd[p + "attn.c_proj.bias"]
        h = F.layer_norm(x, (D,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], 1e-5)
        g = F.gelu(h @ sd[p + "mlp.c_fc.weight"] + sd[p + "mlp.c_fc.bias"], approximate="tanh")
        x = x + g @ sd[p + "mlp.c_proj.weight"] + sd[p + "mlp.c_proj.bias"]
    return F.layer_norm(x, (D,), sd["tfmr.ln_f.weight"], sd["tfmr.ln_f.bias"], 1e-5), new_past


def process_logits_turbo(l, ids, temperature=0.8, top_k=1000, top_p=0.95, repetition_penalty=1.2):
    """LogitsProcessorList order of inference_turbo (t3.py:396-404): Temperature -> TopK -> TopP -> RepetitionPenalty."""
    if temperature > 0 and temperature != 1.0:
        l = l / temperature
    if top_k > 0:
        kth = torch.topk(l, min(top_k, l.numel()))[0][-1]
        l = l.masked_fill(l < kth, float("-inf"))
    if top_p < 1.0:
        sl, si = torch.sort(l, descending=False)
        rm = sl.softmax(-1).cumsum(-1) <= (1 - top_p)
        rm[-1] = False
        l = l.masked_fill(torch.zeros_like(rm).scatter(0, si, rm), float("-inf"))
    if repetition_penalty != 1.0:
        u = torch.unique(ids)
        s = l[u]
        l = l.clone()
        l[u] = torch.where(s < 0, s * repetition_penalty, s / repetition_penalty)
    return l


def t3_inference_turbo(sd, n_layers, n_heads, cond_in, text_tokens, max_gen_len, uniforms, temperature=0.8, top_k=1000,
                       top_p=0.95, repetition_penalty=1.2, ban_eos=False, return_logits=False):
    """T3.inference_turbo (t3.py:392-468) for one utterance: prefix [speaker | 375 prompt-token embeddings | text |
    start-speech], no CFG, no positional embedding of our own (GPT-2's wpe does it).  Returns ids without a trailing EOS."""
    stop = 6562
    spk = F.linear(cond_in["speaker_emb"].view(-1, 256), sd["cond_enc.spkr_enc.weight"], sd["cond_enc.spkr_enc.bias"])[:, None]
    prm = sd["speech_emb.weight"][cond_in["cond_prompt_speech_tokens"].view(1, -1)]
    emb = torch.cat([spk, prm, sd["text_emb.weight"][text_tokens.view(1, -1)], sd["speech_emb.weight"][START_SPEECH].view(1, 1, -1)], 1)
    hid, past = gpt2_forward(sd, emb, n_layers, n_heads)
    logits = F.linear(hid[:, -1], sd["speech_head.weight"], sd["speech_head.bias"])[0]
    out, all_logits = [], []
    ids = torch.tensor([START_SPEECH])
    for i in range(max_gen_len + 1):
        if return_logits:
            all_logits.append(logits.clone())
        l = process_logits_turbo(logits, ids, temperature, top_k, top_p, repetition_penalty)
        pr = torch.softmax(l, -1)
        if ban_eos:
            pr[stop] = 0.0
        tok = sample_inverse_cdf(pr, uniforms[i])
        out.append(tok)
        if tok == stop:
            break
        ids = torch.tensor(out)
        hid, past = gpt2_forward(sd, sd["speech_emb.weight"][tok].view(1, 1, -1), n_layers, n_heads, past)
        logits = F.linear(hid[:, -1], sd["speech_head.weight"], sd["speech_head.bias"])[0]
    if out and out[-1] == stop:
        out = out[:-1]
    res = torch.tensor(out, dtype=torch.long)
    return (res, torch.stack(all_logits)) if return_logits else res
