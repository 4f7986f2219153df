"""TEST INFRASTRUCTURE ONLY -- CPU (torch / numpy, fp32) restatement of the voice-prompt / voice-conversion front-end
(SURVEY.md 8a row a16, 8f N1/N2).  Never imported by the product path.

Pinned against the UNMODIFIED reference (tests/golden/make_golden_frontend.py imports the reference modules and stores golden
vectors; tests/test_oracle_golden.py checks this file against them):
  * s3_log_mel              reference models/s3tokenizer/s3tokenizer.py:128-168
  * mel_spectrogram_24k     reference models/s3gen/utils/mel.py:41-85
  * campplus_forward        reference models/s3gen/xvector.py:60-428 (network body; features in)
  * ve_inference            reference models/voice_encoder/voice_encoder.py:139-200 (partials + 3-layer LSTM + projection)
  * ve_num_wins / ve_frame_step  reference voice_encoder.py:55-81

PARITY UNPINNED (the arithmetic lives in third-party packages that are absent from /root/reference and from this container;
restated from their published behaviour, anchored only on the reference's call sites):
  * s3tokenizer_quantize    `s3tokenizer` package (pyproject.toml:17, unpinned version): S3TokenizerV2.quantize -- AudioEncoderV2
                            (2 strided convs, 6 FSMN-attention blocks with rotary embeddings) + FSQ codebook; SURVEY.md A.6.
                            Call sites: s3tokenizer.py:116-126, s3gen.py:147, tts.py:194, vc.py:97.
  * kaldi_fbank             torchaudio.compliance.kaldi.fbank(num_mel_bins=80) defaults (xvector.py:51)
  * ve_melspectrogram       librosa.stft(center=True, reflect) + librosa.filters.mel (voice_encoder/melspec.py:24-60); the
                            filterbank is the slaney restatement that utils/mel.py pins
  * resample / trim_silence librosa.load / librosa.resample / torchaudio Resample / librosa.effects.trim (tts.py:184-186,
                            s3gen.py:136-146, voice_encoder.py:262-270)
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .ref_torch import slaney_mel_filter


# ----------------------------------------------------------------------------- S3 tokenizer front-end (pinned)


def s3_log_mel(wav16, n_fft=400, hop=160, n_mels=128):
    """(L,) or (1, L) 16 kHz -> (128, n_frames) log-mel of S3Tokenizer.log_mel_spectrogram."""
    wav16 = torch.as_tensor(wav16, dtype=torch.float32).view(1, -1)
    stft = torch.stft(wav16, n_fft, hop, window=torch.hann_window(n_fft), return_complex=True)
    mag = stft[..., :-1].abs() ** 2
    fb = slaney_mel_filter(16000, n_fft, n_mels, 0.0, 8000.0)
    spec = fb @ mag
    log_spec = torch.clamp(spec, min=1e-10).log10()
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    return ((log_spec + 4.0) / 4.0)[0]


def mel_spectrogram_24k(wav24, n_fft=1920, hop=480, n_mels=80, fmax=8000.0):
    """(1, L) 24 kHz -> (1, 80, frames): the S3Gen prompt-feature extractor (matcha mel_spectrogram, center=False)."""
    y = torch.as_tensor(wav24, dtype=torch.float32).view(1, -1)
    pad = (n_fft - hop) // 2
    y = F.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    spec = torch.view_as_real(torch.stft(y, n_fft, hop_length=hop, win_length=n_fft, window=torch.hann_window(n_fft), center=False,
                                         normalized=False, onesided=True, return_complex=True))
    spec = torch.sqrt(spec.pow(2).sum(-1) + 1e-9)
    fb = slaney_mel_filter(24000, n_fft, n_mels, 0.0, fmax)
    return torch.log(torch.clamp(fb @ spec, min=1e-5))


# ----------------------------------------------------------------------------- Kaldi fbank (unpinned restatement)


def _povey_window(n):
    return torch.hann_window(n, periodic=False).pow(0.85)


def kaldi_mel_banks(num_bins=80, padded=512, sr=16000.0, low=20.0, high=0.0):
    """torchaudio.compliance.kaldi.get_mel_banks (no VTLN): triangular filters on the mel scale 1127 ln(1 + f / 700)."""
    nyq = 0.5 * sr
    high = high + nyq if high <= 0 else high
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
    n_fft_bins = padded // 2
    fft_bin_width = sr / padded
    ml, mh = mel(low), mel(high)
    delta = (mh - ml) / (num_bins + 1)
    b = torch.arange(num_bins, dtype=torch.float64).unsqueeze(1)
    left, center, right = ml + b * delta, ml + (b + 1) * delta, ml + (b + 2) * delta
    melf = 1127.0 * torch.log(1.0 + fft_bin_width * torch.arange(n_fft_bins, dtype=torch.float64) / 700.0).unsqueeze(0)
    up, down = (melf - left) / (center - left), (right - melf) / (right - center)
    return torch.clamp(torch.minimum(up, down), min=0.0).float()  # (num_bins, padded/2)


def kaldi_fbank(wav16, num_mel_bins=80):
    """Kaldi.fbank(waveform (1, L), num_mel_bins=80) with torchaudio defaults: 25 ms / 10 ms frames, snip_edges, dither 0, DC removal,
    pre-emphasis 0.97 (replicate first sample), povey window, 512-point power spectrum, log mel energies floored at float eps.
    Returns (frames, 80)."""
    x = torch.as_tensor(wav16, dtype=torch.float32).view(-1)
    wl, ws, pad = 400, 160, 512
    if x.numel() < wl:
        return torch.zeros(0, num_mel_bins)
    m = 1 + (x.numel() - wl) // ws
    fr = x.unfold(0, wl, ws)[:m].clone()
    fr = fr - fr.mean(1, keepdim=True)
    prev = F.pad(fr.unsqueeze(0), (1, 0), mode="replicate").squeeze(0)
    fr = fr - 0.97 * prev[:, :-1]
    fr = fr * _povey_window(wl)
    fr = F.pad(fr, (0, pad - wl))
    spec = torch.fft.rfft(fr).abs().pow(2.0)
    mel = spec[:, : pad // 2] @ kaldi_mel_banks(num_mel_bins, pad).t()
    return torch.clamp(mel, min=torch.finfo(torch.float32).eps).log()


# ----------------------------------------------------------------------------- CAMPPlus (pinned body)


def _bn(sd, p, x, dim=1, affine=True, eps=1e-5):
    shape = [1] * x.dim()
    shape[dim] = -1
    y = (x - sd[p + ".running_mean"].view(shape)) / torch.sqrt(sd[p + ".running_var"].view(shape) + eps)
    if affine:
        y = y * sd[p + ".weight"].view(shape) + sd[p + ".bias"].view(shape)
    return y


def _res_block(sd, p, x, stride):
    out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], stride=(stride, 1), padding=1)))
    out = _bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], padding=1))
    sc = x
    if p + ".shortcut.0.weight" in sd:
        sc = _bn(sd, p + ".shortcut.1", F.conv2d(x, sd[p + ".shortcut.0.weight"], stride=(stride, 1)))
    return F.relu(out + sc)


def _seg_pool(x, seg_len=100):
    seg = F.avg_pool1d(x, kernel_size=seg_len, stride=seg_len, ceil_mode=True)
    shape = seg.shape
    return seg.unsqueeze(-1).expand(*shape, seg_len).reshape(*shape[:-1], -1)[..., : x.shape[-1]]


CAMPPLUS_BLOCKS = ((12, 3, 1), (24, 3, 2), (16, 3, 2))  # (layers, kernel, dilation) of the three dense blocks (xvector.py:377-379)


def campplus_forward(sd, feats, prefix=""):
    """feats (B, T, 80) mean-normalised Kaldi fbank -> (B, 192) x-vector (CAMPPlus.forward, output_level 'segment')."""
    g = lambda k: prefix + k
    sdp = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)} if prefix else sd
    x = feats.permute(0, 2, 1).unsqueeze(1)
    x = F.relu(_bn(sdp, "head.bn1", F.conv2d(x, sdp["head.conv1.weight"], padding=1)))
    for layer in ("head.layer1", "head.layer2"):
        x = _res_block(sdp, layer + ".0", x, 2)
        x = _res_block(sdp, layer + ".1", x, 1)
    x = F.relu(_bn(sdp, "head.bn2", F.conv2d(x, sdp["head.conv2.weight"], stride=(2, 1), padding=1)))
    x = x.reshape(x.shape[0], x.shape[1] * x.shape[2], x.shape[3])
    x = F.relu(_bn(sdp, "xvector.tdnn.nonlinear.batchnorm", F.conv1d(x, sdp["xvector.tdnn.linear.weight"], stride=2, padding=2)))
    for bi, (n_layers, k, dil) in enumerate(CAMPPLUS_BLOCKS):
        for li in range(n_layers):
            p = f"xvector.block{bi + 1}.tdnnd{li + 1}"
            h = F.conv1d(F.relu(_bn(sdp, p + ".nonlinear1.batchnorm", x)), sdp[p + ".linear1.weight"])
            h = F.relu(_bn(sdp, p + ".nonlinear2.batchnorm", h))
            y = F.conv1d(h, sdp[p + ".cam_layer.linear_local.weight"], padding=(k - 1) // 2 * dil, dilation=dil)
            ctx = h.mean(-1, keepdim=True) + _seg_pool(h)
            ctx = F.relu(F.conv1d(ctx, sdp[p + ".cam_layer.linear1.weight"], sdp[p + ".cam_layer.linear1.bias"]))
            m = torch.sigmoid(F.conv1d(ctx, sdp[p + ".cam_layer.linear2.weight"], sdp[p + ".cam_layer.linear2.bias"]))
            x = torch.cat([x, y * m], 1)
        p = f"xvector.transit{bi + 1}"
        x = F.conv1d(F.relu(_bn(sdp, p + ".nonlinear.batchnorm", x)), sdp[p + ".linear.weight"])
    x = F.relu(_bn(sdp, "xvector.out_nonlinear.batchnorm", x))
    stats = torch.cat([x.mean(-1), x.std(-1, unbiased=True)], -1)
    out = F.conv1d(stats.unsqueeze(-1), sdp["xvector.dense.linear.weight"]).squeeze(-1)
    return _bn(sdp, "xvector.dense.nonlinear.batchnorm", out, affine=False)


def campplus_inference(sd, wav16, prefix=""):
    """CAMPPlus.inference (xvector.py:425-428) for one 16 kHz waveform: fbank -> mean normalisation -> network."""
    f = kaldi_fbank(wav16)
    f = f - f.mean(0, keepdim=True)
    return campplus_forward(sd, f[None], prefix)


# ----------------------------------------------------------------------------- voice encoder (LSTM body pinned; mel front unpinned)

VE_PARTIAL, VE_MELS, VE_SR = 160, 40, 16000


def ve_frame_step(overlap=0.5, rate=1.3):
    step = int(np.round(VE_PARTIAL * (1 - overlap))) if rate is None else int(np.round((VE_SR / rate) / VE_PARTIAL))
    assert 0 < step <= VE_PARTIAL
    return step


def ve_num_wins(n_frames, step, min_coverage=0.8):
    n_wins, rem = divmod(max(n_frames - VE_PARTIAL + step, 0), step)
    if n_wins == 0 or (rem + (VE_PARTIAL - step)) / VE_PARTIAL >= min_coverage:
        n_wins += 1
    return n_wins, VE_PARTIAL + step * (n_wins - 1)


def ve_melspectrogram(wav16):
    """melspectrogram(wav, hp).T of voice_encoder/melspec.py: |STFT(400, hop 160, hann, center reflect)|^2 -> 40 slaney mels, 'amp'.
    Returns (frames, 40) with frames = 1 + len // 160."""
    y = torch.as_tensor(wav16, dtype=torch.float32).view(1, -1)
    spec = torch.stft(y, 400, 160, win_length=400, window=torch.hann_window(400), center=True, pad_mode="reflect", return_complex=True)
    mag = spec.abs() ** 2.0
    fb = slaney_mel_filter(VE_SR, 400, VE_MELS, 0.0, 8000.0)
    return (fb @ mag)[0].t().contiguous()


def lstm_forward(sd, x, prefix="lstm.", n_layers=3):
    """nn.LSTM(batch_first) restated: x (B, T, in) -> final hidden state of the top layer (B, H)."""
    h_last = None
    for l in range(n_layers):
        wi, wh = sd[f"{prefix}weight_ih_l{l}"], sd[f"{prefix}weight_hh_l{l}"]
        b = sd[f"{prefix}bias_ih_l{l}"] + sd[f"{prefix}bias_hh_l{l}"]
        H = wh.shape[1]
        h, c = torch.zeros(x.shape[0], H), torch.zeros(x.shape[0], H)
        pre = x @ wi.t() + b
        outs = []
        for t in range(x.shape[1]):
            g = pre[:, t] + h @ wh.t()
            i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        x = torch.stack(outs, 1)
        h_last = h
    return h_last


def ve_inference(sd, mel, rate=1.3, overlap=0.5, min_coverage=0.8, prefix=""):
    """VoiceEncoder.inference for one utterance: mel (T, 40) -> (256,) L2-normalised utterance embedding."""
    sdp = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)} if prefix else sd
    step = ve_frame_step(overlap, rate)
    n, target = ve_num_wins(mel.shape[0], step, min_coverage)
    if target > mel.shape[0]:
        mel = torch.cat([mel, torch.zeros(target - mel.shape[0], mel.shape[1])])
    parts = torch.stack([mel[i * step: i * step + VE_PARTIAL] for i in range(n)])
    h = lstm_forward(sdp, parts)
    raw = F.relu(h @ sdp["proj.weight"].t() + sdp["proj.bias"])
    emb = raw / torch.linalg.norm(raw, dim=1, keepdim=True)
    m = emb.mean(0)
    return m / torch.linalg.norm(m)


def trim_silence(y, top_db=20.0, frame_length=2048, hop_length=512):
    """librosa.effects.trim(y, top_db)[0]: frames whose RMS (centered frames, zero... reflect-free constant padding) is within top_db of
    the peak RMS are non-silent; keep from the first to the last non-silent frame."""
    y = np.asarray(y, dtype=np.float32)
    pad = frame_length // 2
    yp = np.pad(y, (pad, pad), mode="constant")
    n = 1 + (len(yp) - frame_length) // hop_length
    idx = np.arange(frame_length)[None, :] + hop_length * np.arange(n)[:, None]
    rms = np.sqrt(np.mean(yp[idx].astype(np.float64) ** 2, axis=1))
    db = 20.0 * np.log10(np.maximum(rms, 1e-10)) - 20.0 * np.log10(max(rms.max(), 1e-10))  # power_to_db(rms^2, ref=max)
    nz = np.flatnonzero(db > -top_db)
    if nz.size == 0:
        return y[:0]
    start, end = int(nz[0]) * hop_length, min(len(y), (int(nz[-1]) + 1) * hop_length)
    return y[start:end]


def resample(x, sr_in, sr_out):
    """Band-limited polyphase resampling (scipy.signal.resample_poly, Kaiser beta 5).  Stands in for librosa.resample (soxr) and
    torchaudio.transforms.Resample, neither of which is available: the product uses the identical routine, so the prompt path is
    self-consistent, but it is not bit-comparable to the reference's resamplers."""
    from scipy.signal import resample_poly
    if sr_in == sr_out:
        return np.asarray(x, dtype=np.float32)
    g = math.gcd(int(sr_in), int(sr_out))
    return resample_poly(np.asarray(x, dtype=np.float64), int(sr_out) // g, int(sr_in) // g).astype(np.float32)


# ----------------------------------------------------------------------------- S3TokenizerV2 (third-party: PARITY UNPINNED)

S3TOK = dict(n_mels=128, n_state=1280, n_head=20, n_layer=6, fsmn_k=31)


def s3tok_rope(T, dim=64, theta=10000.0):
    inv = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
    fr = torch.outer(torch.arange(T).float(), inv)
    return torch.cat([fr.cos(), fr.cos()], -1), torch.cat([fr.sin(), fr.sin()], -1)  # (T, 64) rotate-half form


def _rot_half(x):
    return torch.cat([-x[..., x.shape[-1] // 2:], x[..., : x.shape[-1] // 2]], -1)


def s3tokenizer_quantize(sd, mel, prefix="tokenizer."):
    """S3TokenizerV2.quantize restated for one utterance: mel (128, T) -> (T // 4,) int64 ids.
    encoder.conv1/conv2: Conv1d k3 s2 p1 + GELU; 6 x [x += attn(LN(x)); x += mlp(LN(x))]; attention = 20 heads x 64, q and k scaled by
    64^-0.25, rotary (rotate-half) on q, k, fp32 softmax, plus the FSMN memory (depthwise Conv1d k31 over v, + v) added to the head
    output before the output projection; FSQ: Linear 1280 -> 8, tanh * 0.999, round, + 1, base-3 digits."""
    g = lambda k: sd[prefix + k]
    D, Hh, hd = S3TOK["n_state"], S3TOK["n_head"], S3TOK["n_state"] // S3TOK["n_head"]
    x = mel[None]
    x = F.gelu(F.conv1d(x, g("encoder.conv1.weight"), g("encoder.conv1.bias"), stride=2, padding=1))
    x = F.gelu(F.conv1d(x, g("encoder.conv2.weight"), g("encoder.conv2.bias"), stride=2, padding=1))
    x = x.permute(0, 2, 1)  # (1, T', D)
    T = x.shape[1]
    cos, sin = s3tok_rope(T, hd)
    for i in range(S3TOK["n_layer"]):
        p = f"encoder.blocks.{i}."
        h = F.layer_norm(x, (D,), g(p + "attn_ln.weight"), g(p + "attn_ln.bias"))
        q = F.linear(h, g(p + "attn.query.weight"), g(p + "attn.query.bias")).view(1, T, Hh, hd)
        k = F.linear(h, g(p + "attn.key.weight")).view(1, T, Hh, hd)
        v = F.linear(h, g(p + "attn.value.weight"), g(p + "attn.value.bias"))
        q = q * cos[None, :, None] + _rot_half(q) * sin[None, :, None]
        k = k * cos[None, :, None] + _rot_half(k) * sin[None, :, None]
        mem = F.conv1d(v.transpose(1, 2), g(p + "attn.fsmn_block.weight"), padding=15, groups=D).transpose(1, 2) + v
        sc = hd ** -0.25
        a = torch.softmax(((q * sc).permute(0, 2, 1, 3) @ (k * sc).permute(0, 2, 3, 1)).float(), -1)
        o = (a @ v.view(1, T, Hh, hd).permute(0, 2, 1, 3)).permute(0, 2, 1, 3).reshape(1, T, D)
        x = x + F.linear(o + mem, g(p + "attn.out.weight"), g(p + "attn.out.bias"))
        h = F.layer_norm(x, (D,), g(p + "mlp_ln.weight"), g(p + "mlp_ln.bias"))
        x = x + F.linear(F.gelu(F.linear(h, g(p + "mlp.0.weight"), g(p + "mlp.0.bias"))), g(p + "mlp.2.weight"), g(p + "mlp.2.bias"))
    hq = F.linear(x[0], g("quantizer._codebook.project_down.weight"), g("quantizer._codebook.project_down.bias")).float()
    hq = (hq.tanh() * 0.9990000128746033).round() + 1
    return (hq * (3 ** torch.arange(8)).float()).sum(-1).long(), hq


def s3_tokenize(sd, wav16, max_len=None, prefix="tokenizer."):
    """S3Tokenizer.forward for one waveform: log-mel -> optional truncation to 4 * max_len frames -> quantize."""
    mel = s3_log_mel(wav16)
    if max_len is not None:
        mel = mel[..., : max_len * 4]
    return s3tokenizer_quantize(sd, mel, prefix)[0]
