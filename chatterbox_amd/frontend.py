"""Voice-prompt analysis and the voice-conversion front-end on MI355X:

  S3TokenizerEngine   S3Tokenizer.forward = log-mel (reference s3tokenizer.py:128-168) + S3TokenizerV2.quantize (third-party
                      `s3tokenizer` package: AudioEncoderV2 + FSQ -- PARITY UNPINNED, restated from SURVEY.md A.6)
  Mel24kExtractor     the S3Gen prompt-feature extractor (reference s3gen/utils/mel.py:41-85)
  CAMPPlusEngine      x-vector speaker encoder (reference s3gen/xvector.py:60-428) on Kaldi fbank features (torchaudio
                      compliance.kaldi.fbank: unpinned restatement)
  VoiceEncoderEngine  3-layer LSTM utterance embedding over overlapping partials (reference voice_encoder.py:139-200)
  PromptAnalyzer      S3Gen.embed_ref (s3gen.py:118-171) + ChatterboxTTS.prepare_conditionals (tts.py:182-206)

Like the rest of the package, Python only sequences launches of libcbx_hip.so.  Every contraction is a GEMM on the shared MFMA
kernels: a framed DFT is `frames @ basis^T` where `frames` is the waveform itself viewed as overlapping rows (row stride = hop), so
no frame matrix is materialised; Kaldi's per-frame DC removal / pre-emphasis / povey window are linear and folded into its DFT basis;
the 2-D convolutions of CAMPPlus' front module become 1-D convolutions over time whose "channels" are (channel, frequency) pairs
(block-Toeplitz weights built once at load, BatchNorm folded in).  This path runs once per voice, in exact fp32 (precision 1).
Host-side signal conditioning that the reference also does on the CPU (file decoding, resampling, silence trimming) is numpy / scipy.
"""
import math

import numpy as np
import torch

from . import ops

S3_SR, S3GEN_SR = 16000, 24000


def _slaney_mel(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """librosa.filters.mel (slaney scale, slaney norm) -- the filterbank of s3tokenizer.py:40-44, utils/mel.py:56, melspec.py:9-17."""
    fmax = fmax or sr / 2.0
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    h2m = lambda f: np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)
    m2h = lambda m: np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)
    ff = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mf = m2h(np.linspace(h2m(np.float64(fmin)), h2m(np.float64(fmax)), n_mels + 2))
    fd = np.diff(mf)
    ramps = mf[:, None] - ff[None, :]
    w = np.maximum(0, np.minimum(-ramps[:-2] / fd[:-1, None], ramps[2:] / fd[1:, None]))
    return torch.from_numpy((w * (2.0 / (mf[2:n_mels + 2] - mf[:n_mels]))[:, None]).astype(np.float32))


def _pad_cols(w, mult=4):
    """Zero-pad the K dimension of a (N, K) matrix to a multiple of `mult` (16-byte rows for the GEMM loader)."""
    k = (w.shape[1] + mult - 1) // mult * mult
    if k == w.shape[1]:
        return w.contiguous()
    out = torch.zeros(w.shape[0], k, dtype=w.dtype)
    out[:, : w.shape[1]] = w
    return out


class FramedDFT:
    """spec[t] = [Re X_t(0..F-1) | Im X_t(0..F-1)] for frames x[t*hop : t*hop + n_win] of a waveform, as ONE GEMM whose A operand is
    the waveform viewed as overlapping rows.  `pre` (n_win x n_win, optional) is a linear per-frame map applied before the window
    (Kaldi: DC removal + pre-emphasis); zero-padding to n_fft is implicit (the basis only has n_win columns)."""

    def __init__(self, n_fft, window, dev, pre=None, n_bins=None):
        n_win = window.numel()
        self.n_win, self.F = n_win, (n_bins or n_fft // 2 + 1)
        n = torch.arange(n_win, dtype=torch.float64)
        f = torch.arange(self.F, dtype=torch.float64)
        ang = 2.0 * math.pi * f[:, None] * n[None, :] / n_fft
        basis = torch.cat([torch.cos(ang), -torch.sin(ang)], 0) * window.double()[None, :]
        if pre is not None:
            basis = basis @ pre.double()
        self.basis = basis.float().contiguous().to(dev)

    def __call__(self, wav, hop, n_frames):
        """wav: 1-D device tensor holding at least (n_frames - 1) * hop + n_win samples.  Returns (n_frames, 2F)."""
        assert wav.dim() == 1 and wav.numel() >= (n_frames - 1) * hop + self.n_win and hop % 4 == 0
        spec = torch.empty(n_frames, 2 * self.F, device=wav.device)
        with ops.gemm_precision(1):
            ops.gemm(wav, self.basis, spec, M=n_frames, N=2 * self.F, K=self.n_win, lda=hop, ldw=self.n_win, ldc=2 * self.F)
        return spec


class _MelFront:
    """|DFT|^p -> mel filterbank, channel-last (frames, n_mels)."""

    def __init__(self, sr, n_fft, n_mels, dev, fmax=None, mode=0, eps=0.0):
        self.dev, self.n_fft, self.mode, self.eps = dev, n_fft, mode, eps
        self.dft = FramedDFT(n_fft, torch.hann_window(n_fft), dev)
        self.fb = _pad_cols(_slaney_mel(sr, n_fft, n_mels, 0.0, fmax)).to(dev)  # (n_mels, F padded to a multiple of 4)

    def mel(self, wav_padded, hop, n_frames):
        spec = self.dft(wav_padded, hop, n_frames)
        pw = torch.zeros(n_frames, self.fb.shape[1], device=self.dev)
        ops.cplx_power(spec, pw[:, : self.dft.F], self.mode, self.eps)
        out = torch.empty(n_frames, self.fb.shape[0], device=self.dev)
        with ops.gemm_precision(1):
            ops.linear(pw, self.fb, out)
        return out


# ----------------------------------------------------------------------------- S3 tokenizer


class S3TokenizerEngine:
    N_STATE, N_HEAD, N_MELS = 1280, 20, 128

    @ops.on_device
    def __init__(self, sd, device="cuda", prefix="tokenizer."):
        self.dev = dev = torch.device(device)
        self.front = _MelFront(S3_SR, 400, self.N_MELS, dev, fmax=8000.0, mode=0)
        g = lambda k: sd[prefix + k].float()
        d = lambda t: t.contiguous().to(dev)
        self.available = (prefix + "encoder.conv1.weight") in sd
        if not self.available:
            return
        conv = lambda w: d(w.permute(0, 2, 1).reshape(w.shape[0], -1))  # (N, Cin, k) -> tap-major (N, k * Cin)
        self.c1 = (conv(g("encoder.conv1.weight")), d(g("encoder.conv1.bias")))
        self.c2 = (conv(g("encoder.conv2.weight")), d(g("encoder.conv2.bias")))
        self.blocks = []
        i = 0
        while (prefix + f"encoder.blocks.{i}.attn.query.weight") in sd:
            p = f"encoder.blocks.{i}."
            D = self.N_STATE
            self.blocks.append(dict(
                ln1=(d(g(p + "attn_ln.weight")), d(g(p + "attn_ln.bias"))),
                wqkv=d(torch.cat([g(p + "attn.query.weight"), g(p + "attn.key.weight"), g(p + "attn.value.weight")], 0)),
                bqkv=d(torch.cat([g(p + "attn.query.bias"), torch.zeros(D), g(p + "attn.value.bias")], 0)),  # key has no bias
                fsmn=d(g(p + "attn.fsmn_block.weight").view(D, -1)),
                wo=(d(g(p + "attn.out.weight")), d(g(p + "attn.out.bias"))),
                ln2=(d(g(p + "mlp_ln.weight")), d(g(p + "mlp_ln.bias"))),
                w1=(d(g(p + "mlp.0.weight")), d(g(p + "mlp.0.bias"))), w2=(d(g(p + "mlp.2.weight")), d(g(p + "mlp.2.bias")))))
            i += 1
        self.fsq = (d(g("quantizer._codebook.project_down.weight")), d(g("quantizer._codebook.project_down.bias")))
        self._rope = {}

    @ops.on_device
    @torch.inference_mode()
    def log_mel(self, wav16):
        """S3Tokenizer.log_mel_spectrogram: (L,) 16 kHz -> (L // 160, 128) channel-last (the reference returns its transpose)."""
        x = torch.as_tensor(wav16, dtype=torch.float32).view(-1).to(self.dev)
        T = x.numel() // 160  # 1 + L // 160 centred frames, the last one dropped (s3tokenizer.py:158)
        xp = torch.nn.functional.pad(x[None, None], (200, 200), mode="reflect").view(-1).contiguous()
        mel = self.front.mel(xp, 160, T)
        ops.unary(mel, mel, ops.UN_LOG10_CLAMP, 1e-10)
        mx = torch.empty(1, device=self.dev)
        ops.reduce_max(mel, mx)
        ops.unary(mel, mel, ops.UN_FLOOR_AFFINE, 8.0, 4.0, dev_scalar=mx)
        return mel

    def _rope_tables(self, T):
        if T not in self._rope:
            inv = 1.0 / (10000.0 ** (torch.arange(0, 64, 2)[:32].float() / 64))
            fr = torch.outer(torch.arange(T).float(), inv)
            self._rope[T] = (torch.cat([fr.cos(), fr.cos()], -1).contiguous().to(self.dev), torch.cat([fr.sin(), fr.sin()], -1).contiguous().to(self.dev))
        return self._rope[T]

    @ops.on_device
    @torch.inference_mode()
    def quantize(self, mel):
        """S3TokenizerV2.quantize for one utterance: mel (T, 128) channel-last -> (T // 4,) int64 speech tokens."""
        assert self.available, "no S3TokenizerV2 weights (tokenizer.*) in this checkpoint"
        dev, D, H = self.dev, self.N_STATE, self.N_HEAD
        T0 = mel.shape[0]
        T1 = (T0 + 2 - 3) // 2 + 1
        T2 = (T1 + 2 - 3) // 2 + 1
        f = lambda *s: torch.empty(*s, device=dev)
        with ops.gemm_precision(1):
            x1, x = f(1, T1, D), f(1, T2, D)
            ops.conv1d(mel.view(1, T0, -1), self.c1[0], x1, taps=3, cin=self.N_MELS, bias=self.c1[1], stride=2, pad_left=1, act=ops.GELU_ERF)
            ops.conv1d(x1, self.c2[0], x, taps=3, cin=D, bias=self.c2[1], stride=2, pad_left=1, act=ops.GELU_ERF)
            x = x.view(T2, D)
            cos, sin = self._rope_tables(T2)
            pos = torch.arange(T2, dtype=torch.int32, device=dev)
            h, qkv, att, mem, g = f(T2, D), f(T2, 3 * D), f(T2, D), f(1, T2, D), f(T2, 4 * D)
            for b in self.blocks:
                ops.layernorm(x, b["ln1"][0], b["ln1"][1], h, 1e-5)
                ops.linear(h, b["wqkv"], qkv, bias=b["bqkv"])
                ops.rope_kv(qkv, pos, cos, sin, None, None, H)
                q4 = qkv.view(1, T2, 3, H, 64)
                ops.flash_attn(q4[:, :, 0], q4[:, :, 1], q4[:, :, 2], att.view(1, T2, H, 64), 0.125)  # (q 64^-1/4) . (k 64^-1/4)
                ops.dwconv1d(qkv[:, 2 * D:].unsqueeze(0), b["fsmn"], mem, taps=31, pad_left=15, add_input=True)  # FSMN memory over v
                ops.axpby(mem.view(T2, D), att, a=1.0, b=1.0)
                ops.linear(att, b["wo"][0], x, bias=b["wo"][1], residual=x)
                ops.layernorm(x, b["ln2"][0], b["ln2"][1], h, 1e-5)
                ops.linear(h, b["w1"][0], g, bias=b["w1"][1], act=ops.GELU_ERF)
                ops.linear(g, b["w2"][0], x, bias=b["w2"][1], residual=x)
            hq = f(T2, 8)
            ops.linear(x, self.fsq[0], hq, bias=self.fsq[1])
        idx = torch.empty(T2, dtype=torch.int64, device=dev)
        ops.fsq_index(hq, idx)
        return idx

    def __call__(self, wav16, max_len=None):
        """S3Tokenizer.forward for ONE waveform (the reference loops over a list): returns (tokens (1, n), lens (1,))."""
        mel = self.log_mel(wav16)
        if max_len is not None:
            mel = mel[: max_len * 4]
        tok = self.quantize(mel.contiguous())
        return tok[None], torch.tensor([tok.numel()])


# ----------------------------------------------------------------------------- 24 kHz prompt mel


class Mel24kExtractor:
    @ops.on_device
    def __init__(self, device="cuda"):
        self.dev = torch.device(device)
        self.front = _MelFront(S3GEN_SR, 1920, 80, self.dev, fmax=8000.0, mode=1, eps=1e-9)

    @ops.on_device
    @torch.inference_mode()
    def __call__(self, wav24):
        """mel_spectrogram(y).transpose(1, 2): (L,) 24 kHz -> (frames, 80), frames = L // 480."""
        x = torch.as_tensor(wav24, dtype=torch.float32).view(-1).to(self.dev)
        xp = torch.nn.functional.pad(x[None, None], (720, 720), mode="reflect").view(-1).contiguous()
        n = (xp.numel() - 1920) // 480 + 1
        mel = self.front.mel(xp, 480, n)
        return ops.unary(mel, mel, ops.UN_LOG_CLAMP, 1e-5)


# ----------------------------------------------------------------------------- CAMPPlus


def _bn_affine(sd, p, affine=True, eps=1e-5):
    s = 1.0 / torch.sqrt(sd[p + ".running_var"].float() + eps)
    if affine:
        s = s * sd[p + ".weight"].float()
        return s, sd[p + ".bias"].float() - sd[p + ".running_mean"].float() * s
    return s, -sd[p + ".running_mean"].float() * s


def _toeplitz_conv2d(w, f_in, stride, pad, scale=None):
    """Conv2d over (freq, time) with kernel (kf, kt), stride (stride, 1), padding (pad, pad_t) -> the tap-major weight of a Conv1d over
    time whose channels are (c, f) pairs: out channel co * F_out + fo, in channel ci * F_in + fi, tap dt."""
    co, ci, kf, kt = w.shape
    f_out = (f_in + 2 * pad - kf) // stride + 1
    W = torch.zeros(co, f_out, kt, ci, f_in)
    for fo in range(f_out):
        for df in range(kf):
            fi = stride * fo + df - pad
            if 0 <= fi < f_in:
                W[:, fo, :, :, fi] = w[:, :, df, :].permute(0, 2, 1)
    if scale is not None:
        W = W * scale.view(co, 1, 1, 1, 1)
    return W.reshape(co * f_out, kt * ci * f_in), f_out


class CAMPPlusEngine:
    BLOCKS = ((12, 3, 1), (24, 3, 2), (16, 3, 2))

    @ops.on_device
    def __init__(self, sd, device="cuda", prefix="speaker_encoder."):
        self.dev = dev = torch.device(device)
        self.available = (prefix + "head.conv1.weight") in sd
        # Kaldi fbank front: frame -> (I - 11^T / n) -> pre-emphasis -> povey window -> 512-point DFT, all folded into one basis
        n = 400
        dc = torch.eye(n, dtype=torch.float64) - 1.0 / n
        pe = torch.eye(n, dtype=torch.float64)
        pe[torch.arange(1, n), torch.arange(0, n - 1)] = -0.97
        pe[0, 0] = 1.0 - 0.97  # x[-1] := x[0] (replicate)
        povey = torch.hann_window(n, periodic=False, dtype=torch.float64).pow(0.85)
        self.dft = FramedDFT(512, povey, dev, pre=pe @ dc, n_bins=256)
        self.melb = self._kaldi_mel_banks().to(dev)  # (80, 256)
        if not self.available:
            return
        s = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        d = lambda t: t.float().contiguous().to(dev)
        # ---- FCM head: Conv2d -> Toeplitz Conv1d over time, BatchNorm folded (scale into the rows, shift as bias per (c, f))
        self.head = []

        def conv2d(wkey, bnkey, f_in, stride, pad):
            sc, sh = _bn_affine(s, bnkey)
            W, f_out = _toeplitz_conv2d(s[wkey].float(), f_in, stride, pad, scale=sc)
            return dict(w=d(W), b=d(sh.repeat_interleave(f_out)), cin=s[wkey].shape[1] * f_in, taps=s[wkey].shape[3], n=W.shape[0]), f_out

        self.conv1, F = conv2d("head.conv1.weight", "head.bn1", 80, 1, 1)
        self.res = []
        for layer in ("head.layer1", "head.layer2"):
            for j in (0, 1):
                q = f"{layer}.{j}"
                stride = 2 if j == 0 else 1
                c1, F1 = conv2d(q + ".conv1.weight", q + ".bn1", F, stride, 1)
                c2, _ = conv2d(q + ".conv2.weight", q + ".bn2", F1, 1, 1)
                sc = conv2d(q + ".shortcut.0.weight", q + ".shortcut.1", F, stride, 0)[0] if (q + ".shortcut.0.weight") in s else None
                self.res.append((c1, c2, sc))
                F = F1
        self.conv2, F = conv2d("head.conv2.weight", "head.bn2", F, 2, 1)
        # ---- TDNN + dense blocks
        conv1 = lambda w: w.float().permute(0, 2, 1).reshape(w.shape[0], -1)  # (N, Cin, k) -> (N, k * Cin)
        sc, sh = _bn_affine(s, "xvector.tdnn.nonlinear.batchnorm")
        self.tdnn = (d(conv1(s["xvector.tdnn.linear.weight"]) * sc[:, None]), d(sh))
        self.blocks, self.transits = [], []
        for bi, (n_layers, k, dil) in enumerate(self.BLOCKS):
            layers = []
            for li in range(n_layers):
                q = f"xvector.block{bi + 1}.tdnnd{li + 1}"
                a1 = _bn_affine(s, q + ".nonlinear1.batchnorm")
                s2, h2 = _bn_affine(s, q + ".nonlinear2.batchnorm")
                layers.append(dict(bn1=(d(a1[0]), d(a1[1])), w1=d(conv1(s[q + ".linear1.weight"]) * s2[:, None]), b1=d(h2),
                                   wl=d(conv1(s[q + ".cam_layer.linear_local.weight"])), k=k, dil=dil,
                                   c1=(d(conv1(s[q + ".cam_layer.linear1.weight"])), d(s[q + ".cam_layer.linear1.bias"])),
                                   c2=(d(conv1(s[q + ".cam_layer.linear2.weight"])), d(s[q + ".cam_layer.linear2.bias"]))))
            self.blocks.append(layers)
            q = f"xvector.transit{bi + 1}"
            a = _bn_affine(s, q + ".nonlinear.batchnorm")
            self.transits.append(((d(a[0]), d(a[1])), d(conv1(s[q + ".linear.weight"]))))
        a = _bn_affine(s, "xvector.out_nonlinear.batchnorm")
        self.out_bn = (d(a[0]), d(a[1]))
        sc, sh = _bn_affine(s, "xvector.dense.nonlinear.batchnorm", affine=False)
        self.dense = (d(conv1(s["xvector.dense.linear.weight"]) * sc[:, None]), d(sh))

    @staticmethod
    def _kaldi_mel_banks(num_bins=80, padded=512, sr=16000.0, low=20.0):
        mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
        ml, mh = mel(low), mel(0.5 * sr)
        delta = (mh - ml) / (num_bins + 1)
        b = torch.arange(num_bins, dtype=torch.float64).unsqueeze(1)
        left, center, right = ml + b * delta, ml + (b + 1) * delta, ml + (b + 2) * delta
        melf = 1127.0 * torch.log(1.0 + (sr / padded) * torch.arange(padded // 2, dtype=torch.float64) / 700.0).unsqueeze(0)
        return torch.clamp(torch.minimum((melf - left) / (center - left), (right - melf) / (right - center)), min=0.0).float().contiguous()

    @ops.on_device
    @torch.inference_mode()
    def fbank(self, wav16):
        """Kaldi.fbank(num_mel_bins=80) - per-utterance mean (xvector.py:46-53): (L,) -> (frames, 80)."""
        x = torch.as_tensor(wav16, dtype=torch.float32).view(-1).to(self.dev).contiguous()
        m = 1 + (x.numel() - 400) // 160
        spec = self.dft(x, 160, m)
        pw = torch.empty(m, 256, device=self.dev)
        ops.cplx_power(spec, pw, 0, 0.0)
        feat = torch.empty(m, 80, device=self.dev)
        with ops.gemm_precision(1):
            ops.linear(pw, self.melb, feat)
        ops.unary(feat, feat, ops.UN_LOG_CLAMP, float(torch.finfo(torch.float32).eps))
        st = torch.empty(160, device=self.dev)
        ops.stats_pool(feat, st)
        neg = torch.empty(80, device=self.dev)
        ops.unary(st[:80].view(1, 80), neg.view(1, 80), ops.UN_AFFINE, -1.0, 0.0)
        return ops.affine_act(feat, feat, torch.ones(80, device=self.dev), neg)

    def _conv(self, c, x, T, relu=True, residual=None, out2_relu=False):
        out = torch.empty(1, T, c["n"], device=self.dev)
        if out2_relu:  # relu(conv + shortcut): the epilogue's second output carries the activation applied after the residual add
            o2 = torch.empty_like(out)
            ops.conv1d(x, c["w"], out, taps=c["taps"], cin=c["cin"], bias=c["b"], pad_left=c["taps"] // 2, residual=residual, out2=o2,
                       act2=ops.LRELU)
            return o2
        return ops.conv1d(x, c["w"], out, taps=c["taps"], cin=c["cin"], bias=c["b"], pad_left=c["taps"] // 2,
                          act=ops.LRELU if relu else ops.NONE)

    @ops.on_device
    @torch.inference_mode()
    def forward(self, feats):
        """CAMPPlus.forward for one utterance: feats (T, 80) -> (192,) x-vector."""
        assert self.available, "no CAMPPlus weights (speaker_encoder.*) in this checkpoint"
        dev = self.dev
        T = feats.shape[0]
        f = lambda *s: torch.empty(*s, device=dev)
        with ops.gemm_precision(1):
            x = self._conv(self.conv1, feats.contiguous().view(1, T, 80), T)
            for c1, c2, sc in self.res:
                h = self._conv(c1, x, T)
                short = x if sc is None else self._conv(sc, x, T, relu=False)
                x = self._conv(c2, h, T, residual=short, out2_relu=True)
            x = self._conv(self.conv2, x, T)  # (1, T, 320), channel = c * 10 + f
            T2 = (T + 4 - 5) // 2 + 1
            ch = 128
            width = ch + self.BLOCKS[0][0] * 32
            buf = f(1, T2, width)
            ops.conv1d(x, self.tdnn[0], buf[:, :, :ch], taps=5, cin=320, bias=self.tdnn[1], stride=2, pad_left=2, act=ops.LRELU)
            n_seg = (T2 + 99) // 100
            for bi, layers in enumerate(self.blocks):
                for li, L in enumerate(layers):
                    cin = ch + li * 32
                    xin = buf[0, :, :cin]
                    a = f(T2, cin)
                    ops.affine_act(xin, a, L["bn1"][0], L["bn1"][1], ops.LRELU)          # BN -> ReLU (pre-activation)
                    h = f(1, T2, 128)
                    ops.linear(a, L["w1"], h.view(T2, 128), bias=L["b1"], act=ops.LRELU)  # 1x1 conv, BN folded, ReLU
                    y = buf[:, :, cin:cin + 32]
                    ops.conv1d(h, L["wl"], y, taps=L["k"], cin=128, dil=L["dil"], pad_left=(L["k"] - 1) // 2 * L["dil"])
                    ctx, c1, m = f(n_seg, 128), f(n_seg, 64), f(n_seg, 32)
                    ops.seg_context(h.view(T2, 128), ctx, 100)
                    ops.linear(ctx, L["c1"][0], c1, bias=L["c1"][1], act=ops.LRELU)
                    ops.linear(c1, L["c2"][0], m, bias=L["c2"][1])
                    ops.seg_gate_mul(buf[0, :, cin:cin + 32], m, 100)
                ch += len(layers) * 32
                (sc, sh), wt = self.transits[bi]
                a = f(T2, ch)
                ops.affine_act(buf[0], a, sc, sh, ops.LRELU)
                ch //= 2
                nxt_width = ch + (self.BLOCKS[bi + 1][0] * 32 if bi + 1 < len(self.BLOCKS) else 0)
                nbuf = f(1, T2, nxt_width)
                ops.linear(a, wt, nbuf[0, :, :ch])
                buf = nbuf
            a = f(T2, ch)
            ops.affine_act(buf[0], a, self.out_bn[0], self.out_bn[1], ops.LRELU)
            st = f(1, 2 * ch)
            ops.stats_pool(a, st.view(-1))
            out = f(1, 192)
            ops.linear(st, self.dense[0], out, bias=self.dense[1])
        return out.view(-1)

    def inference(self, wav16):
        """CAMPPlus.inference([wav]) for one 16 kHz waveform -> (1, 192)."""
        return self.forward(self.fbank(wav16))[None]


# ----------------------------------------------------------------------------- voice encoder


class VoiceEncoderEngine:
    PARTIAL, MELS = 160, 40

    @ops.on_device
    def __init__(self, sd, device="cuda"):
        self.dev = dev = torch.device(device)
        self.front = _MelFront(S3_SR, 400, self.MELS, dev, fmax=8000.0, mode=0)
        d = lambda t: t.float().contiguous().to(dev)
        self.layers = []
        for l in range(3):
            wi = sd[f"lstm.weight_ih_l{l}"].float()
            self.layers.append(dict(wi=d(_pad_cols(wi, 64) if l == 0 else wi), wh=d(sd[f"lstm.weight_hh_l{l}"]),
                                    b=d(sd[f"lstm.bias_ih_l{l}"].float() + sd[f"lstm.bias_hh_l{l}"].float())))
        self.proj = (d(sd["proj.weight"]), d(sd["proj.bias"]))

    @ops.on_device
    @torch.inference_mode()
    def melspectrogram(self, wav16):
        """melspectrogram(wav, hp).T (voice_encoder/melspec.py): (L,) -> (1 + L // 160, 40) power mels."""
        x = torch.as_tensor(wav16, dtype=torch.float32).view(-1).to(self.dev)
        n = 1 + x.numel() // 160
        xp = torch.nn.functional.pad(x[None, None], (200, 200), mode="reflect").view(-1).contiguous()
        return self.front.mel(xp, 160, n)

    @staticmethod
    def frame_step(overlap=0.5, rate=1.3):
        return int(np.round(160 * (1 - overlap))) if rate is None else int(np.round((S3_SR / rate) / 160))

    @staticmethod
    def num_wins(n_frames, step, min_coverage=0.8):
        n_wins, rem = divmod(max(n_frames - 160 + step, 0), step)
        if n_wins == 0 or (rem + (160 - step)) / 160 >= min_coverage:
            n_wins += 1
        return n_wins, 160 + step * (n_wins - 1)

    def _l2norm(self, x, out):
        C = x.shape[1]
        # F.normalize: x / max(||x||, 1e-12).  eps = 1e-24 / C under the root: identical for any representable non-zero row, 0 (not NaN) for a zero row
        return ops.layernorm(x, torch.ones(C, device=self.dev), None, out, 1e-24 / C, rms=True, scale=1.0 / math.sqrt(C))

    @ops.on_device
    @torch.inference_mode()
    def inference(self, mel, rate=1.3, overlap=0.5, min_coverage=0.8):
        """VoiceEncoder.inference for one utterance: mel (T, 40) -> (256,) utterance embedding (L2-normalised mean of the L2-normalised
        partial embeddings)."""
        dev, P, H = self.dev, self.PARTIAL, 256
        step = self.frame_step(overlap, rate)
        n, target = self.num_wins(mel.shape[0], step, min_coverage)
        melp = torch.zeros(max(target, mel.shape[0]), 64, device=dev)  # K padded 40 -> 64 for the input projection GEMM
        melp[: mel.shape[0], : self.MELS] = mel
        idx = (torch.arange(n, device=dev)[:, None] * step + torch.arange(P, device=dev)[None, :]).view(-1)
        x = melp.index_select(0, idx).contiguous()  # (n * 160, 64): partial b = rows b*160 .. b*160+159
        embs = []
        for b0 in range(0, n, 64):  # the recurrent GEMV serves <= 64 rows
            nb = min(64, n - b0)
            seq = x.view(n, P, -1)[b0:b0 + nb]
            hT = None
            for li, L in enumerate(self.layers):
                pre = torch.empty(nb, P, 4 * H, device=dev)
                with ops.gemm_precision(1):
                    ops.linear(seq.reshape(nb * P, -1), L["wi"], pre.view(nb * P, 4 * H), bias=L["b"])
                h, c = torch.zeros(nb, H, device=dev), torch.zeros(nb, H, device=dev)
                hh, outs = torch.empty(nb, 4 * H, device=dev), torch.empty(nb, P, H, device=dev)
                for t in range(P):
                    ops.gemv(h, L["wh"], hh, nw=4)
                    ops.lstm_cell(pre[:, t], hh, c, outs[:, t])
                    h = outs[:, t]
                seq, hT = outs, h
            raw = torch.empty(nb, H, device=dev)
            with ops.gemm_precision(1):
                ops.linear(hT.contiguous(), self.proj[0], raw, bias=self.proj[1], act=ops.LRELU)
            embs.append(self._l2norm(raw, torch.empty_like(raw)))
        e = torch.cat(embs)
        if n > 1:
            st = torch.empty(2 * H, device=dev)
            ops.stats_pool(e, st)
            m = st[:H].clone().view(1, H)
        else:
            m = e
        return self._l2norm(m, torch.empty_like(m)).view(-1)

    def embeds_from_wavs(self, wavs, sample_rate=S3_SR, trim_top_db=20.0, rate=1.3):
        """VoiceEncoder.embeds_from_wavs (as_spk=False): list of numpy waveforms -> (B, 256) CPU tensor."""
        out = []
        for w in wavs:
            w = resample(np.asarray(w, dtype=np.float32), sample_rate, S3_SR)
            if trim_top_db:
                w = trim_silence(w, trim_top_db)
            out.append(self.inference(self.melspectrogram(torch.from_numpy(w)), rate=rate).cpu())
        return torch.stack(out)


# ----------------------------------------------------------------------------- host-side signal conditioning (CPU, as in the reference)


def resample(x, sr_in, sr_out):
    """Band-limited polyphase resampling (scipy.signal.resample_poly).  Stands in for librosa.resample / torchaudio Resample (neither
    is installable here): not bit-comparable with the reference's resamplers -- INTEGRATION.md, known deviations."""
    if int(sr_in) == int(sr_out):
        return np.asarray(x, dtype=np.float32)
    from scipy.signal import resample_poly
    g = math.gcd(int(sr_in), int(sr_out))
    return resample_poly(np.asarray(x, dtype=np.float64), int(sr_out) // g, int(sr_in) // g).astype(np.float32)


def trim_silence(y, top_db=20.0, frame_length=2048, hop_length=512):
    """librosa.effects.trim(y, top_db)[0] restated: keep from the first to the last frame whose RMS is within top_db of the peak."""
    y = np.asarray(y, dtype=np.float32)
    pad = frame_length // 2
    yp = np.pad(y, (pad, pad), mode="constant")
    n = 1 + (len(yp) - frame_length) // hop_length
    idx = np.arange(frame_length)[None, :] + hop_length * np.arange(n)[:, None]
    rms = np.sqrt(np.mean(yp[idx].astype(np.float64) ** 2, axis=1))
    db = 20.0 * np.log10(np.maximum(rms, 1e-10)) - 20.0 * np.log10(max(rms.max(), 1e-10))
    nz = np.flatnonzero(db > -top_db)
    if nz.size == 0:
        return y[:0]
    return y[int(nz[0]) * hop_length: min(len(y), (int(nz[-1]) + 1) * hop_length)]


def load_wav(path, sr):
    """librosa.load(path, sr=sr): decode (scipy.io.wavfile: PCM / float WAV), mix down to mono, resample."""
    from scipy.io import wavfile
    rate, data = wavfile.read(str(path))
    if data.dtype.kind == "i":
        data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    elif data.dtype.kind == "u":
        data = (data.astype(np.float32) - 128.0) / 128.0
    data = data.astype(np.float32)
    if data.ndim == 2:
        data = data.mean(axis=1)
    return resample(data, rate, sr), sr


# ----------------------------------------------------------------------------- embed_ref / prepare_conditionals


class PromptAnalyzer:
    """S3Gen.embed_ref (s3gen.py:118-171) and the voice-prompt half of prepare_conditionals (tts.py:182-206, tts_turbo.py:241-270,
    mtl_tts.py:253-277): everything that turns a reference waveform into `Conditionals`."""
    ENC_COND_LEN, DEC_COND_LEN = 6 * S3_SR, 10 * S3GEN_SR

    def __init__(self, s3gen_sd, ve_sd=None, device="cuda"):
        self.dev = torch.device(device)
        self.tokenizer = S3TokenizerEngine(s3gen_sd, device)
        self.speaker_encoder = CAMPPlusEngine(s3gen_sd, device)
        self.mel = Mel24kExtractor(device)
        self.ve = VoiceEncoderEngine(ve_sd, device) if ve_sd is not None else None

    def embed_ref(self, ref_wav, ref_sr):
        """-> dict(prompt_token (1, n), prompt_token_len, prompt_feat (1, 2n, 80), prompt_feat_len=None, embedding (1, 192))."""
        w = np.asarray(ref_wav, dtype=np.float32).reshape(-1)
        w24 = resample(w, ref_sr, S3GEN_SR)
        w16 = resample(w, ref_sr, S3_SR)
        feat = self.mel(torch.from_numpy(w24))[None]
        xvec = self.speaker_encoder.inference(torch.from_numpy(w16))
        tok, tlen = self.tokenizer(torch.from_numpy(w16))
        if feat.shape[1] != 2 * tok.shape[1]:  # s3gen.py:152-158
            tok = tok[:, : feat.shape[1] // 2]
            tlen = torch.tensor([tok.shape[1]])
        return dict(prompt_token=tok, prompt_token_len=tlen, prompt_feat=feat, prompt_feat_len=None, embedding=xvec)

    def t3_prompt(self, ref_16k_wav, plen, enc_cond_len=None):
        """(speaker_emb (1, 256), cond_prompt_speech_tokens (1, <= plen)) of T3Cond.  enc_cond_len: samples of the 16 kHz prompt that
        are tokenised -- 6 s for English / Multilingual (tts.py:107,194), 15 s for Turbo / Nano (tts_turbo.py:112,258: 375 tokens)."""
        assert self.ve is not None, "no voice-encoder weights (ve.safetensors) loaded"
        tokens = None
        if plen:
            cut = self.ENC_COND_LEN if enc_cond_len is None else int(enc_cond_len)
            tokens, _ = self.tokenizer(torch.from_numpy(np.asarray(ref_16k_wav[:cut], dtype=np.float32)), max_len=plen)
        ve = self.ve.embeds_from_wavs([ref_16k_wav], sample_rate=S3_SR).mean(0, keepdim=True)
        return ve, tokens
