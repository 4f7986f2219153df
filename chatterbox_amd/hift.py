"""HiFT vocoder (mel -> 24 kHz waveform) on MI355X: host-side mirror of `HiFTGenerator.inference`
(reference models/s3gen/hifigan.py:462-474, decode :412-444; F0 predictor f0_predictor.py:52-55).

Channel-last activations (B, time, C); every Conv1d / ConvTranspose1d is the implicit-GEMM kernel:
  * weight_norm folded at load; ConvTranspose1d phase-packed into a 3-tap stride-1 conv whose (T, s*Cout) output IS
    the (T*s, Cout) upsampled tensor (no scatter, no zero-stuffing);
  * Snake / leaky-ReLU / ELU ride in GEMM epilogues (second output = activation the NEXT conv needs), the residual
    adds, the source fusion `x + si` and the 1/3 average of the three ResBlocks are epilogue accumulations;
  * ragged batches: non-causal convs zero-fill beyond each row's own length (`lens`), so every valid sample equals
    the batch-1 reference value.
"""
import os

import torch

from . import ops, weights


class HiFTEngine:
    UPS = ((8, 16, 4), (5, 11, 3), (3, 7, 2))  # (stride, kernel, padding)
    SRC_DOWN = ((15, 30, 7), (3, 6, 1), (1, 1, 0))  # (stride, kernel, padding) of source_downs
    RB_K = (3, 7, 11)
    SRC_RB_K = (7, 7, 11)

    @ops.on_device
    def __init__(self, sd, device="cuda", precision=None):
        self.dev = dev = torch.device(device)
        # numerics policy of the decoder convs; the F0 predictor always runs exact (its output is integrated into a phase
        # over ~10^5 samples, which amplifies any error in f0)
        self.precision = int(os.environ.get("CBX_S3GEN_PRECISION", "16")) if precision is None else int(precision)
        ops.enable_range_flag(dev)  # raised by a precision-16 launch that meets an operand outside the fp16 range (engine.py repeats at 6)
        d = lambda t: t.float().contiguous().to(dev)
        h = "mel2wav."
        fw = lambda p: weights.fold_weight_norm(sd, p)
        self.f0 = []
        for j, cin in zip((0, 2, 4, 6, 8), (80, 512, 512, 512, 512)):
            p = h + f"f0_predictor.condnet.{j}"
            self.f0.append((d(weights.pack_conv(fw(p))), d(sd[p + ".bias"]), cin))
        self.f0_cls = (d(sd[h + "f0_predictor.classifier.weight"]), d(sd[h + "f0_predictor.classifier.bias"]))
        self.src_w = d(sd[h + "m_source.l_linear.weight"].view(-1))
        self.src_b = float(sd[h + "m_source.l_linear.bias"].view(-1)[0])
        self.conv_pre = (d(weights.pack_conv(fw(h + "conv_pre"))), d(sd[h + "conv_pre.bias"]))
        self.ups = []
        for i, (s, k, p) in enumerate(self.UPS):
            wp, bp = weights.pack_conv_transpose(fw(h + f"ups.{i}"), sd[h + f"ups.{i}.bias"], s, p)
            self.ups.append((d(wp), d(bp)))
        self.src_down = []
        for i in range(3):
            w = sd[h + f"source_downs.{i}.weight"]
            self.src_down.append((d(weights.pack_conv(w, cin_pad=32)), d(sd[h + f"source_downs.{i}.bias"])))

        def resblock(p):
            out = []
            for j in range(3):
                out.append(dict(c1=(d(weights.pack_conv(fw(p + f"convs1.{j}"))), d(sd[p + f"convs1.{j}.bias"])),
                                c2=(d(weights.pack_conv(fw(p + f"convs2.{j}"))), d(sd[p + f"convs2.{j}.bias"])),
                                a1=d(sd[p + f"activations1.{j}.alpha"]), a2=d(sd[p + f"activations2.{j}.alpha"])))
            return out

        self.src_rb = [resblock(h + f"source_resblocks.{i}.") for i in range(3)]
        self.rb = [resblock(h + f"resblocks.{i}.") for i in range(9)]
        wpost = fw(h + "conv_post")  # (18, 64, 7) -> output columns padded to 32 for the iSTFT kernel's row stride
        self.conv_post = (d(weights.pack_conv(wpost)), d(sd[h + "conv_post.bias"]))
        # decode() and the F0 predictor + source through the stage-level C entry points cbx_hift_decode / cbx_hift_f0_source (ABI v12; the same
        # launches with the same arguments as the Python sequencing below -- bit-identical results on the MI355X: tests/test_zzz_stage_seams_gpu.py).
        # The default since round 5, so every vocoder golden passes through them; the identity tests flip this attribute.
        self.c_seam = True
        self._c_static = None

    # ------------------------------------------------------------------ pieces
    def _resblock(self, rb, k, x, a_first, C, lens, out, alpha, beta, out2=None, act2=ops.NONE, act2_slope=0.0, ws=None):
        """ResBlock.forward (hifigan.py:155-161).  x (B,L,C) is NOT modified; a_first = snake(x, alpha1[0]).
        The last conv accumulates `out = beta*out + alpha*(xt + x)` (and optionally out2 = act2(out))."""
        cur_x, cur_a = x, a_first
        t1, xa, xb, an = ws["t1"], ws["xa"], ws["xb"], ws["an"]
        for j, dil in enumerate((1, 3, 5)):
            r = rb[j]
            pad1 = (k * dil - dil) // 2
            ops.conv1d(cur_a, r["c1"][0], t1, taps=k, cin=C, bias=r["c1"][1], dil=dil, pad_left=pad1, lens=lens, act=ops.SNAKE,
                       act_param=r["a2"])
            if j < 2:
                nxt = xa if cur_x is not xa else xb
                ops.conv1d(t1, r["c2"][0], nxt, taps=k, cin=C, bias=r["c2"][1], pad_left=(k - 1) // 2, lens=lens, residual=cur_x,
                           out2=an, act2=ops.SNAKE, act2_param=rb[j + 1]["a1"])
                cur_x, cur_a = nxt, an
            else:
                ops.conv1d(t1, r["c2"][0], out, taps=k, cin=C, bias=r["c2"][1], pad_left=(k - 1) // 2, lens=lens, residual=cur_x,
                           alpha=alpha, beta=beta, out2=out2, act2=act2, act2_slope=act2_slope)

    @ops.on_device
    @torch.inference_mode()
    def f0_predict(self, mel, lens=None):
        """ConvRNNF0Predictor.forward: mel (B,T,80) -> f0 (B,T)."""
        B, T, _ = mel.shape
        bufs = [torch.empty(B, T, 512, device=self.dev), torch.empty(B, T, 512, device=self.dev)]
        x = mel
        for n, (w, bias, cin) in enumerate(self.f0):
            ops.conv1d(x, w, bufs[n % 2], taps=3, cin=cin, bias=bias, pad_left=1, lens=lens, act=ops.ELU)
            x = bufs[n % 2]
        f0 = torch.empty(B * T, 1, device=self.dev)
        ops.linear(x.view(B * T, 512), self.f0_cls[0], f0, bias=self.f0_cls[1], act=ops.ABS)
        return f0.view(B, T)

    @ops.on_device
    @torch.inference_mode()
    def source(self, f0, phase, noise):
        """f0_upsamp + SourceModuleHnNSF (hifigan.py:467-469, 201-231, 267-283) -> s (B, 480*T)."""
        B, T = f0.shape
        s = torch.empty(B, 480 * T, device=self.dev)
        cum = torch.empty(B, 9, T, dtype=torch.float64, device=self.dev)
        ops.hift_source(f0, phase.reshape(B, 9).contiguous(), noise.contiguous(), self.src_w, self.src_b, s, cum)
        return s

    def _decode_c(self, mel, s, lens, fade):
        """decode() as ONE call of cbx_hift_decode: the constant half of cbx_hift_t (weights) is built once per engine."""
        import ctypes

        from ._lib import HiftDecode, check, lib
        p = ops._p
        if self._c_static is None:
            d = HiftDecode()
            d.conv_pre_w, d.conv_pre_b, d.conv_post_w, d.conv_post_b = p(self.conv_pre[0]), p(self.conv_pre[1]), p(self.conv_post[0]), p(self.conv_post[1])
            for i in range(3):
                d.ups_w[i], d.ups_b[i], d.src_down_w[i], d.src_down_b[i] = p(self.ups[i][0]), p(self.ups[i][1]), p(self.src_down[i][0]), p(self.src_down[i][1])
            for dst, src in [(d.src_rb[i], self.src_rb[i]) for i in range(3)] + [(d.rb[i], self.rb[i]) for i in range(9)]:
                for j in range(3):
                    dst.c1_w[j], dst.c1_b[j], dst.c2_w[j], dst.c2_b[j] = p(src[j]["c1"][0]), p(src[j]["c1"][1]), p(src[j]["c2"][0]), p(src[j]["c2"][1])
                    dst.a1[j], dst.a2[j] = p(src[j]["a1"]), p(src[j]["a2"])
            self._c_static = d
        d = HiftDecode()
        ctypes.memmove(ctypes.byref(d), ctypes.byref(self._c_static), ctypes.sizeof(HiftDecode))
        dev, (B, T, _) = self.dev, mel.shape
        f = lambda *sh: torch.empty(*sh, device=dev)
        L3 = 120 * T + 1
        wav = f(B, 480 * T)
        lens5 = None if lens is None else torch.stack([lens, lens * 8, lens * 40, lens * 120 + 1, lens * 480]).int().contiguous()
        ws = dict(spec=f(B, L3, 32), post=f(B, L3, 32), x0=f(B, T, 512))
        wide = [f(B, L3, 64) for _ in range(11)]
        d.B, d.precision, d.fade, d.T = B, ops.GEMM_PRECISION, int(bool(fade)), T
        d.mel, d.s, d.wav, d.lens = p(mel), p(s), p(wav), p(lens5)
        d.spec, d.post, d.x0 = p(ws["spec"]), p(ws["post"]), p(ws["x0"])
        for k, t in zip(("xs", "t1", "xa", "xb", "an", "si", "sa", "acc", "a0"), wide):
            setattr(d, k, p(t))
        d.nxt[0], d.nxt[1] = p(wide[9]), p(wide[10])
        check(lib.cbx_hift_decode(ctypes.byref(d), ops._stream()), "cbx_hift_decode")
        return wav

    def _f0_source_c(self, mel, phase, noise, lens):
        """f0_predict + source as ONE call of cbx_hift_f0_source (ABI v12): the same launches with the same arguments."""
        import ctypes

        from ._lib import HiftF0, check, lib
        p, dev, (B, T, _) = ops._p, self.dev, mel.shape
        f = lambda *sh: torch.empty(*sh, device=dev)
        d = HiftF0()
        for n, (w, bias, _cin) in enumerate(self.f0):
            d.f0_w[n], d.f0_b[n] = p(w), p(bias)
        keep = (f(B, T, 512), f(B, T, 512), f(B * T, 1), torch.empty(B, 9, T, dtype=torch.float64, device=dev), phase.reshape(B, 9).contiguous(),
                noise.contiguous())
        s = f(B, 480 * T)
        d.B, d.T, d.mel, d.lens, d.cls_w, d.cls_b, d.src_w, d.src_b = B, T, p(mel), p(lens), p(self.f0_cls[0]), p(self.f0_cls[1]), p(self.src_w), self.src_b
        d.buf0, d.buf1, d.f0, d.cum, d.phase, d.noise, d.s = p(keep[0]), p(keep[1]), p(keep[2]), p(keep[3]), p(keep[4]), p(keep[5]), p(s)
        check(lib.cbx_hift_f0_source(ctypes.byref(d), ops._stream()), "cbx_hift_f0_source")
        return s

    @ops.on_device
    @torch.inference_mode()
    def decode(self, mel, s, lens=None, fade=True):
        """HiFTGenerator.decode: mel (B,T,80), s (B,480T) -> wav (B,480T).  lens (B,) int32 valid mel frames or None."""
        dev, (B, T, _) = self.dev, mel.shape
        if self.c_seam and not ops.TIMER and mel.is_contiguous() and s.is_contiguous() and mel.dtype == torch.float32:
            return self._decode_c(mel, s, lens, fade)
        f = lambda *sh: torch.empty(*sh, device=dev)
        L3 = 120 * T + 1
        spec = f(B, L3, 32)
        ops.hift_stft(s, spec, None if lens is None else (lens * 480).int())
        ln = [None] * 4 if lens is None else [lens, (lens * 8).int(), (lens * 40).int(), (lens * 120 + 1).int()]
        spec_len = None if lens is None else (lens * 120 + 1).int()
        x = f(B, T, 512)
        ops.conv1d(mel, self.conv_pre[0], x, taps=7, cin=80, bias=self.conv_pre[1], pad_left=3, lens=ln[0], act=ops.LRELU,
                   act_slope=0.1)
        Ls = (8 * T, 40 * T, 120 * T + 1)
        Cs = (256, 128, 64)
        for i, (st, k, p) in enumerate(self.UPS):
            C, L = Cs[i], Ls[i]
            Tin = x.shape[1]
            xs = f(B, L, C)
            if i < 2:
                ops.conv1d(x, self.ups[i][0], xs.view(B, Tin, st * C), taps=3, cin=2 * C, bias=self.ups[i][1], pad_left=1, lens=ln[i])
            else:
                # ReflectionPad1d((1,0)) (hifigan.py:421-422): conv output goes to rows 1.., row 0 := row 2
                shifted = torch.as_strided(xs, (B, Tin, st * C), (L * C, st * C, 1), xs.storage_offset() + C)
                ops.conv1d(x, self.ups[i][0], shifted, taps=3, cin=2 * C, bias=self.ups[i][1], pad_left=1, lens=ln[i])
                ops.axpby(xs[:, 2], xs[:, 0], 1.0, 0.0)
            ws = dict(t1=f(B, L, C), xa=f(B, L, C), xb=f(B, L, C), an=f(B, L, C))
            # fusion: x += source_resblock(source_down(s_stft))   (hifigan.py:424-427)
            sd_s, sd_k, sd_p = self.SRC_DOWN[i]
            si, sa = f(B, L, C), f(B, L, C)
            ops.conv1d(spec, self.src_down[i][0], si, taps=sd_k, cin=32, bias=self.src_down[i][1], stride=sd_s, pad_left=sd_p,
                       lens=spec_len, out2=sa, act2=ops.SNAKE, act2_param=self.src_rb[i][0]["a1"])
            self._resblock(self.src_rb[i], self.SRC_RB_K[i], si, sa, C, ln[i + 1], xs, 1.0, 1.0, ws=ws)
            # mean of the three ResBlocks; the last one also emits the leaky-ReLU the next stage consumes
            acc, nxt = f(B, L, C), f(B, L, C)
            a0 = f(B, L, C)
            for j in range(3):
                rb = self.rb[i * 3 + j]
                ops.act(xs.view(B * L, C), a0.view(B * L, C), ops.SNAKE, param=rb[0]["a1"])
                last = j == 2
                self._resblock(rb, self.RB_K[j], xs, a0, C, ln[i + 1], acc, 1.0 / 3, 0.0 if j == 0 else 1.0,
                               out2=nxt if last else None, act2=ops.LRELU, act2_slope=0.1 if i < 2 else 0.01, ws=ws)
            x = nxt
        post = torch.zeros(B, L3, 32, device=dev)
        ops.conv1d(x, self.conv_post[0], post, taps=7, cin=64, bias=self.conv_post[1], pad_left=3, lens=ln[3])
        wav = f(B, 480 * T)
        ops.hift_istft(post, wav, 0.99, 480 if fade else 0)
        return wav

    @ops.on_device
    @torch.inference_mode()
    def inference(self, mel, phase=None, noise=None, lens=None, fade=True, cache_source=None):
        """HiFTGenerator.inference + S3Gen trim_fade.  mel (B,T,80) channel-last.  Returns (wav (B,480T), source (B,480T)).
        cache_source (B, L): the source of an earlier chunk replaces the first L samples (hifigan.py:470-472)."""
        B, T, _ = mel.shape
        if phase is None:
            phase = (torch.rand(B, 9, device=self.dev) * 2 - 1) * 3.141592653589793
            phase[:, 0] = 0
        if noise is None:
            noise = torch.randn(B, 9, 480 * T, device=self.dev)
        if self.c_seam and not ops.TIMER and mel.is_contiguous() and mel.dtype == torch.float32:
            s = self._f0_source_c(mel, phase.to(self.dev).float(), noise.to(self.dev).float(), lens)
        else:
            with ops.gemm_precision(1):
                f0 = self.f0_predict(mel, lens)
            s = self.source(f0, phase.to(self.dev).float(), noise.to(self.dev).float())
        if cache_source is not None and cache_source.shape[1]:
            s[:, : cache_source.shape[1]] = cache_source.to(self.dev)
        with ops.gemm_precision(self.precision):
            return self.decode(mel, s, lens, fade), s
