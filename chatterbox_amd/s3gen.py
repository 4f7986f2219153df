"""S3Gen token->mel on MI355X: host-side mirror of `CausalMaskedDiffWithXvec.inference` (reference
models/s3gen/flow.py:131-198): UpsampleConformerEncoder -> encoder_proj -> 10-step Euler CFM with CFG over the
ConditionalDecoder estimator.  All tensors are channel-last (rows = frames), so every Conv1d/Linear is one call of
the implicit-GEMM kernel and the attention reads Q/K/V straight out of the fused projection buffer.

Batched (the reference is batch-1): utterances are right-padded; causal convs never look right, attention masks
keys by per-row length, and the look-ahead conv zero-fills beyond each row's own length, so every valid frame gets
exactly the value a batch-1 reference call would produce.
"""
import math

import os

import torch

from . import ops, weights


def _dev(t, dev):
    return t.float().contiguous().to(dev)


class FlowEngine:
    @ops.on_device
    def __init__(self, sd, device="cuda", meanflow=False, precision=None):
        self.dev = dev = torch.device(device)
        self.meanflow = meanflow
        # numerics policy of the encoder / CFM GEMMs and attention (cbx_gemm_t.precision): 1 exact fp32 MFMA; 16 f16x3 = the DEFAULT
        # (two fp16 planes, fp32-level error at the bf16x3 cost: every parity tolerance of the fp32 path holds at every BASELINE shape incl.
        # the 60 s waveform; operands must stay inside the fp16 range, which is checked on the device -- engine.py repeats at 6 otherwise);
        # 6 bf16x6 = fp32-level with the fp32 exponent range, 1.45x the matrix-core time; 3 bf16x3 = opt-in (mel-L1 ~1e-5; the 60 s
        # WAVEFORM only at the bf16-mode tolerance, because the F0 -> phase integration amplifies the mel error -- DESIGN.md section 1)
        self.precision = int(os.environ.get("CBX_S3GEN_PRECISION", "16")) if precision is None else int(precision)
        ops.enable_range_flag(dev)  # raised by a precision-16 launch that meets an operand outside the fp16 range (engine.py repeats at 6)
        d = lambda t: _dev(t, dev)
        self.emb = d(sd["flow.input_embedding.weight"])
        self.spk_w, self.spk_b = d(sd["flow.spk_embed_affine_layer.weight"]), d(sd["flow.spk_embed_affine_layer.bias"])
        self.proj_w, self.proj_b = d(sd["flow.encoder_proj.weight"]), d(sd["flow.encoder_proj.bias"])
        e = "flow.encoder."

        def embed(p):
            return dict(w=d(sd[p + "out.0.weight"]), b=d(sd[p + "out.0.bias"]), lnw=d(sd[p + "out.1.weight"]), lnb=d(sd[p + "out.1.bias"]))

        def conf(p):
            a = p + "self_attn."
            wq, bq = sd[a + "linear_q.weight"], sd[a + "linear_q.bias"]
            u, v = sd[a + "pos_bias_u"].reshape(-1), sd[a + "pos_bias_v"].reshape(-1)
            # fused projection [q+u | q+v | k | v]: the two pos-bias copies of q come out of the GEMM for free
            w4 = torch.cat([wq, wq, sd[a + "linear_k.weight"], sd[a + "linear_v.weight"]], 0)
            b4 = torch.cat([bq + u, bq + v, sd[a + "linear_k.bias"], sd[a + "linear_v.bias"]], 0)
            f = p + "feed_forward."
            return dict(w4=d(w4), b4=d(b4), wpos=d(sd[a + "linear_pos.weight"]), wo=d(sd[a + "linear_out.weight"]),
                        bo=d(sd[a + "linear_out.bias"]), ln_mha=(d(sd[p + "norm_mha.weight"]), d(sd[p + "norm_mha.bias"])),
                        ln_ff=(d(sd[p + "norm_ff.weight"]), d(sd[p + "norm_ff.bias"])), w1=d(sd[f + "w_1.weight"]),
                        b1=d(sd[f + "w_1.bias"]), w2=d(sd[f + "w_2.weight"]), b2=d(sd[f + "w_2.bias"]))

        def count(prefix):
            n = 0
            while f"{prefix}{n}.norm_ff.weight" in sd:
                n += 1
            return n

        self.embed, self.up_embed = embed(e + "embed."), embed(e + "up_embed.")
        self.pl1 = (d(weights.pack_conv(sd[e + "pre_lookahead_layer.conv1.weight"])), d(sd[e + "pre_lookahead_layer.conv1.bias"]))
        self.pl2 = (d(weights.pack_conv(sd[e + "pre_lookahead_layer.conv2.weight"])), d(sd[e + "pre_lookahead_layer.conv2.bias"]))
        self.up_conv = (d(weights.pack_conv(sd[e + "up_layer.conv.weight"])), d(sd[e + "up_layer.conv.bias"]))
        self.enc = [conf(e + f"encoders.{i}.") for i in range(count(e + "encoders."))]
        self.up_enc = [conf(e + f"up_encoders.{i}.") for i in range(count(e + "up_encoders."))]
        self.after_norm = (d(sd[e + "after_norm.weight"]), d(sd[e + "after_norm.bias"]))

        # ---- CFM estimator
        q = "flow.decoder.estimator."
        self.t1 = (d(sd[q + "time_mlp.linear_1.weight"]), d(sd[q + "time_mlp.linear_1.bias"]))
        self.t2 = (d(sd[q + "time_mlp.linear_2.weight"]), d(sd[q + "time_mlp.linear_2.bias"]))
        self.t_mix = d(sd[q + "time_embed_mixer.weight"]) if meanflow else None

        def stage(p, tail):
            r = p + "0."
            s = dict(c1=(d(weights.pack_conv(sd[r + "block1.block.0.weight"])), d(sd[r + "block1.block.0.bias"])),
                     n1=(d(sd[r + "block1.block.2.weight"]), d(sd[r + "block1.block.2.bias"])),
                     c2=(d(weights.pack_conv(sd[r + "block2.block.0.weight"])), d(sd[r + "block2.block.0.bias"])),
                     n2=(d(sd[r + "block2.block.2.weight"]), d(sd[r + "block2.block.2.bias"])),
                     res=(d(weights.pack_conv(sd[r + "res_conv.weight"])), d(sd[r + "res_conv.bias"])),
                     cin=sd[r + "res_conv.weight"].shape[1], tb=[])
            for j in range(4):
                t = p + f"1.{j}."
                s["tb"].append(dict(
                    n1=(d(sd[t + "norm1.weight"]), d(sd[t + "norm1.bias"])),
                    wqkv=d(torch.cat([sd[t + f"attn1.to_{c}.weight"] for c in "qkv"], 0)),
                    wo=d(sd[t + "attn1.to_out.0.weight"]), bo=d(sd[t + "attn1.to_out.0.bias"]),
                    n3=(d(sd[t + "norm3.weight"]), d(sd[t + "norm3.bias"])),
                    w1=d(sd[t + "ff.net.0.proj.weight"]), b1=d(sd[t + "ff.net.0.proj.bias"]),
                    w2=d(sd[t + "ff.net.2.weight"]), b2=d(sd[t + "ff.net.2.bias"])))
            if tail:
                s["tail"] = (d(weights.pack_conv(sd[p + "2.weight"])), d(sd[p + "2.bias"]))
            return s

        self.stages = [stage(q + "down_blocks.0.", True)]
        i = 0
        while f"{q}mid_blocks.{i}.0.mlp.1.weight" in sd:
            self.stages.append(stage(q + f"mid_blocks.{i}.", False))
            i += 1
        self.stages.append(stage(q + "up_blocks.0.", True))
        self.n_mid = i
        # all 14 ResNet time-MLPs (Mish -> Linear 1024->256) concatenated: one GEMM gives every block's time bias
        names = [q + "down_blocks.0."] + [q + f"mid_blocks.{k}." for k in range(i)] + [q + "up_blocks.0."]
        self.tmlp_w = d(torch.cat([sd[n + "0.mlp.1.weight"] for n in names], 0))
        self.tmlp_b = d(torch.cat([sd[n + "0.mlp.1.bias"] for n in names], 0))
        self.fin = dict(c=(d(weights.pack_conv(sd[q + "final_block.block.0.weight"])), d(sd[q + "final_block.block.0.bias"])),
                        n=(d(sd[q + "final_block.block.2.weight"]), d(sd[q + "final_block.block.2.bias"])),
                        proj=(d(weights.pack_conv(sd[q + "final_proj.weight"])), d(sd[q + "final_proj.bias"])))
        self._pe_cache, self._tb_cache = {}, {}  # device-resident constants: rel-pos tables per length, time biases per schedule
        # CFM estimator on plane-format operands (round 3; gemm_planes.hip / attention_planes.hip): the f16x3 arithmetic with weights
        # split into their two fp16 planes ONCE (here, lazily) and activations written in plane format by their producers
        self.use_planes = os.environ.get("CBX_PLANES", "1") != "0"
        self.fused_qkv = os.environ.get("CBX_FUSED_QKV", "1") != "0"  # q | k | V^T of a transformer block from one GEMM launch (ABI v8)
        # LayerNorm produced by the epilogue of the N = 256 Linear in front of it (cbx_gemm_pl_t.ln_w, ABI v13): 1 = norm3 from the attention out-projection,
        # 2 = also the NEXT block's norm1 from ff2, 0 (default) = LayerNorm launches of their own.  Parity-green in all three modes, but NOT faster in the
        # flow: in a warm micro-benchmark out-projection + norm3 goes 33.4 -> 28.8 us (ff2 + norm1 38.1 -> 44.0), inside the estimator the row-spanning
        # 64 x 256 launch averages 44.4 us (rocprofv3) where out-projection + LayerNorm cost ~30: serial flow 204.5 (0) / 220.4 (1) ms on one box,
        # 195.4 (0) / 199.8 (2) on another (profiles/r05_bench_layernorm_epilogue_ab.log).  Kept as an opt-in (CBX_FUSED_LN) until the tile has loader waves.
        self.fused_ln = int(os.environ.get("CBX_FUSED_LN", "0"))
        self._pw = None
        # the token encoder (flash rel-pos form) and the Euler loop (plane-format path) through the stage-level C entry points cbx_s3gen_encode /
        # cbx_cfm_solve (ABI v12; the same launches with the same arguments as the Python sequencing below -- bit-identical results on the MI355X:
        # tests/test_zzz_stage_seams_gpu.py).  The default since round 5, so every S3Gen golden passes through them; the Python sequencing stays as
        # the per-kernel-timed path (ops.TIMER) and as the other side of the identity tests (which flip this attribute).
        self.c_seam = True
        self._cfm_static = self._enc_static = None
        # launch geometry of the estimator's plane GEMMs / attention (cbx_gemm_pl_t.tile, cbx_flash_attn_planes_v: per call since ABI v13).  (0, 0) = the
        # library's measured defaults; co_resident(True) = the forms that leave half of every CU to another stream (engine.synthesize_pipelined)
        self.gemm_tile = self.attn_version = 0

    def co_resident(self, on):
        """The CFM estimator's kernels on their co-resident forms (one 8-wave / 4-wave workgroup per CU: profiles/r05_overlap_*) or back on the defaults.
        Same arithmetic, bit-identical mel."""
        from ._lib import ATTN_PL_CORESIDENT, PL_TILE_CORESIDENT
        self.gemm_tile, self.attn_version = (PL_TILE_CORESIDENT, ATTN_PL_CORESIDENT) if on else (0, 0)

    # ------------------------------------------------------------------ conformer encoder
    def _rel_pos_table(self, T, dev=None, dm=512):
        """EspnetRelPositionalEncoding (transformer/embedding.py:224-294): row r <-> relative position T-1-r.  A constant of the
        length: built once and kept on the device (like the RoPE tables), not rebuilt and re-uploaded per call."""
        if T not in self._pe_cache:
            pos = torch.arange(T - 1, -T, -1, dtype=torch.float32)[:, None]
            div = torch.exp(torch.arange(0, dm, 2, dtype=torch.float32) * -(math.log(10000.0) / dm))
            pe = torch.zeros(2 * T - 1, dm)
            pe[:, 0::2] = torch.sin(pos * div)
            pe[:, 1::2] = torch.cos(pos * div)
            if len(self._pe_cache) >= 16:
                self._pe_cache.pop(next(iter(self._pe_cache)))
            self._pe_cache[T] = pe.to(self.dev)
        return self._pe_cache[T]

    def _conformer(self, lw, x, B, T, pe, lens, ws):
        """ConformerEncoderLayer.forward + RelPositionMultiHeadedAttention (encoder_layer.py:160-236, attention.py:249-330)."""
        M = B * T
        h, q4 = ws["h"][:M], ws["q4"][:M]
        ops.layernorm(x, lw["ln_mha"][0], lw["ln_mha"][1], h, 1e-12)
        ops.linear(h, lw["w4"], q4, bias=lw["b4"])
        P = 2 * T - 1
        pp = ws["pp"][:P]
        ops.linear(pe, lw["wpos"], pp)
        q5 = q4.view(B, T, 4, 8, 64)
        att = ws["att"][:M]
        if ws.get("ac") is None:  # flash form: no (T, T) / (T, 2T-1) score tensors (cbx_flash_relpos_f32)
            ops.flash_relpos(q5, pp, att.view(B, T, 8, 64), 0.125, key_lens=lens)
        else:
            ac, bd, pr = ws["ac"], ws["bd"], ws["pr"]
            ops.bmm(q5[:, :, 0].permute(0, 2, 1, 3), q5[:, :, 2].permute(0, 2, 1, 3), ac[..., :T])
            ops.bmm(q5[:, :, 1].permute(0, 2, 1, 3), pp.view(1, P, 8, 64).permute(0, 2, 1, 3).expand(B, 8, P, 64), bd[..., :P])
            ops.softmax_relpos(ac[..., :T], bd, pr, 0.125, key_lens=lens)
            ops.bmm(pr[..., :T], q5[:, :, 3].permute(0, 2, 1, 3), att.view(B, T, 8, 64).permute(0, 2, 1, 3), nn=True)
        ops.linear(att, lw["wo"], x, bias=lw["bo"], residual=x)
        ops.layernorm(x, lw["ln_ff"][0], lw["ln_ff"][1], h, 1e-12)
        f = ws["ff"][:M]
        ops.linear(h, lw["w1"], f, bias=lw["b1"], act=ops.SILU)
        ops.linear(f, lw["w2"], x, bias=lw["b2"], residual=x)

    # bytes of the materialised rel-pos score tensors (ac, bd, probabilities: 4 * T2^2 floats per head) one encoder call may hold; a
    # larger batch is walked in row groups (rows are independent).  60 s of audio = 1.15 GB per utterance: of 288 GB, not a capacity
    # problem, but a bound keeps one long VC batch from taking the allocator's whole pool.
    ENC_SCORE_BYTES = int(float(os.environ.get("CBX_ENC_SCORE_GB", "32")) * 2 ** 30)
    # CBX_ENC_FLASH: "1" (default since round 4) = the encoder's rel-pos attention always runs in the flash form (cbx_flash_relpos_f32: no score
    # tensors, memory O(T), exact fp32 MFMA) -- measured on one box (profiles/r04_prefill_encoder_ab.log): 6.2 ms against 9.0-9.4 for the
    # materialised form at the bench shape (B = 8, 500 tokens), 6.8 against 10.6 at 60 s; "0" = always materialised (ac / bd on the engine's GEMM
    # precision; a batch above ENC_SCORE_BYTES is walked in row groups); "auto" = flash only for the calls the materialised form would have to split.
    ENC_FLASH = os.environ.get("CBX_ENC_FLASH", "1")

    def encode(self, tok, lens):
        """tok (B,N) int64 padded with any valid id, lens (B,) int32 -> mu (B, 2N, 80) channel-last."""
        B, N = tok.shape
        per_row = 8 * 4 * (2 * N) ** 2 * 4
        group = max(1, min(B, self.ENC_SCORE_BYTES // max(per_row, 1)))
        if self.ENC_FLASH == "1" or (self.ENC_FLASH != "0" and group < B):
            return self._encode_rows(tok, lens, flash=True)
        if group >= B:
            return self._encode_rows(tok, lens)
        return torch.cat([self._encode_rows(tok[i:i + group], lens[i:i + group]) for i in range(0, B, group)], 0)

    def _encode_c(self, ids, lens, B, N):
        """_encode_rows (flash form) as ONE call of cbx_s3gen_encode (ABI v12): the same launches with the same arguments."""
        import ctypes

        from ._lib import Conformer, S3Encode, check, lib
        p, dev, T2 = ops._p, self.dev, 2 * N
        if self._enc_static is None:
            def layers(ls):
                arr = (Conformer * max(1, len(ls)))()
                for a, lw in zip(arr, ls):
                    a.ln_mha_w, a.ln_mha_b, a.ln_ff_w, a.ln_ff_b = p(lw["ln_mha"][0]), p(lw["ln_mha"][1]), p(lw["ln_ff"][0]), p(lw["ln_ff"][1])
                    for k in ("w4", "b4", "wpos", "wo", "bo", "w1", "b1", "w2", "b2"):
                        setattr(a, k, p(lw[k]))
                return arr
            d = S3Encode()
            d.n_enc, d.n_up, d.enc, d.up_enc = len(self.enc), len(self.up_enc), layers(self.enc), layers(self.up_enc)
            d.emb, d.after_w, d.after_b, d.proj_w, d.proj_b = p(self.emb), p(self.after_norm[0]), p(self.after_norm[1]), p(self.proj_w), p(self.proj_b)
            d.e_w, d.e_b, d.e_lnw, d.e_lnb = (p(self.embed[k]) for k in ("w", "b", "lnw", "lnb"))
            d.u_w, d.u_b, d.u_lnw, d.u_lnb = (p(self.up_embed[k]) for k in ("w", "b", "lnw", "lnb"))
            d.pl1_w, d.pl1_b, d.pl2_w, d.pl2_b, d.up_w, d.up_b = p(self.pl1[0]), p(self.pl1[1]), p(self.pl2[0]), p(self.pl2[1]), p(self.up_conv[0]), p(self.up_conv[1])
            self._enc_static = (d, d.enc, d.up_enc)
        d = S3Encode()
        ctypes.memmove(ctypes.byref(d), ctypes.byref(self._enc_static[0]), ctypes.sizeof(S3Encode))
        f = lambda *s: torch.empty(*s, device=dev)
        pe, pe2, lens2 = self._rel_pos_table(N, dev), self._rel_pos_table(T2, dev), (lens * 2).to(torch.int32)
        ws = dict(x0=f(B * N, 512), xa=f(B * N, 512), y1=f(B * N, 512), x2=f(B * N, 512), xu=f(B * T2, 512), xb=f(B * T2, 512), h=f(B * T2, 512),
                  q4=f(B * T2, 2048), pp=f(2 * T2, 512), att=f(B * T2, 512), ff=f(B * T2, 2048))
        mu = f(B, T2, 80)
        d.B, d.N, d.precision = B, N, ops.GEMM_PRECISION
        d.ids, d.lens, d.lens2, d.pe, d.pe2, d.mu = p(ids), p(lens), p(lens2), p(pe), p(pe2), p(mu)
        for k, t in ws.items():
            setattr(d, k, p(t))
        check(lib.cbx_s3gen_encode(ctypes.byref(d), ops._stream()), "cbx_s3gen_encode")
        return mu

    @ops.on_device
    def _encode_rows(self, tok, lens, flash=False):
        dev, (B, N) = self.dev, tok.shape
        T2 = 2 * N
        f = lambda *s: torch.empty(*s, device=dev)
        if flash and self.c_seam and not ops.TIMER and lens.dtype == torch.int32 and lens.is_contiguous():
            ids = tok.reshape(-1).clone()
            ids[(torch.arange(N, device=dev)[None, :] >= lens[:, None]).reshape(-1)] = -1
            return self._encode_c(ids, lens, B, N)
        Tp, Pp = (T2 + 3) // 4 * 4, (2 * T2 - 1 + 3) // 4 * 4
        ws = dict(h=f(B * T2, 512), q4=f(B * T2, 2048), pp=f(2 * T2, 512), att=f(B * T2, 512), ff=f(B * T2, 2048), Tp=Tp, Pp=Pp)
        if flash:
            ws.update(ac=None, bd=None, pr=None)
        else:
            ws.update(ac=f(B, 8, T2, Tp), bd=f(B, 8, T2, Pp), pr=f(B, 8, T2, Tp))
        ids = tok.reshape(-1).clone()
        pad = (torch.arange(N, device=dev)[None, :] >= lens[:, None]).reshape(-1)
        ids[pad] = -1  # `input_embedding(token) * mask` (flow.py:161-166): padded rows are zero vectors
        x0 = f(B * N, 512)
        ops.embed(ids, self.emb, x0)

        def embed(ew, xin, M):
            y = f(M, 512)
            ops.linear(xin, ew["w"], y, bias=ew["b"])
            ops.layernorm(y, ew["lnw"], ew["lnb"], y, 1e-5, scale=math.sqrt(512.0))
            return y

        x = embed(self.embed, x0, B * N)
        # PreLookaheadLayer (upsample_encoder.py:81-96): right-pad 3 conv k4 -> leaky_relu -> left-pad 2 conv k3 -> + x
        y1, x3 = f(B, N, 512), x.view(B, N, 512)
        ops.conv1d(x3, self.pl1[0], y1, taps=4, cin=512, bias=self.pl1[1], pad_left=0, lens=lens, act=ops.LRELU, act_slope=0.01)
        x2 = f(B, N, 512)
        ops.conv1d(y1, self.pl2[0], x2, taps=3, cin=512, bias=self.pl2[1], pad_left=2, residual=x3)
        x = x2.view(B * N, 512)
        pe = self._rel_pos_table(N, dev)
        wsN = ws if flash else dict(
            ws, ac=ws["ac"].view(-1)[: B * 8 * N * ((N + 3) // 4 * 4)].view(B, 8, N, (N + 3) // 4 * 4),
            bd=ws["bd"].view(-1)[: B * 8 * N * ((2 * N - 1 + 3) // 4 * 4)].view(B, 8, N, (2 * N - 1 + 3) // 4 * 4),
            pr=ws["pr"].view(-1)[: B * 8 * N * ((N + 3) // 4 * 4)].view(B, 8, N, (N + 3) // 4 * 4))
        for lw in self.enc:
            self._conformer(lw, x, B, N, pe, lens, wsN)
        # Upsample1D (upsample_encoder.py:59-63): nearest x2, left-pad 4, conv k5 -- fused in the A-operand address map
        xu = f(B, T2, 512)
        ops.conv1d(x.view(B, N, 512), self.up_conv[0], xu, taps=5, cin=512, bias=self.up_conv[1], pad_left=4, up=2)
        x = embed(self.up_embed, xu.view(B * T2, 512), B * T2)
        pe2 = self._rel_pos_table(T2, dev)
        lens2 = (lens * 2).to(torch.int32)
        for lw in self.up_enc:
            self._conformer(lw, x, B, T2, pe2, lens2, ws)
        ops.layernorm(x, self.after_norm[0], self.after_norm[1], x, 1e-5)
        mu = f(B, T2, 80)
        ops.linear(x, self.proj_w, mu.view(B * T2, 80), bias=self.proj_b)
        return mu

    # ------------------------------------------------------------------ CFM estimator (decoder.py:243-333)
    def _resnet(self, sw, xin, cin, rows, T, tb, ws, x):
        """CausalResnetBlock1D: block1 -> + time bias -> block2 -> + res_conv(xin), written to `x` (must not alias xin:
        the 1x1 res_conv reads whole input rows while other workgroups write output column tiles)."""
        M = rows * T
        a, b = ws["ra"], ws["rb"]
        ops.conv1d(xin, sw["c1"][0], a, taps=3, cin=cin, bias=sw["c1"][1], pad_left=2)
        a2, b2 = a.view(M, 256), b.view(M, 256)
        ops.layernorm(a2, sw["n1"][0], sw["n1"][1], a2, 1e-5, act=ops.MISH, post_add=tb)
        ops.conv1d(a, sw["c2"][0], b, taps=3, cin=256, bias=sw["c2"][1], pad_left=2)
        ops.layernorm(b2, sw["n2"][0], sw["n2"][1], b2, 1e-5, act=ops.MISH)
        ops.conv1d(xin, sw["res"][0], x, taps=1, cin=cin, bias=sw["res"][1], residual=b)
        return x

    def _tblock(self, tw, x, rows, T, lens, ws):
        """BasicTransformerBlock (matcha/transformer.py:243-316) with diffusers Attention/GELU semantics."""
        M = rows * T
        x2, h, qkv, att, ff = x.view(M, 256), ws["h"], ws["qkv"], ws["att"], ws["ff"]
        fuse = ops.ln_fusable(M, 256)  # norm1 / norm3 folded into the q/k/v and ff1 projections: a statistics pass instead of a LayerNorm pass
        if fuse:
            ops.linear(x2, tw["wqkv"], qkv, ln=(ops.row_stats(x2, ws["stats"]), tw["n1"][0], tw["n1"][1]))
        else:
            ops.layernorm(x2, tw["n1"][0], tw["n1"][1], h, 1e-5)
            ops.linear(h, tw["wqkv"], qkv)
        q5 = qkv.view(rows, T, 3, 8, 64)
        ops.flash_attn(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2], att.view(rows, T, 8, 64), 0.125, key_lens=lens)
        ops.linear(att, tw["wo"], x2, bias=tw["bo"], residual=x2)
        if fuse:
            ops.linear(x2, tw["w1"], ff, bias=tw["b1"], act=ops.GELU_ERF, ln=(ops.row_stats(x2, ws["stats"]), tw["n3"][0], tw["n3"][1]))
        else:
            ops.layernorm(x2, tw["n3"][0], tw["n3"][1], h, 1e-5)
            ops.linear(h, tw["w1"], ff, bias=tw["b1"], act=ops.GELU_ERF)
        ops.linear(ff, tw["w2"], x2, bias=tw["b2"], residual=x2)

    def _estimator(self, xin, rows, T, lens, tbias, ws):
        """One ConditionalDecoder.forward on the packed input xin (rows,T,320) -> ws['v'] (rows,T,80)."""
        st = self.stages
        x = self._resnet(st[0], xin, 320, rows, T, tbias[0], ws, ws["x"])
        for tw in st[0]["tb"]:
            self._tblock(tw, x, rows, T, lens, ws)
        cat = ws["cat"]  # [x | skip] for the up block: the skip half is written here right away
        ops.axpby(x.view(rows * T, 256), cat.view(rows * T, 512)[:, 256:], 1.0, 0.0)
        y = ws["y"]
        ops.conv1d(x, st[0]["tail"][0], y, taps=3, cin=256, bias=st[0]["tail"][1], pad_left=2)
        cur = y
        for k in range(1, 1 + self.n_mid):
            x = self._resnet(st[k], cur, 256, rows, T, tbias[k], ws, ws["x"] if cur is not ws["x"] else ws["x2"])
            for tw in st[k]["tb"]:
                self._tblock(tw, x, rows, T, lens, ws)
            cur = x
        ops.axpby(cur.view(rows * T, 256), cat.view(rows * T, 512)[:, :256], 1.0, 0.0)
        up = st[-1]
        x = self._resnet(up, cat, 512, rows, T, tbias[-1], ws, ws["x"])
        for tw in up["tb"]:
            self._tblock(tw, x, rows, T, lens, ws)
        ops.conv1d(x, up["tail"][0], y, taps=3, cin=256, bias=up["tail"][1], pad_left=2)
        a = ws["ra"]
        ops.conv1d(y, self.fin["c"][0], a, taps=3, cin=256, bias=self.fin["c"][1], pad_left=2)
        a2 = a.view(rows * T, 256)
        ops.layernorm(a2, self.fin["n"][0], self.fin["n"][1], a2, 1e-5, act=ops.MISH)
        ops.conv1d(a, self.fin["proj"][0], ws["v"], taps=1, cin=256, bias=self.fin["proj"][1])
        return ws["v"]

    # ------------------------------------------------------------------ CFM estimator on plane-format operands
    def _plane_weights(self):
        if self._pw is None:
            sp = ops.split_planes

            def stage(s):
                d = dict(c1=sp(s["c1"][0]), c2=sp(s["c2"][0]), res=sp(s["res"][0]), tb=[])
                for t in s["tb"]:  # q | k rows of the fused projection; the v rows are the A operand of the swapped (V^T) product
                    wqkv = sp(t["wqkv"])  # one image; q | k rows and v rows are row ranges of it
                    d["tb"].append(dict(wqkv=wqkv, wqk=wqkv.rows_view(0, 1024), wv=wqkv.rows_view(1024, 512), wo=sp(t["wo"]), w1=sp(t["w1"]), w2=sp(t["w2"])))
                if "tail" in s:
                    d["tail"] = sp(s["tail"][0])
                return d

            self._pw = dict(stages=[stage(s) for s in self.stages], fin_c=sp(self.fin["c"][0]), fin_proj=sp(self.fin["proj"][0]))
        return self._pw

    def _resnet_pl(self, sw, pw, inP, cin, rows, T, tb, ws):
        """CausalResnetBlock1D on plane operands: inP (Planes, rows*T x cin) -> ws['x'] fp32."""
        M = rows * T
        a, b, x = ws["ra"], ws["rb"], ws["x"]
        cv = lambda src, w, c, **k: ops.conv1d_planes(src, w, B=rows, T=T, cin=c, **k)
        cv(inP, pw["c1"], cin, taps=3, out=a, bias=sw["c1"][1], pad_left=2)
        ops.layernorm_planes(a.view(M, 256), sw["n1"][0], sw["n1"][1], ws["aP"], 1e-5, act=ops.MISH, post_add=tb)
        cv(ws["aP"], pw["c2"], 256, taps=3, out=b, bias=sw["c2"][1], pad_left=2)
        b2 = b.view(M, 256)
        ops.layernorm(b2, sw["n2"][0], sw["n2"][1], b2, 1e-5, act=ops.MISH)
        cv(inP, pw["res"], cin, taps=1, out=x, bias=sw["res"][1], residual=b)
        return x

    def _tblock_pl(self, tw, pw, x, rows, T, lens, ws, outP=None, pre_normed=False, next_ln=None):
        """BasicTransformerBlock on plane operands.  x (rows,T,256) fp32 residual stream, updated in place -- except by the LAST block of a
        stage (outP given): its result only feeds convolutions, so it is written in plane format alone.
        fused_ln (round 5, cbx_gemm_pl_t.ln_w): norm3 is produced by the out-projection's epilogue and the NEXT block's norm1 (`next_ln`) by ff2's -- the
        row-spanning 64 x 256 tile finishes whole rows -- so a block launches no LayerNorm of its own (`pre_normed`: hP already holds norm1(x))."""
        M = rows * T
        x2, hP, qkP, vtP, attP, ffP = x.view(M, 256), ws["hP"], ws["qkP"], ws["vtP"], ws["attP"], ws["ffP"]
        if not pre_normed:
            ops.layernorm_planes(x2, tw["n1"][0], tw["n1"][1], hP, 1e-5)
        if self.fused_qkv and T % 4 == 0:
            # to_q | to_k | to_v as ONE Linear: the q | k columns go to qkP, the v columns are stored transposed (V^T[z] = 512 x T per row group of T)
            ops.gemm_planes(hP, pw["wqkv"], M=M, N=1536, K=256, P=qkP, PT=vtP, pt_n0=1024, pt_T=T, pt_zs=512 * vtP.ld)
        else:
            ops.linear_planes(hP, pw["wqk"], outp=qkP)
            # V^T[z] (512 x T) = W_v h[z]^T: the same products with the operands swapped, so the store is the transposed tile
            ops.gemm_planes(pw["wv"], hP, M=512, N=T, K=256, nz1=rows, w_s1=T * hP.ld, P=vtP, p_s1=512 * vtP.ld)
        ops.flash_attn_planes(qkP.cols(0, 512), qkP.cols(512, 512), vtP, attP, Z=rows, H=8, T=T, vt_sb=512 * vtP.ld, scale=0.125, key_lens=lens)
        if self.fused_ln:
            ops.linear_planes(attP, pw["wo"], out=x2, bias=tw["bo"], residual=x2, ln=tw["n3"], lnp=hP)
        else:
            ops.linear_planes(attP, pw["wo"], out=x2, bias=tw["bo"], residual=x2)
            ops.layernorm_planes(x2, tw["n3"][0], tw["n3"][1], hP, 1e-5)
        ops.linear_planes(hP, pw["w1"], outp=ffP, bias=tw["b1"], act=ops.GELU_ERF)
        ops.linear_planes(ffP, pw["w2"], out=x2 if outP is None else None, outp=outP, bias=tw["b2"], residual=x2, ln=next_ln, lnp=hP if next_ln is not None else None)

    def _estimator_pl(self, xinP, rows, T, lens, tbias, ws):
        """ConditionalDecoder.forward (decoder.py:243-333) on plane operands: xinP Planes (rows*T, 320) -> ws['v'] (rows,T,80) fp32."""
        st, pws = self.stages, self._plane_weights()["stages"]
        catP, xP, yP = ws["catP"], ws["xP"], ws["yP"]
        cv = lambda src, w, **k: ops.conv1d_planes(src, w, B=rows, T=T, cin=256, **k)

        def block(k, inP, cin, outP):
            x = self._resnet_pl(st[k], pws[k], inP, cin, rows, T, tbias[k], ws)
            tbs = st[k]["tb"]
            for j, (tw, pw) in enumerate(zip(tbs, pws[k]["tb"])):
                last = j == len(tbs) - 1
                self._tblock_pl(tw, pw, x, rows, T, lens, ws, outP if last else None, pre_normed=self.fused_ln >= 2 and j > 0,
                                next_ln=tbs[j + 1]["n1"] if (self.fused_ln >= 2 and not last) else None)

        skip, xh = catP.cols(256, 256), catP.cols(0, 256)  # [x | skip] of the up block: both halves are written in place by their producers
        block(0, xinP, 320, skip)
        cur = xh if self.n_mid == 0 else yP
        cv(skip, pws[0]["tail"], taps=3, outp=cur, bias=st[0]["tail"][1], pad_left=2)
        for k in range(1, 1 + self.n_mid):
            nxt = xh if k == self.n_mid else xP
            block(k, cur, 256, nxt)
            cur = nxt
        block(len(st) - 1, catP, 512, xP)
        cv(xP, pws[-1]["tail"], taps=3, outp=yP, bias=st[-1]["tail"][1], pad_left=2)
        pw = self._plane_weights()
        a = ws["ra"]
        cv(yP, pw["fin_c"], taps=3, out=a, bias=self.fin["c"][1], pad_left=2)
        ops.layernorm_planes(a.view(rows * T, 256), self.fin["n"][0], self.fin["n"][1], ws["aP"], 1e-5, act=ops.MISH)
        cv(ws["aP"], pw["fin_proj"], taps=1, out=ws["v"], bias=self.fin["proj"][1])
        return ws["v"]

    def _cfm_descriptor(self):
        """The constant part of cbx_cfm_t (weights of every stage as cbx_cfm_stage_t / cbx_cfm_tblock_t host arrays), built once per engine."""
        if self._cfm_static is None:
            from ._lib import CfmSolve, CfmStage, CfmTBlock, PlanesRef
            ref = lambda P: PlanesRef(P.ptr, P.ld, P.lo)
            p = ops._p
            pw = self._plane_weights()
            keep, stages = [], (CfmStage * len(self.stages))()
            for k, (sw, sp) in enumerate(zip(self.stages, pw["stages"])):
                tbs = (CfmTBlock * len(sw["tb"]))()
                for j, (tw, tp) in enumerate(zip(sw["tb"], sp["tb"])):
                    t = tbs[j]
                    t.n1_w, t.n1_b, t.n3_w, t.n3_b, t.bo, t.b1, t.b2 = (p(tw["n1"][0]), p(tw["n1"][1]), p(tw["n3"][0]), p(tw["n3"][1]), p(tw["bo"]),
                                                                        p(tw["b1"]), p(tw["b2"]))
                    t.wqkv, t.wo, t.w1, t.w2 = ref(tp["wqkv"]), ref(tp["wo"]), ref(tp["w1"]), ref(tp["w2"])
                st = stages[k]
                st.c1, st.c2, st.res = ref(sp["c1"]), ref(sp["c2"]), ref(sp["res"])
                st.c1_b, st.n1_w, st.n1_b, st.c2_b = p(sw["c1"][1]), p(sw["n1"][0]), p(sw["n1"][1]), p(sw["c2"][1])
                st.n2_w, st.n2_b, st.res_b = p(sw["n2"][0]), p(sw["n2"][1]), p(sw["res"][1])
                if "tail" in sw:
                    st.tail, st.tail_b = ref(sp["tail"]), p(sw["tail"][1])
                st.cin, st.n_tb, st.tb = int(sw["cin"]), len(sw["tb"]), tbs
                keep.append(tbs)
            d = CfmSolve()
            d.n_stages, d.stages = len(self.stages), stages
            d.fin_c, d.fin_proj = ref(pw["fin_c"]), ref(pw["fin_proj"])
            d.fin_c_b, d.fin_n_w, d.fin_n_b, d.fin_proj_b = p(self.fin["c"][1]), p(self.fin["n"][0]), p(self.fin["n"][1]), p(self.fin["proj"][1])
            self._cfm_static = (d, stages, keep)
        return self._cfm_static[0]

    def _cfm_solve_c(self, xin, xinP, lens_r, tb, t_span, ws, B, rows, T, n_steps, cfg, cfg_rate):
        """solve_euler through cbx_cfm_solve: one ctypes call instead of ~4400 per utterance batch."""
        import ctypes

        from ._lib import CfmSolve, PlanesRef, check, lib
        d = CfmSolve()
        ctypes.memmove(ctypes.byref(d), ctypes.byref(self._cfm_descriptor()), ctypes.sizeof(CfmSolve))
        ref = lambda P: PlanesRef(P.ptr, P.ld, P.lo)
        dt = (ctypes.c_float * n_steps)(*[float(t_span[k + 1] - t_span[k]) for k in range(n_steps)])
        assert tb.is_contiguous() and tb.shape == (n_steps, len(self.stages), 256) and lens_r.dtype == torch.int32 and lens_r.numel() == rows
        d.rows, d.B, d.n_steps, d.cfg, d.fused_qkv, d.fused_ln, d.T = rows, B, n_steps, int(cfg), int(self.fused_qkv), int(self.fused_ln), T
        d.cfg_rate, d.dt = cfg_rate, dt
        d.gemm_tile, d.attn_version = int(self.gemm_tile), int(self.attn_version)
        d.tbias, d.lens, d.xin, d.xinP = ops._p(tb), ops._p(lens_r), ops._p(xin), ref(xinP)
        d.ra, d.rb, d.x, d.v = ops._p(ws["ra"]), ops._p(ws["rb"]), ops._p(ws["x"]), ops._p(ws["v"])
        for k in ("aP", "hP", "qkP", "attP", "ffP", "xP", "yP", "catP", "vtP"):
            setattr(d, k, ref(ws[k]))
        check(lib.cbx_cfm_solve(ctypes.byref(d), ops._stream()), "cbx_cfm_solve")

    def _planes_ok(self, rows, T):
        """The plane-format path serves the f16x3 numerics (precision 16) with 31-bit operand offsets; anything else runs the fp32-operand kernels."""
        # (cbx_gemm_planes addresses one batch of an operand / output with 32-bit byte offsets: the widest plane tensor, the feed-forward
        # intermediate, has 4096-byte rows and is passed as ONE batch of rows * T rows)
        return self.use_planes and ops.GEMM_PRECISION == 16 and rows * T > 32 and T % 2 == 0 and (rows * T + 512) * 4096 < 2 ** 31

    def _time_bias(self, t_vals, r_vals=None):
        """SinusoidalPosEmb -> TimestepEmbedding (-> meanflow mixer) -> every ResNet's Mish+Linear (matcha/decoder.py:20-29,
        105-117; decoder.py:264-268): returns (n_steps, n_resnets, 256)."""
        dev = self.dev

        def tmlp(tv):
            half = 160
            fr = torch.exp(torch.arange(half).float() * -(math.log(10000) / (half - 1)))
            e = 1000.0 * tv[:, None] * fr[None]
            e = torch.cat([e.sin(), e.cos()], -1).to(dev)
            n = e.shape[0]
            h1, h2 = torch.empty(n, 1024, device=dev), torch.empty(n, 1024, device=dev)
            ops.linear(e, self.t1[0], h1, bias=self.t1[1], act=ops.SILU)
            ops.linear(h1, self.t2[0], h2, bias=self.t2[1])
            return h2

        temb = tmlp(t_vals)
        if r_vals is not None:
            cat = torch.cat([temb, tmlp(r_vals)], 1).contiguous()
            temb = torch.empty_like(temb)
            ops.linear(cat, self.t_mix, temb)
        n = temb.shape[0]
        m = torch.empty(n, 1024, device=dev)
        ops.act(temb, m, ops.MISH)
        out = torch.empty(n, self.tmlp_w.shape[0], device=dev)
        ops.linear(m, self.tmlp_w, out, bias=self.tmlp_b)
        return out.view(n, -1, 256)

    @ops.on_device
    def cfm(self, mu, lens, spk, cond, z, n_steps=10, cfg_rate=0.7):
        """CausalConditionalCFM.forward + solve_euler (flow_matching.py:78-145,196-233).
        mu/cond/z (B,T,80) channel-last, lens (B,) int32 valid frames, spk (B,80).  Returns x (B,T,80)."""
        dev, (B, T, _) = self.dev, mu.shape
        cfg = not self.meanflow
        rows = 2 * B if cfg else B
        f = lambda *s: torch.empty(*s, device=dev)
        planes = self._planes_ok(rows, T)
        if planes:
            M, Tp = rows * T, (T + 7) // 8 * 8
            PL = lambda C, **k: ops.Planes(M, C, dev, **k)
            ws = dict(ra=f(rows, T, 256), rb=f(rows, T, 256), x=f(rows, T, 256), v=f(rows, T, 80), aP=PL(256), hP=PL(256), qkP=PL(1024),
                      attP=PL(512), ffP=PL(1024), xP=PL(256), yP=PL(256), catP=PL(512),
                      vtP=ops.Planes(rows * 512, Tp, dev, zero=True))  # V^T rows are padded to 8 keys; the pad stays zero
        else:
            ws = dict(ra=f(rows, T, 256), rb=f(rows, T, 256), x=f(rows, T, 256), x2=f(rows, T, 256), y=f(rows, T, 256),
                      cat=f(rows, T, 512),
                      h=f(rows * T, 256), qkv=f(rows * T, 1536), att=f(rows * T, 512), ff=f(rows * T, 1024), v=f(rows, T, 80),
                      stats=f(rows * T, 2))
        xin = torch.zeros(rows, T, 320, device=dev)
        xin[:B, :, 0:80] = z
        xin[:B, :, 80:160] = mu
        xin[:B, :, 160:240] = spk[:, None, :]
        xin[:B, :, 240:320] = cond
        if cfg:
            xin[B:, :, 0:80] = z
        lens_r = torch.cat([lens, lens]).contiguous() if cfg else lens
        t_span = torch.linspace(0, 1, n_steps + 1)
        if not self.meanflow:
            t_span = 1 - torch.cos(t_span * 0.5 * math.pi)
        if n_steps not in self._tb_cache:  # the schedule is a function of n_steps only: every ResNet's time bias is a load-time constant
            self._tb_cache[n_steps] = self._time_bias(t_span[:-1], t_span[1:] if self.meanflow else None)
        tb = self._tb_cache[n_steps]
        if planes:  # the packed estimator input in plane format: [mu | spk | cond] are split once, x after every Euler step
            xin2 = xin.view(rows * T, 320)
            xinP = ops.split_planes(xin2)
        if planes and self.c_seam and not ops.TIMER:
            self._cfm_solve_c(xin, xinP, lens_r, tb, t_span, ws, B, rows, T, n_steps, cfg, cfg_rate)
            return xin[:B, :, :80]
        for k in range(n_steps):
            if planes:
                if k:
                    ops.split_planes(xin2[:, :80], xinP.cols(0, 80))
                with ops.planes_geometry(self.gemm_tile, self.attn_version):
                    v = self._estimator_pl(xinP, rows, T, lens_r, tb[k], ws)
            else:
                v = self._estimator(xin, rows, T, lens_r, tb[k], ws)
            ops.cfm_euler(xin, v, B, T, 80, float(t_span[k + 1] - t_span[k]), cfg_rate, cfg=cfg)
        return xin[:B, :, :80]

    # ------------------------------------------------------------------ flow.inference
    @ops.on_device
    @torch.inference_mode()
    def inference(self, tokens, token_lens, ref, z=None, n_steps=10, hold_back=None):
        with ops.gemm_precision(self.precision):
            return self._inference(tokens, token_lens, ref, z, n_steps, hold_back)

    def _inference(self, tokens, token_lens, ref, z=None, n_steps=10, hold_back=None):
        """tokens (B,N) int64 (right-padded), token_lens (B,), ref dict as produced by S3Gen.embed_ref -- or a LIST of B such dicts, one voice
        per utterance (the reference takes a ref_dict per call, s3gen.py:173-229; a device batch may mix voices: prompt tokens / prompt mels
        of different lengths are left-aligned per row) --, z optional injected noise (B, 2(P+N)max, 80) channel-last.  Returns mel
        (B, frames, 80) channel-last, row b valid for 2 N_b - (prompt_feat frames_b - 2 P_b) frames: (B, 2N, 80) for whole-token prompts.
        hold_back (B,) ints: chunked synthesis -- the last hold_back[b] frames of utterance b are not generated (`finalize=False` of
        flow.py:170-171, whose reference branch raises; semantics restated)."""
        dev = self.dev
        B, N = tokens.shape
        refs = list(ref) if isinstance(ref, (list, tuple)) else [ref] * B
        assert len(refs) == B, f"{len(refs)} voices for {B} utterances"
        one = all(r is refs[0] for r in refs)
        ptoks = [r["prompt_token"].to(dev).long().view(-1) for r in (refs[:1] if one else refs)]
        Ps = [int(t.numel()) for t in ptoks] * (B if one else 1)
        Pmax = max(Ps)
        if one:
            tok = torch.cat([ptoks[0].view(1, -1).expand(B, -1), tokens.to(dev).long()], 1).contiguous()
        else:  # row b = [prompt_b | tokens_b | padding]: the encoder and the CFM mask by length, so what the padding holds is irrelevant
            nb = [int(v) for v in token_lens.tolist()]
            L = max(p + n for p, n in zip(Ps, nb))  # longest row: the batch is 2 L mel frames long
            tok = torch.zeros(B, L, dtype=torch.long, device=dev)
            for b in range(B):
                tok[b, : Ps[b]] = ptoks[b]
                tok[b, Ps[b]: Ps[b] + nb[b]] = tokens[b, : nb[b]].to(dev).long()
        lens = (token_lens.to(dev).to(torch.int32) + torch.tensor(Ps, dtype=torch.int32, device=dev)).contiguous()
        mu = self.encode(tok, lens)
        T = mu.shape[1]
        xv = torch.stack([r["embedding"].to(dev).float().view(-1) for r in (refs[:1] if one else refs)]).contiguous()
        emb = torch.empty_like(xv)  # F.normalize(embedding, dim=1) (flow.py:150): x / ||x|| = rmsnorm(x) / sqrt(C)
        # eps = 1e-24 / C under the root = F.normalize's max(||x||, 1e-12): a zero embedding gives 0, not NaN
        ops.layernorm(xv, torch.ones(xv.shape[1], device=dev), None, emb, 1e-24 / xv.shape[1], rms=True, scale=1.0 / math.sqrt(xv.shape[1]))
        spk = torch.empty(xv.shape[0], 80, device=dev)
        ops.linear(emb, self.spk_w, spk, bias=self.spk_b)
        # mel_len1 = prompt_feat.shape[1] (flow.py:170-175): normally 2P; one frame more when the prompt is not a whole number of
        # 40 ms tokens (embed_ref trims the tokens, not the mel, s3gen.py:152-158) -- the output then has 2N - (mel_len1 - 2P) frames
        pfs = [r["prompt_feat"].to(dev).float().view(-1, 80) for r in (refs[:1] if one else refs)]
        Pms = [int(t.shape[0]) for t in pfs] * (B if one else 1)
        cond = torch.zeros(B, T, 80, device=dev)
        if one:
            cond[:, : Pms[0]] = pfs[0]
        else:
            for b in range(B):
                cond[b, : Pms[b]] = pfs[b]
        if z is None:
            z = torch.randn(B, T, 80, device=dev)
        mel_lens = (2 * lens).to(torch.int32)
        if hold_back is not None:  # chunked synthesis: the encoder's 3-token lookahead frames are masked out of the CFM like padding
            mel_lens = (mel_lens - torch.as_tensor(hold_back, dtype=torch.int32).to(dev)).contiguous()
        x = self.cfm(mu, mel_lens, spk.expand(B, -1) if one else spk, cond, z.to(dev), n_steps)
        if one or all(pm == Pms[0] and p == Ps[0] for pm, p in zip(Pms, Ps)):
            return x[:, Pms[0]:, :].contiguous()
        Ws = [2 * (Ps[b] + nb[b]) - Pms[b] for b in range(B)]  # generated frames per row: 2 n_b - (prompt_feat frames_b - 2 P_b)
        out = torch.zeros(B, max(Ws), 80, device=dev)            # left-aligned per row
        for b in range(B):
            out[b, : Ws[b]] = x[b, Pms[b]: Pms[b] + Ws[b]]
        return out
