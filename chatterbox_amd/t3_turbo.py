"""Turbo / Nano T3 (GPT-2 backbone) on MI355X: host-side mirror of `T3.inference_turbo` (reference models/t3/t3.py:392-468,
configuration tts_turbo.py:153-167).  Same kernel set as the Llama path with other epilogues: LayerNorm(+bias) instead of
RMSNorm, fused c_attn with bias (HF Conv1D weights are stored [in,out] and transposed once at load), no RoPE (learned wpe is
added when the token is embedded), gelu_new in the c_fc epilogue, speech head with bias, no CFG (rows = utterances), sampler
in the Temperature -> TopK -> TopP -> RepetitionPenalty order.  One decode step is a hipGraph: 5 launches per layer for <= 16 rows
(LayerNorm folded into the consuming GEMVs, _forward_decode_v2), 7 beyond.
"""
import os

import torch

from . import ops
from .t3 import VoicePrefixCache

START_SPEECH, STOP_SPEECH = 6561, 6562


class T3TurboEngine(VoicePrefixCache):
    MAX_BATCH = 64
    @ops.on_device
    def __init__(self, sd, device="cuda", n_layers=None):
        self.dev = dev = torch.device(device)
        if n_layers is None:
            n_layers = 0
            while f"tfmr.h.{n_layers}.ln_1.weight" in sd:
                n_layers += 1
        self.L = n_layers
        d = lambda t: t.float().contiguous().to(dev)
        tw = lambda t: t.float().t().contiguous().to(dev)  # Conv1D [in,out] -> [out,in]
        self.D = sd["tfmr.wpe.weight"].shape[1]
        self.H = self.D // 64
        self.layers = []
        for i in range(n_layers):
            p = f"tfmr.h.{i}."
            self.layers.append(dict(
                ln1=(d(sd[p + "ln_1.weight"]), d(sd[p + "ln_1.bias"])), ln2=(d(sd[p + "ln_2.weight"]), d(sd[p + "ln_2.bias"])),
                wqkv=tw(sd[p + "attn.c_attn.weight"]), bqkv=d(sd[p + "attn.c_attn.bias"]),
                wo=tw(sd[p + "attn.c_proj.weight"]), bo=d(sd[p + "attn.c_proj.bias"]),
                wfc=tw(sd[p + "mlp.c_fc.weight"]), bfc=d(sd[p + "mlp.c_fc.bias"]),
                wpr=tw(sd[p + "mlp.c_proj.weight"]), bpr=d(sd[p + "mlp.c_proj.bias"])))
        # decode path: lane-ordered packed images of the streamed weights (every wave-level load = 1 KiB contiguous, cbx.h)
        # decode tuning (same knobs as T3Engine._TUNE; CBX_TURBO_TUNE="d_ks=4,d_nw=8,o_nw=8,half_tiles=0" overrides for an A/B):
        # 8-column tiles for the two N = D projections (twice the workgroups), split-K factor / waves of the MLP projection;
        # qkv_tc = 12 / od_tc = 4 (ABI v9): c_attn resp. the two N = D projections on N / 12 resp. N / 4 workgroups, d_ks = 1: the MLP
        # projection adds bias + residual itself (no partial images, no fold in the next c_attn GEMV)
        # head_ct: column tiles per workgroup of the head GEMV (cbx_gemv_t.col_tiles: 6563 columns = 411 tiles -> 206 workgroups, one round of the chip)
        # row_path (round 6, ABI v14): batch 1 runs on the single-row streaming kernels (ops.gemv_row / ops.decode_attn_parts: row-major weights, no
        # MFMA padding, no LDS reduction, no partial images; _forward_decode_row); row_splits / row_chunks: context slices per (row, head) and
        # 16-position chunks in flight per workgroup of its attention
        self.tune = dict(d_ks=2, d_nw=16, o_nw=8, half_tiles=1, qkv_tc=0, od_tc=0, head_ct=2, row_path=1, row_splits=16, row_chunks=2)
        for kv in filter(None, os.environ.get("CBX_TURBO_TUNE", "").split(",")):
            k, v = kv.split("=")
            assert k.strip() in self.tune, f"CBX_TURBO_TUNE: unknown knob {k!r} (known: {sorted(self.tune)})"
            self.tune[k.strip()] = int(v)
        # launch knobs of the decode attention / GEMVs, per ENGINE (they travel in every call's descriptor: cbx_decode_attn_t, cbx_gemv_t.flags)
        from .autotune import env_knobs
        self.knobs = env_knobs()
        for lw in self.layers:
            for k in ("wqkv", "wo", "wfc", "wpr"):
                lw[k + "_pk"] = ops.pack_gemv_weight(lw[k])
            if self.tune["half_tiles"]:
                lw["wo_pk8"] = ops.pack_gemv_weight(lw["wo"], half_tile=True)
                lw["wpr_pk8"] = ops.pack_gemv_weight(lw["wpr"], half_tile=True)
        self.lnf = (d(sd["tfmr.ln_f.weight"]), d(sd["tfmr.ln_f.bias"]))
        self.decode_mode = os.environ.get("CBX_T3_DECODE", "v2")
        self.wpe = d(sd["tfmr.wpe.weight"])
        self.text_emb, self.speech_emb = d(sd["text_emb.weight"]), d(sd["speech_emb.weight"])
        self.head, self.head_b = d(sd["speech_head.weight"]), d(sd["speech_head.bias"])
        self.V = self.head.shape[0]
        self.head_pk = ops.pack_gemv_weight(self.head)
        # LayerNorm folded into the consuming GEMV (cbx_gemv_t.ln_cw / ln_cb): out = rstd (sum_k x w W - mean cw) + cb with the layer
        # constants cw[n] = sum_k w[k] W[n][k], cb[n] = sum_k b[k] W[n][k] + bias[n]
        def ln_consts(ln, W, bias):
            cw, cb = torch.empty(1, W.shape[0], device=dev), torch.empty(1, W.shape[0], device=dev)
            ops.gemv(ln[0].view(1, -1), W, cw, nw=4)
            ops.gemv(ln[1].view(1, -1), W, cb, bias=bias, nw=4)
            return cw.view(-1), cb.view(-1)
        for lw in self.layers:
            lw["c_qkv"] = ln_consts(lw["ln1"], lw["wqkv"], lw["bqkv"])
            lw["c_fc"] = ln_consts(lw["ln2"], lw["wfc"], lw["bfc"])
        self.c_head = ln_consts(self.lnf, self.head, self.head_b)
        self.spkr_w, self.spkr_b = d(sd["cond_enc.spkr_enc.weight"]), d(sd["cond_enc.spkr_enc.bias"])
        # split-K factors of the two down-projections: K must be a multiple of 32 * ksplit * 4
        self.ks_o = 4 if self.D % 512 == 0 else 2
        self.ks_p = 8
        self._state = {}
        self.share_prefix, self._prefix_cache = os.environ.get("CBX_T3_SHARE_PREFIX", "1") == "1", []  # VoicePrefixCache: speaker + prompt tokens of a voice prefilled once
        self.c_prefill = os.environ.get("CBX_T3_CSTEP", "1") == "1"  # the prefill through cbx_gpt2_prefill (one ctypes call instead of nine launches per layer)
        self.time_decode, self.decode_events = False, []  # (start, end, steps, prefill lengths, rows) per generate() when enabled (bench.py)

    def _forward_decode(self, st):
        ws = st["dws"]
        x, h, qkv, att, g, po, pd = ws["x"], ws["h"], ws["qkv"], ws["att"], ws["g"], ws["po"], ws["pd"]
        ops.embed(st["next_ids"], self.speech_emb, x, table2=self.wpe, ids2=st["positions"])
        part = None
        for i, lw in enumerate(self.layers):
            ops.add_rmsnorm(x, part, lw["ln1"][0], h, bias=lw["ln1"][1], rms=False)
            ops.gemv(h, lw["wqkv_pk"], qkv, N=3 * self.D, bias=lw["bqkv"], nw=8, w_packed=True)
            ops.decode_attn_rope(qkv, st["positions"], None, None, st["kc"][i], st["vc"][i], att, 0.125, geom=st["da"])
            ops.gemv(att, lw["wo_pk"], po, N=self.D, bias=lw["bo"], ksplit=self.ks_o, nw=4, w_packed=True)
            ops.add_rmsnorm(x, po, lw["ln2"][0], h, bias=lw["ln2"][1], rms=False)
            ops.gemv(h, lw["wfc_pk"], g, N=4 * self.D, bias=lw["bfc"], nw=8, act=ops.GELU_TANH, w_packed=True)
            ops.gemv(g, lw["wpr_pk"], pd, N=self.D, bias=lw["bpr"], ksplit=self.ks_p, nw=4, w_packed=True)
            part = pd
        ops.add_rmsnorm(x, part, self.lnf[0], h, bias=self.lnf[1], rms=False)
        ops.gemv(h, self.head_pk, st["logits"], N=self.V, bias=self.head_b, nw=4, w_packed=True)

    def _forward_decode_v2(self, st):
        """5 launches per GPT-2 layer (rows <= 16): LayerNorm folded into the c_attn / c_fc / head GEMVs, the attention projection adds
        bias + residual in its epilogue, the MLP projection emits split-K partial images that the next consumer sums into its operand
        (same structure as T3Engine._forward_decode_v2)."""
        ws, D = st["dws"], self.D
        B, tn = st["B"], self.tune
        dks = tn["d_ks"]
        qtc, odtc = self._tiles()
        qt, ot = (0 if qtc == 16 else qtc), (0 if odtc == 16 else odtc)
        cur, nxt, qkv, att, g, pd = ws["x_pk"], ws["x2_pk"], ws["qkv"], ws["att_pk"], ws["g_pk"], ws["pd_pk"][:dks]
        pk = dict(w_packed=True, x_packed=True, M=B, flags=ops.gemv_flags(self.knobs.get("pre_epi"), self.knobs.get("deep")))
        ops.embed(st["next_ids"], self.speech_emb, cur, table2=self.wpe, ids2=st["positions"], out_packed=True)
        red = {}
        for i, lw in enumerate(self.layers):
            ops.gemv(cur, self._image(lw, "wqkv", qtc), qkv, N=3 * D, K=D, nw=8, norm_w=lw["ln1"][0], ln_cw=lw["c_qkv"][0], ln_cb=lw["c_qkv"][1],
                     half_tile=qt, **red, **pk)
            if red:
                cur, nxt = nxt, cur
            ops.decode_attn_rope(qkv, st["positions"], None, None, st["kc"][i], st["vc"][i], att, 0.125, out_packed=True, geom=st["da"])
            ops.gemv(att, self._image(lw, "wo", odtc), cur, N=D, K=D, nw=tn["o_nw"], bias=lw["bo"], res=cur, out_packed=True, half_tile=ot, **pk)
            ops.gemv(cur, lw["wfc_pk"], g, N=4 * D, K=D, nw=8, norm_w=lw["ln2"][0], ln_cw=lw["c_fc"][0], ln_cb=lw["c_fc"][1],
                     act=ops.GELU_TANH, out_packed=True, **pk)
            if dks > 1:
                ops.gemv(g, self._image(lw, "wpr", odtc), pd, N=D, K=4 * D, ksplit=dks, nw=tn["d_nw"], bias=lw["bpr"], out_packed=True, half_tile=ot, **pk)
                red = dict(xpart=pd, x_out=nxt)
            else:
                ops.gemv(g, self._image(lw, "wpr", odtc), cur, N=D, K=4 * D, nw=tn["d_nw"], bias=lw["bpr"], res=cur, out_packed=True, half_tile=ot, **pk)
        if red:
            red["x_out"] = None
        ops.gemv(cur, self.head_pk, st["logits"], N=self.V, K=D, nw=8, norm_w=self.lnf[0], ln_cw=self.c_head[0], ln_cb=self.c_head[1],
                 col_tiles=int(tn.get("head_ct") or 0) if D % 256 == 0 else 0, **red, **pk)

    def _forward_decode_row(self, st):
        """Batches of 1 .. 2 utterances (same-box A/B, profiles/r06_k_few_row_path_small_batches_ab.log: 0.594 against 0.859 ms / token at B = 1, 0.780 against
        0.893 at B = 2, 1.077 against 0.909 at B = 4 -- the VALU contraction costs M times the single row's, the MFMA tile does not): 5 launches per GPT-2 layer on the few-row kernels (include/cbx.h "few-row decode").  LayerNorm lives in the
        prologue of the c_attn / c_fc / head GEMVs, the attention leaves `row_splits` partial records per (row, head) that the c_proj GEMV merges in ITS
        prologue, both projections add bias + residual in place.  Weights are the row-major matrices the prefill uses (no packed images)."""
        ws, tn = st["dws"], self.tune
        x, qkv, g, parts = ws["x"], ws["qkv"], ws["g"], ws["parts"]
        ops.embed(st["next_ids"], self.speech_emb, x, table2=self.wpe, ids2=st["positions"])
        for i, lw in enumerate(self.layers):
            ops.gemv_row(x, lw["wqkv"], qkv, bias=lw["bqkv"], ln=lw["ln1"])
            ops.decode_attn_parts(qkv, st["positions"], st["kc"][i], st["vc"][i], parts, 0.125, chunks=tn["row_chunks"])
            ops.gemv_row(None, lw["wo"], x, bias=lw["bo"], res=x, parts=parts)
            ops.gemv_row(x, lw["wfc"], g, bias=lw["bfc"], ln=lw["ln2"], act=ops.GELU_TANH)
            ops.gemv_row(g, lw["wpr"], x, bias=lw["bpr"], res=x)
        ops.gemv_row(x, self.head, st["logits"], bias=self.head_b, ln=self.lnf)

    def _tiles(self):
        """(c_attn tile width, attention / MLP projection tile width) of the current tune: 16, 12, 8 or 4 output columns per workgroup."""
        tn = self.tune
        return tn.get("qkv_tc") or 16, tn.get("od_tc") or (8 if tn.get("half_tiles") else 16)

    def _image(self, lw, name, tc):
        """Packed decode image of layer weight `name` for `tc`-column tiles (packed on first use; generate() calls _prepare_tune() before
        anything is captured)."""
        key = f"{name}_pk" if tc == 16 else f"{name}_pk{tc}"
        if key not in lw:
            lw[key] = ops.pack_gemv_weight(lw[name], half_tile=tc)
        return lw[key]

    def _prepare_tune(self):
        qtc, odtc = self._tiles()
        assert qtc in (16, 12) and odtc in (16, 8, 4) and self.tune["d_ks"] in (1, 2, 4), f"decode tune {self.tune}"
        for lw in self.layers:
            self._image(lw, "wqkv", qtc), self._image(lw, "wo", odtc), self._image(lw, "wpr", odtc)

    def _sample(self, st):
        ops.t3_sample(logits=st["logits"], ld=st["logits"].stride(0), V=self.V, B=st["B"], cfg=0, order=1, eos_token=STOP_SPEECH,
                      dev_params=st["samp_dev"], seen=st["seen"], uniforms=st["uniforms"], max_steps=st["max_steps"], step=st["step"],
                      out_tokens=st["out_tokens"], done=st["done"], n_generated=st["n_generated"], next_ids=st["next_ids"],
                      next_pos_ids=st["next_pos_ids"], positions=st["positions"], ctx_lens=st["ctx_lens"])

    def _forward(self, st):
        if self.decode_mode == "v2" and st["B"] <= 2 and self.tune.get("row_path") and self.D % 256 == 0:
            return self._forward_decode_row(st)
        if self.decode_mode == "v2" and st["B"] <= 16:
            return self._forward_decode_v2(st)
        self._forward_decode(st)

    def _decode_step(self, st):
        self._forward(st)
        self._sample(st)

    def _get_state(self, B, max_ctx, max_steps):
        key = (B, max_ctx, max_steps)
        if key in self._state:
            return self._state[key]
        self._state.clear()
        dev, D = self.dev, self.D
        f = lambda *s: torch.empty(*s, device=dev)
        i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=dev)
        st = dict(B=B, max_ctx=max_ctx, max_steps=max_steps,
                  kc=torch.zeros(self.L, B, self.H, max_ctx, 64, device=dev), vc=torch.zeros(self.L, B, self.H, max_ctx, 64, device=dev),
                  logits=f(B, self.V), seen=torch.zeros(B, self.V, dtype=torch.uint8, device=dev), uniforms=f(B, max_steps), step=i32(B),
                  out_tokens=torch.zeros(B, max_steps, dtype=torch.int64, device=dev), done=i32(B), n_generated=i32(B),
                  next_ids=torch.zeros(B, dtype=torch.int64, device=dev), next_pos_ids=i32(B), positions=i32(B), ctx_lens=i32(B),
                  dws=dict(x=f(B, D), h=f(B, D), qkv=f(B, 3 * D), att=f(B, D), g=f(B, 4 * D), po=f(self.ks_o, B, D), pd=f(self.ks_p, B, D),
                           # packed operand images of the v2 path (rows padded to a 16-row tile, pad rows stay 0)
                           x_pk=torch.zeros((B + 15) // 16 * 16, D, device=dev), x2_pk=torch.zeros((B + 15) // 16 * 16, D, device=dev),
                           att_pk=torch.zeros((B + 15) // 16 * 16, D, device=dev), g_pk=torch.zeros((B + 15) // 16 * 16, 4 * D, device=dev),
                           pd_pk=torch.zeros(4, (B + 15) // 16 * 16, D, device=dev),
                           # split-context attention records of the batch-1 row path (ops.decode_attn_parts -> ops.gemv_row(parts=...))
                           parts=torch.zeros(B, self.H, max(1, min(16, int(self.tune.get("row_splits") or 8))), ops.ATTN_PART_REC, device=dev)),
                  graph=None, samp_dev=torch.zeros(B, 8, device=dev),
                  # geometry + caller-owned split-context workspace of this state's attention launches (cbx_decode_attn_t, ABI v10)
                  da=ops.DecodeAttnGeom(dev, unroll=0 if int(self.knobs["da_u"]) == 4 else int(self.knobs["da_u"]), pipeline=int(self.knobs["da_pipe"]),
                                        split=B * self.H < 128))
        self._state[key] = st
        return st

    def _prefill_c(self, xf, h, qkv, att, g, pos, crow, st, B, S, prefix):
        """The prefill through cbx_gpt2_prefill (include/cbx.h, ABI v16): the nine launches per layer of generate()'s Python sequence, issued in C."""
        import ctypes
        from ._lib import Gpt2Layer, Gpt2Prefill, check, lib
        p = lambda t: t.data_ptr()
        arr = (Gpt2Layer * self.L)()  # built per call: a cached array would outlive a re-loaded weight tensor
        for i, lw in enumerate(self.layers):
            a = arr[i]
            a.ln1_w, a.ln1_b, a.ln2_w, a.ln2_b = p(lw["ln1"][0]), p(lw["ln1"][1]), p(lw["ln2"][0]), p(lw["ln2"][1])
            a.wqkv, a.bqkv, a.wo, a.bo, a.wfc, a.bfc, a.wpr, a.bpr = (p(lw[k]) for k in ("wqkv", "bqkv", "wo", "bo", "wfc", "bfc", "wpr", "bpr"))
        d = Gpt2Prefill()
        d.n_layers, d.rows, d.S, d.prefix, d.dim, d.n_heads, d.eps, d.attn_scale, d.layers = self.L, B, S, prefix, self.D, self.H, 1e-5, 0.125, arr
        d.x, d.h, d.qkv, d.att, d.g, d.positions, d.cache_rows, d.kc, d.vc = p(xf), p(h), p(qkv), p(att), p(g), p(pos), p(crow), p(st["kc"]), p(st["vc"])
        d.kv_layer_stride, d.kv_row_stride, d.kv_head_stride = st["kc"].stride(0), st["kc"].stride(1), st["kc"].stride(2)
        check(lib.cbx_gpt2_prefill(ctypes.byref(d), ops._stream()), "cbx_gpt2_prefill")

    @ops.on_device
    @torch.inference_mode()
    def generate(self, conds, text_tokens, max_gen_len=1000, temperature=0.8, top_k=1000, top_p=0.95, repetition_penalty=1.2,
                 uniforms=None, ban_eos=False, ban_from=0, use_graph=True, poll_every=16, debug_logits=False):
        """conds: one cond dict (speaker_emb (1,256), cond_prompt_speech_tokens (1,375)) or a list of B; text_tokens: list of B
        1-D LongTensors (GPT-2 BPE ids, no SOT/EOT).  Returns a list of B 1-D LongTensors without the trailing EOS."""
        dev, B, D = self.dev, len(text_tokens), self.D
        voice = conds if isinstance(conds, dict) else None  # one voice for the whole batch: its conditioning prefix may be cached
        conds = [conds] * B if isinstance(conds, dict) else conds
        assert B >= 1, "empty batch"
        if uniforms is not None:
            uniforms = torch.as_tensor(uniforms, dtype=torch.float32)
            assert uniforms.numel() % B == 0 and uniforms.numel() // B >= max_gen_len + 1, \
                f"uniforms must hold at least max_gen_len + 1 = {max_gen_len + 1} draws per utterance"
            uniforms = uniforms.view(B, -1)
        if B > self.MAX_BATCH:  # one row per utterance (no CFG); the decode GEMV serves M <= 64 rows
            assert not debug_logits, "sub-batching is only defined for the plain token path"
            out = []
            for lo in range(0, B, self.MAX_BATCH):
                hi = min(B, lo + self.MAX_BATCH)
                out += self.generate(conds[lo:hi], text_tokens[lo:hi], max_gen_len=max_gen_len, temperature=temperature, top_k=top_k,
                                     top_p=top_p, repetition_penalty=repetition_penalty, uniforms=None if uniforms is None else uniforms[lo:hi],
                                     ban_eos=ban_eos, ban_from=ban_from, use_graph=use_graph, poll_every=poll_every)
            return out
        n_prompt = [int(c["cond_prompt_speech_tokens"].numel()) for c in conds]
        tl = [int(t.numel()) for t in text_tokens]
        s0 = [1 + n_prompt[b] + tl[b] + 1 for b in range(B)]
        S = max(s0)
        n_samples = max_gen_len + 1
        max_ctx = (S + n_samples + 63) // 64 * 64
        assert max_ctx <= self.wpe.shape[0], "context exceeds GPT-2 n_positions"
        st = self._get_state(B, max_ctx, n_samples)
        if self.decode_mode == "v2":
            self._prepare_tune()
        # sampling parameters live in device memory (cbx_sampler_t.dev_params): no graph re-capture when a request changes them
        st["samp_dev"].copy_(torch.tensor([0.0, float(temperature), 0.0, float(top_p), float(repetition_penalty), float(top_k),
                                           float(STOP_SPEECH if ban_eos else -1), float(ban_from)]).repeat(B, 1), non_blocking=True)
        for k in ("seen", "step", "done", "n_generated", "out_tokens"):
            st[k].zero_()
        st["seen"][:, START_SPEECH] = 1  # the first processor call sees ids = [start token] (t3.py:428)
        if uniforms is None:
            st["uniforms"].uniform_()
        else:
            st["uniforms"].copy_(torch.as_tensor(uniforms, dtype=torch.float32).view(B, -1)[:, :n_samples])

        # ---- prefill: [speaker | prompt-token embeddings | text | start-speech] + wpe (prepare_input_embeds, t3.py:102-130,407-423)
        # The 1 + n_prompt conditioning positions see only themselves (causal) and carry absolute positions: with their K / V cached (VoicePrefixCache) only the
        # text positions and the start token are computed -- 65 of 441 positions at 64 text tokens -- against keys read from the KV cache.
        exact = ops._prec() not in (3, 6, 16)
        pre = self._voice_prefix(voice) if voice is not None and exact else None
        if pre is not None and pre["P"] != 1 + n_prompt[0]:
            pre = None
        P0 = pre["P"] if pre is not None else 0
        Sx = S - P0
        x = torch.zeros(B, Sx, D, device=dev)
        for b in range(B):
            pos = torch.arange(s0[b], dtype=torch.int32, device=dev)
            if pre is None:
                ops.linear(conds[b]["speaker_emb"].to(dev).float().view(1, 256), self.spkr_w, x[b, 0:1], bias=self.spkr_b)
                ops.axpby(self.wpe[0:1], x[b, 0:1], 1.0, 1.0)
                a, e = 1, 1 + n_prompt[b]
                ops.embed(conds[b]["cond_prompt_speech_tokens"].to(dev).long().view(-1), self.speech_emb, x[b, a:e], table2=self.wpe, ids2=pos[a:e])
            a, e = 1 + n_prompt[b], 1 + n_prompt[b] + tl[b]
            ops.embed(text_tokens[b].to(dev).long().view(-1), self.text_emb, x[b, a - P0:e - P0], table2=self.wpe, ids2=pos[a:e])
            ops.embed(torch.full((1,), START_SPEECH, dtype=torch.int64, device=dev), self.speech_emb, x[b, e - P0:e - P0 + 1], table2=self.wpe, ids2=pos[e:e + 1])
        M = B * Sx
        xf = x.view(M, D)
        h, qkv, att, g = (torch.empty(M, n, device=dev) for n in (D, 3 * D, D, 4 * D))
        posr = torch.arange(P0, S, dtype=torch.int32, device=dev).repeat(B)
        crow = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(Sx)
        if pre is not None:
            self._paste_voice_prefix(pre, st, B)
        if self.c_prefill and exact and not ops.TIMER:
            self._prefill_c(xf, h, qkv, att, g, posr, crow, st, B, Sx, P0)
        else:
            for i, lw in enumerate(self.layers):
                ops.layernorm(xf, lw["ln1"][0], lw["ln1"][1], h, 1e-5)
                ops.linear(h, lw["wqkv"], qkv, bias=lw["bqkv"])
                ops.rope_kv(qkv, posr, None, None, st["kc"][i], st["vc"][i], self.H, cache_rows=crow)
                q4 = qkv.view(B, Sx, 3, self.H, 64)
                if pre is None:
                    ops.flash_attn(q4[:, :, 0], q4[:, :, 1], q4[:, :, 2], att.view(B, Sx, self.H, 64), 0.125, causal=True)
                else:  # keys / values [cached prefix | text] where the cache keeps them (cbx_flash_attn_kv_f32)
                    ops.flash_attn(q4[:, :, 0], st["kc"][i][:B, :, :S].permute(0, 2, 1, 3), st["vc"][i][:B, :, :S].permute(0, 2, 1, 3),
                                   att.view(B, Sx, self.H, 64), 0.125, causal=True)
                ops.linear(att, lw["wo"], xf, bias=lw["bo"], residual=xf)
                ops.layernorm(xf, lw["ln2"][0], lw["ln2"][1], h, 1e-5)
                ops.linear(h, lw["wfc"], g, bias=lw["bfc"], act=ops.GELU_TANH)
                ops.linear(g, lw["wpr"], xf, bias=lw["bpr"], residual=xf)
        if pre is None and voice is not None and exact:
            self._keep_voice_prefix(voice, st, P=1 + n_prompt[0])
        last = torch.tensor([b * Sx + s0[b] - P0 - 1 for b in range(B)], device=dev)
        hl = xf.index_select(0, last).contiguous()
        ops.layernorm(hl, self.lnf[0], self.lnf[1], st["dws"]["h"], 1e-5)
        ops.linear(st["dws"]["h"], self.head, st["logits"], bias=self.head_b)
        del x, xf, h, qkv, att, g

        s0t = torch.tensor(s0, dtype=torch.int32, device=dev)
        st["positions"].copy_(s0t - 1)
        st["ctx_lens"].copy_(s0t)
        step_logits = [st["logits"].clone()] if debug_logits else None
        self._sample(st)
        # later processor calls see ids = tokens generated so far, without the start token (t3.py:448-449)
        st["seen"][:, START_SPEECH] = (st["out_tokens"][:, 0] == START_SPEECH).to(torch.uint8)
        if debug_logits:
            use_graph = False
        if use_graph and st["graph"] is None and n_samples > 1:
            torch.cuda.synchronize()
            saved = {k: st[k].clone() for k in ("seen", "step", "done", "n_generated", "out_tokens", "next_ids", "next_pos_ids",
                                                "positions", "ctx_lens", "logits")}
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                self._decode_step(st)
            for k, v in saved.items():
                st[k].copy_(v)
            st["graph"] = gr
        ev = None
        if self.time_decode:  # two HIP events around the decode loop on its launch stream (bench.py: in-run decode-step roofline)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        n_replays = 0
        for i in range(1, n_samples):
            n_replays += 1
            if use_graph and st["graph"] is not None:
                st["graph"].replay()
            elif debug_logits:
                self._forward(st)
                step_logits.append(st["logits"].clone())
                self._sample(st)
            else:
                self._decode_step(st)
            if not ban_eos and (i % poll_every == 0) and bool(st["done"].all()):
                break
        if ev is not None:
            ev[1].record()
            self.decode_events.append((ev[0], ev[1], n_replays, list(s0), B))
        n = st["n_generated"].tolist()
        toks = st["out_tokens"].cpu()
        out = []
        for b in range(B):
            t = toks[b, : n[b]]
            out.append(t[:-1].clone() if n[b] and int(t[-1]) == STOP_SPEECH else t.clone())
        return (out, torch.stack(step_logits)) if debug_logits else out
