"""T3 speech-token transformer on MI355X: host-side mirror of `T3.inference` (reference models/t3/t3.py:226-390).

Python only sequences kernel launches; all arithmetic runs in libcbx_hip.so.  One decode step (embedding gather ->
30 x [RMSNorm, fused QKV GEMM, RoPE + KV append, decode attention, O-proj + residual, RMSNorm, gate/up GEMM with
SwiGLU epilogue, down-proj + residual] -> final norm -> speech head -> device sampler) is captured ONCE as a
hipGraph and replayed per token: positions, context lengths, the next token id and the EOS flags live in device
memory, so there is no host synchronisation inside the loop (the reference syncs every token, t3.py:366).

Batching (not in the reference, which is batch-1): utterance b occupies row b (conditional) and row B+b (CFG
unconditional); rows are right-padded to a common prefill length -- causal attention keeps padding invisible and
the decode steps then overwrite it in the KV cache at each row's own position.
"""
import math

import os

import torch

from . import ops, weights

START_SPEECH, STOP_SPEECH = 6561, 6562
START_TEXT, STOP_TEXT = 255, 0


def llama3_rope_tables(max_pos, head_dim=64, theta=500000.0, factor=8.0, low=1.0, high=4.0, orig=8192):
    """cos/sin tables [max_pos][64] for HF rope_type 'llama3' (reference llama_configs.py:22-30)."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    wl = 2 * math.pi / inv
    scaled = torch.where(wl > orig / low, inv / factor, inv)
    smooth = (orig / wl - low) / (high - low)
    mid = (1 - smooth) * scaled / factor + smooth * scaled
    is_mid = ~(wl < orig / high) & ~(wl > orig / low)
    inv = torch.where(is_mid, mid, scaled)
    fr = torch.arange(max_pos).float()[:, None] * inv[None, :]
    emb = torch.cat([fr, fr], -1)
    return emb.cos().contiguous(), emb.sin().contiguous()


class _LoopHandle:
    """Owns a cbx_t3_loop_t (destroyed with the state / geometry entry that holds it)."""

    def __init__(self, h):
        self.h = h

    def __del__(self):
        try:
            from ._lib import lib
            if self.h:
                lib.cbx_t3_loop_destroy(self.h)
        except Exception:
            pass


class VoicePrefixCache:
    """K / V of a voice's conditioning positions at every layer, kept after the first prefill with that voice (round 6; T3Engine: 34 positions, the GPT-2 backbones:
    1 + the prompt tokens).  The prompt of T3.inference / inference_turbo is [conditioning | text | BOS] under a causal mask (t3.py:303-335, 407-423): the conditioning
    positions see only themselves, so their K / V depend on the voice alone.  An entry keeps the conditioning tensors alive and is matched by identity + version
    counter (+ content for host tensors; a device tensor made under inference_mode has no counter: identity alone): a hit means the very tensors the prefix was
    computed from.  No device-side compare: generate() must not synchronise (the throughput schedule enqueues ahead)."""
    _PREFIX_KEEP = 4

    @staticmethod
    def _tver(t):
        try:
            return t._version
        except RuntimeError:  # "Inference tensors do not track version counter"
            return -1

    def _voice_prefix(self, conds):
        if not (self.share_prefix and isinstance(conds, dict)):
            return None
        for ent in self._prefix_cache:
            ok = ent["keys"] == sorted(conds)
            for k, v, ver, snap in ent["items"] if ok else ():
                c = conds[k]
                if not torch.is_tensor(v):
                    ok = not torch.is_tensor(c) and c == v
                else:  # the very tensor, unmodified: version counter where the tensor has one (inference tensors do not), content for host tensors (a few KB)
                    ok = c is v and self._tver(c) == ver and (snap is None or torch.equal(c, snap))
                if not ok:
                    break
            if ok:
                return ent
        return None

    def _keep_voice_prefix(self, conds, st, P=34):
        """After a full prefill: K / V of positions 0 .. P - 1 of cache row 0 (L, H, P, 64)."""
        if not (self.share_prefix and isinstance(conds, dict)) or self._voice_prefix(conds) is not None:
            return
        items = [(k, v, self._tver(v) if torch.is_tensor(v) else None, v.clone() if torch.is_tensor(v) and v.device.type == "cpu" else None)
                 for k, v in sorted(conds.items())]
        ent = dict(keys=sorted(conds), items=items, P=P, k=st["kc"][:, 0, :, :P].clone(), v=st["vc"][:, 0, :, :P].clone(), ev=None)
        if self.dev.type == "cuda":
            ent["ev"] = torch.cuda.Event()
            ent["ev"].record()
        self._prefix_cache.insert(0, ent)
        del self._prefix_cache[self._PREFIX_KEEP:]

    def _paste_voice_prefix(self, pre, st, rows):
        """The cached prefix into rows 0 .. rows - 1 of the KV cache (the entry may have been written on another stream: two decode chains of the throughput schedule)."""
        if pre["ev"] is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(pre["ev"])
            pre["k"].record_stream(cur), pre["v"].record_stream(cur)
        st["kc"][:, :rows, :, :pre["P"]].copy_(pre["k"][:, None])
        st["vc"][:, :rows, :, :pre["P"]].copy_(pre["v"][:, None])


class T3Engine(VoicePrefixCache):
    D, H, HD, F = 1024, 16, 64, 4096
    MAX_BATCH = 32  # utterances per device batch: 2 CFG rows each, decode GEMV serves M <= 64 rows
    # decode launch geometry: waves per 16-column tile (nw) / cross-workgroup K splits; *2 = the packed-operand (v2) path
    # qkv_tc / od_tc: output columns per workgroup of the q/k/v resp. the o / down projections (0: 16 resp. what half_tiles says; 12 puts
    # q/k/v, 4 puts o / down on exactly 256 workgroups -- with od_tc = 4 and d_ks2 = 1 the down projection needs no partial images and the
    # next q/k/v GEMV no partial-sum fold).  CBX_T3_TUNE="qkv_tc=12,od_tc=4,d_ks2=1,d_nw2=8" overrides any of these for an A/B.
    # qkv_ks / qkv_ct (ABI v11): the q/k/v projection as qkv_ks split-K partial sums over workgroups of qkv_ct column tiles, folded by the attention
    # launch (a CU then moves 96 KiB instead of 256 per q/k/v launch: profiles/r04_decode_launch_timeline.txt); head_ct: column tiles of the head GEMV
    _TUNE = dict(qkv_nw=8, o_ks=4, gu_nw=8, d_ks=8, head_nw=4, o_nw2=8, d_ks2=2, d_nw2=16, half_tiles=1, qkv_tc=0, od_tc=0, prefill_prec=0, qkv_ks=4, qkv_ct=3,
                 head_ct=2)

    @ops.on_device
    def __init__(self, sd, device="cuda", n_layers=None, max_pos=4608, weight_dtype=None):
        self.dev = torch.device(device)
        if n_layers is None:
            n_layers = 0
            while f"tfmr.layers.{n_layers}.input_layernorm.weight" in sd:
                n_layers += 1
        self.L = n_layers
        d = lambda t: t.float().contiguous().to(self.dev)
        self.layers = []
        for i in range(n_layers):
            p = f"tfmr.layers.{i}."
            self.layers.append(dict(
                ln1=d(sd[p + "input_layernorm.weight"]), ln2=d(sd[p + "post_attention_layernorm.weight"]),
                wqkv=d(torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)),
                wo=d(sd[p + "self_attn.o_proj.weight"]),
                wgu=d(weights.pack_swiglu(sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"])),
                wd=d(sd[p + "mlp.down_proj.weight"])))
        # decode path: the same weights in the lane-ordered packed layout of cbx_gemv_f32 (every wave-level load is 1 KiB contiguous)
        self.decode_mode = os.environ.get("CBX_T3_DECODE", "v2")
        # OPT-IN decode numerics (CBX_T3_WEIGHTS=bf16 / T3Engine(weight_dtype="bf16")): the decode-step weight images are rounded to bf16
        # (half the streamed bytes; activations, accumulation, KV cache and the prefill stay fp32).  Not the parity path: sampled
        # tokens differ from the fp32 reference's; stated bound: first-step logits within 5e-2 (SURVEY.md 8d bf16 mode).
        self.weight_dtype = weight_dtype or os.environ.get("CBX_T3_WEIGHTS", "fp32")
        assert self.weight_dtype in ("fp32", "bf16") and (self.weight_dtype == "fp32" or self.decode_mode == "v2")
        bf = self.weight_dtype == "bf16"
        if self.decode_mode == "v2":
            for i, lw in enumerate(self.layers):
                p = f"tfmr.layers.{i}."
                lw["wqkv_pk"] = ops.pack_gemv_weight(lw["wqkv"], bf16=bf)
                lw["wo_pk"] = ops.pack_gemv_weight(lw["wo"], bf16=bf)
                lw["wgu_pk"] = ops.pack_gemv_weight(d(torch.cat([sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]], 0)), swiglu=True, bf16=bf)
                lw["wd_pk"] = ops.pack_gemv_weight(lw["wd"], bf16=bf)
                if self._TUNE.get("half_tiles"):  # 8-column tiles for the two N = 1024 projections (128 instead of 64 output tiles)
                    lw["wo_pk8"] = ops.pack_gemv_weight(lw["wo"], half_tile=True, bf16=bf)
                    lw["wd_pk8"] = ops.pack_gemv_weight(lw["wd"], half_tile=True, bf16=bf)
        self.norm = d(sd["tfmr.norm.weight"])
        self.text_emb, self.speech_emb = d(sd["text_emb.weight"]), d(sd["speech_emb.weight"])
        self.text_pos, self.speech_pos = d(sd["text_pos_emb.emb.weight"]), d(sd["speech_pos_emb.emb.weight"])
        self.head = d(sd["speech_head.weight"])
        self.V = self.head.shape[0]
        self.head_pk = ops.pack_gemv_weight(self.head, bf16=bf) if self.decode_mode == "v2" else None
        c = "cond_enc."
        self.spkr_w, self.spkr_b = d(sd[c + "spkr_enc.weight"]), d(sd[c + "spkr_enc.bias"])
        self.emo_w = d(sd[c + "emotion_adv_fc.weight"].view(-1))
        self.pq = d(sd[c + "perceiver.pre_attention_query"][0])
        self.p_ln = (d(sd[c + "perceiver.attn.norm.weight"]), d(sd[c + "perceiver.attn.norm.bias"]))
        self.p_q = (d(sd[c + "perceiver.attn.to_q.weight"]), d(sd[c + "perceiver.attn.to_q.bias"]))
        self.p_kv = (d(torch.cat([sd[c + "perceiver.attn.to_k.weight"], sd[c + "perceiver.attn.to_v.weight"]], 0)),
                     d(torch.cat([sd[c + "perceiver.attn.to_k.bias"], sd[c + "perceiver.attn.to_v.bias"]], 0)))
        self.p_out = (d(sd[c + "perceiver.attn.proj_out.weight"]), d(sd[c + "perceiver.attn.proj_out.bias"]))
        cos, sin = llama3_rope_tables(max_pos)
        self.cos, self.sin = cos.to(self.dev), sin.to(self.dev)
        self.max_pos = max_pos
        self._state = {}
        # decode launch geometry: waves per 16-column tile (nw) and cross-workgroup K splits of the two down-projections
        self.c_step = os.environ.get("CBX_T3_CSTEP", "1") == "1"  # token step through the stage-level C entry point (same kernels)
        self.c_loop = self.c_step and os.environ.get("CBX_T3_CLOOP", "1") == "1"  # ... and the token loop through cbx_t3_loop_* (the hipGraph captured and replayed in C)
        self.time_decode, self.decode_events = False, []  # (start, end, steps, prefill lengths, rows) per generate() when enabled
        self.tune = self._env_tune()
        self.knobs = self._env_knobs()
        # K / V of a voice's 34 conditioning positions at every layer, kept after the first prefill with that voice (round 6; see _voice_prefix)
        self.share_prefix = os.environ.get("CBX_T3_SHARE_PREFIX", "1") == "1"
        self._prefix_cache = []

    # ------------------------------------------------------------------ packed device layout <-> disk (formats.save_packed / load_packed)
    _PLAIN = ("norm", "text_emb", "speech_emb", "text_pos", "speech_pos", "head", "head_pk", "spkr_w", "spkr_b", "emo_w", "pq", "cos", "sin")
    _PAIRS = ("p_ln", "p_q", "p_kv", "p_out")

    def export_packed(self):
        """Every tensor this engine holds after load-time packing, under flat stable names."""
        t = {f"layers.{i}.{k}": v for i, lw in enumerate(self.layers) for k, v in lw.items()}
        t.update({k: getattr(self, k) for k in self._PLAIN if getattr(self, k) is not None})
        for k in self._PAIRS:
            t[k + ".0"], t[k + ".1"] = getattr(self, k)
        return t

    @classmethod
    def from_packed(cls, t, device="cuda", max_pos=4608):
        """Rebuild the engine from `export_packed()` tensors (e.g. memory-mapped by formats.load_packed): no re-packing, one H2D copy each."""
        self = cls.__new__(cls)
        self.dev = torch.device(device)
        d = lambda v: v.to(self.dev) if v.device != self.dev else v
        n = 0
        while f"layers.{n}.ln1" in t:
            n += 1
        self.L = n
        self.layers = [{k.split(".", 2)[2]: d(v) for k, v in t.items() if k.startswith(f"layers.{i}.")} for i in range(n)]
        for k in self._PLAIN:
            setattr(self, k, d(t[k]) if k in t else None)
        for k in self._PAIRS:
            setattr(self, k, (d(t[k + ".0"]), d(t[k + ".1"])))
        self.V = self.head.shape[0]
        self.decode_mode = "v2" if self.head_pk is not None else "v1"
        self.weight_dtype = "bf16" if (self.head_pk is not None and self.head_pk.dtype == torch.bfloat16) else "fp32"
        self.max_pos = self.cos.shape[0]
        self._state = {}
        self.time_decode, self.decode_events = False, []
        self.c_step = os.environ.get("CBX_T3_CSTEP", "1") == "1"
        self.c_loop = self.c_step and os.environ.get("CBX_T3_CLOOP", "1") == "1"
        self.tune = cls._env_tune()
        self.knobs = cls._env_knobs()
        self.share_prefix, self._prefix_cache = os.environ.get("CBX_T3_SHARE_PREFIX", "1") == "1", []
        return self

    @classmethod
    def _env_tune(cls):
        tune = dict(cls._TUNE)
        for kv in filter(None, os.environ.get("CBX_T3_TUNE", "").split(",")):
            k, v = kv.split("=")
            assert k.strip() in tune, f"CBX_T3_TUNE: unknown knob {k!r} (known: {sorted(tune)})"
            tune[k.strip()] = int(v)
        return tune

    @staticmethod
    def _env_knobs():
        """Per-ENGINE launch knobs of the decode attention / GEMVs (da_pipe, da_u, deep, pre_epi): they travel in every call's descriptor
        (cbx_decode_attn_t, cbx_gemv_t.flags; ABI v10), nothing is process-wide.  Defaults from CBX_DA_PIPE / CBX_DA_U / CBX_GEMV_* (A/B scripts)."""
        from .autotune import env_knobs
        return env_knobs()

    def _gf(self):
        return ops.gemv_flags(self.knobs.get("pre_epi"), self.knobs.get("deep"), self.knobs.get("shallow"))

    def _sync_geom(self, st):
        st["da"].unroll, st["da"].pipeline = (0 if int(self.knobs["da_u"]) == 4 else int(self.knobs["da_u"])), int(self.knobs["da_pipe"])

    def _qks(self):
        """Effective split-K factor of the q/k/v projection (0 = the one-tile kernel): the column-tile form serves the 16-column fp32 image."""
        ks = int(self.tune.get("qkv_ks") or 0)
        return ks if ks > 1 and not self.tune.get("qkv_tc") and self.weight_dtype == "fp32" and self.D % (256 * ks) == 0 else 0

    def _hct(self):
        return int(self.tune.get("head_ct") or 0) if self.weight_dtype == "fp32" and self.D % 256 == 0 else 0

    def _tiles(self):
        """(q/k/v tile width, o / down tile width) of the current tune: 16, 12, 8 or 4 output columns per workgroup."""
        tn = self.tune
        od = tn.get("od_tc") or (8 if tn.get("half_tiles") and "wo_pk8" in self.layers[0] else 16)
        return tn.get("qkv_tc") or 16, od

    def _image(self, lw, name, tc):
        """The packed decode image of layer weight `name` for `tc`-column tiles (packed on first use: never inside a stream capture,
        generate() calls _prepare_tune() first)."""
        key = f"{name}_pk" if tc == 16 else f"{name}_pk{tc}"
        if key not in lw:
            lw[key] = ops.pack_gemv_weight(lw[name], half_tile=tc, bf16=self.weight_dtype == "bf16")
        return lw[key]

    def _prepare_tune(self):
        if self.decode_mode != "v2":
            return
        qtc, odtc = self._tiles()
        assert qtc in (16, 12) and odtc in (16, 8, 4), f"tile widths {qtc} / {odtc}"
        tn = self.tune
        assert tn.get("qkv_ks", 0) in (0, 1, 2, 4) and 1 <= tn.get("qkv_ct", 3) <= 4 and 0 <= tn.get("head_ct", 0) <= 4, f"column-tile geometry {tn}"
        # (qkv_tc = 12 or bf16 weight images switch the split-K / column-tile forms off: _qks(), _hct())
        for lw in self.layers:
            self._image(lw, "wqkv", qtc), self._image(lw, "wo", odtc), self._image(lw, "wd", odtc)

    # ------------------------------------------------------------------ decode-step geometry: measured, not guessed (autotune.py)
    def apply_variant(self, tune, knobs=None):
        """Switch the decode geometry of THIS engine: `tune` replaces self.tune, `knobs` (da_pipe, da_u, deep, pre_epi) the launch knobs of the
        decode attention / GEMV load batches -- per-call descriptor fields since ABI v10, so another engine in the process keeps its own.
        Captured decode graphs and C step descriptors bake the geometry in: they are dropped."""
        from .autotune import LIB_KNOBS, canon
        old_key = canon(self.tune, self.knobs) + (("shallow", int(bool(self.knobs.get("shallow")))),)
        self.tune = dict(tune)
        self.knobs = dict(LIB_KNOBS, **(knobs or {}))
        assert int(self.knobs["da_pipe"]) in range(8) and int(self.knobs["da_u"]) in (4, 8, 16)
        new_key = canon(self.tune, self.knobs) + (("shallow", int(bool(self.knobs.get("shallow")))),)
        for st in self._state.values():
            # the captured graph / C step descriptor of a state are kept PER GEOMETRY (alternating synthesize() and synthesize_pipelined()
            # re-captured the decode graph on every switch); they reference the state's own buffers and the weight images, which both outlive the switch
            cache = st.setdefault("geom_cache", {})
            cache[old_key] = (st.get("graph"), st.pop("cstep", None))
            st["graph"], cstep = cache.pop(new_key, (None, None))
            if cstep is not None:
                st["cstep"] = cstep
            self._sync_geom(st)
        if getattr(self, "_co_res", False) and not getattr(self, "_co_switching", False):
            # a geometry adopted WHILE the co-resident form is on becomes the one co_resident(False) returns to (its co-resident keys aside)
            self._co_saved = (dict(self._co_saved[0], **{k: v for k, v in self.tune.items() if k not in ("half_tiles", "d_ks2", "d_nw2")}),
                              dict(self._co_saved[1], **{k: v for k, v in self.knobs.items() if k != "shallow"}))

    def co_resident(self, on):
        """The decode step on the geometry whose workgroups fit on a CU BESIDE a co-resident flow workgroup (engine.synthesize_pipelined; profiles/r05_overlap_*):
        every launch <= 8 waves x <= 128 VGPRs -- the down projection on 512-thread workgroups with 4 split-K partial images (a member of the hardware-green
        allow-list: decode_green.json {d_ks2: 4, d_nw2: 8, half_tiles: 0}), gate | up on its 2-deep load batches (CBX_GEMV_SHALLOW: same products, same order).
        on=False: back to the engine's default geometry.  Switching drops the captured decode graphs."""
        if getattr(self, "_co_res", False) == bool(on):
            return
        self._co_switching = True
        try:
            if on:
                self._co_saved = (dict(self.tune), dict(self.knobs))
                self.apply_variant(dict(self.tune, half_tiles=0, d_ks2=4, d_nw2=8), dict(self.knobs, shallow=1))
            else:
                self.apply_variant(*self._co_saved)
        finally:
            self._co_switching = False
        self._co_res = bool(on)

    @ops.on_device
    @torch.inference_mode()
    def measure_decode(self, B=8, ctx=224, steps=24, reps=2, use_graph=True, slot=7):
        """(ms per token step, logits after ONE step) of the current geometry on a seeded synthetic decode state: 2 B rows that hold `ctx`
        cached positions each.  The step is the one generate() replays (embedding .. sampler, one hipGraph); time = HIP events around
        `steps` replays, best of `reps`.  The logits are a pure function of the seed, so two geometries can be compared bit for bit."""
        import time
        rows = 2 * B
        total = 4 + reps * steps
        max_ctx = (ctx + total + 1 + 63) // 64 * 64
        assert max_ctx <= self.max_pos
        st = self._get_state(B, max_ctx, total + 1, slot)
        self._prepare_tune()
        st["graph"] = None
        st.pop("cstep", None)
        st.pop("geom_cache", None)
        gen = torch.Generator(device=self.dev).manual_seed(20240229)
        for k in ("kc", "vc"):
            st[k].normal_(0.0, 0.5, generator=gen)
        st["uniforms"].uniform_(generator=gen)
        for k in ("seen", "step", "done", "n_generated", "out_tokens"):
            st[k].zero_()
        st["samp_dev"].copy_(torch.tensor([0.5, 0.8, 0.05, 1.0, 1.2, 0.0, float(STOP_SPEECH), 6561.0]).repeat(B, 1))
        ids = torch.tensor([(911 * b + 17) % 6561 for b in range(B)] * 2, dtype=torch.int64)
        st["next_ids"].copy_(ids)
        st["next_pos_ids"].fill_(1)
        st["positions"].fill_(ctx)
        st["ctx_lens"].fill_(ctx + 1)
        self._decode_step(st)
        logits = st["logits"].clone()
        if steps <= 0:  # numerics only
            return 0.0, logits
        cuda = self.dev.type == "cuda"
        if cuda and use_graph:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._decode_step(st)
            step = g.replay
        else:
            step = lambda: self._decode_step(st)
        step()
        best = float("inf")
        for _ in range(reps):
            if cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(steps):
                    step()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1)
            else:
                t0 = time.perf_counter()
                for _ in range(steps):
                    step()
                ms = 1e3 * (time.perf_counter() - t0)
            best = min(best, ms / steps)
        if cuda:
            torch.cuda.synchronize()
        # the state the whole run ended in (a pure function of the seed and of the arithmetic)
        self.last_measure = dict(final_logits=st["logits"].clone(), out_tokens=st["out_tokens"].clone())
        st["graph"] = None
        st.pop("cstep", None)
        return best, logits

    @ops.on_device
    @torch.inference_mode()
    def probe_decode(self, B=8, ctxs=(1, 38, 63, 64, 65, 225, 640), slot=7):
        """Logits of ONE token step per entry of `ctxs` on a seeded synthetic state whose rows hold RAGGED contexts (row r: (c - 1 + 37 r) mod
        704 cached positions): what two geometries that claim the same arithmetic must agree on bit for bit.  The autotuner's identity check
        (one step at the bench context never looked at contexts shorter than one 64-position attention step, or at the split grid)."""
        rows, max_ctx = 2 * B, 704
        st = self._get_state(B, max_ctx, 4, slot)
        self._prepare_tune()
        st["graph"] = None
        st.pop("cstep", None)
        gen = torch.Generator(device=self.dev).manual_seed(20250922)
        for k in ("kc", "vc"):
            st[k].normal_(0.0, 0.5, generator=gen)
        st["uniforms"].uniform_(generator=gen)
        st["samp_dev"].copy_(torch.tensor([0.5, 0.8, 0.05, 1.0, 1.2, 0.0, float(STOP_SPEECH), 6561.0]).repeat(B, 1))
        ids = torch.tensor([(911 * b + 17) % 6561 for b in range(B)] * 2, dtype=torch.int64)
        out = []
        for c in ctxs:
            for k in ("seen", "step", "done", "n_generated", "out_tokens"):
                st[k].zero_()
            pos = torch.tensor([(c - 1 + 37 * r) % (max_ctx - 8) for r in range(rows)], dtype=torch.int32)
            st["next_ids"].copy_(ids)
            st["next_pos_ids"].fill_(1)
            st["positions"].copy_(pos)
            st["ctx_lens"].copy_(pos + 1)
            self._decode_step(st)
            out.append(st["logits"].clone())
        return torch.stack(out)

    def autotune(self, B=8, ctx=224, steps=24, reps=2, min_gain=0.01, allow_reorder=False, in_child=True, timeout=180.0, log=None, tiles=None, attn=None,
                 epi=None, validate=None, green_only=False):
        """Measure the decode-step geometries (autotune.py) and adopt the fastest one whose logits are bit-identical to the current
        geometry's.  in_child: the candidates run in a child process on synthetic weights of this shape, so a faulting candidate cannot take
        the serving process down; its failure leaves the geometry unchanged.  validate: a callable run on THIS engine after the fastest
        candidate overall -- possibly one that sums the down projection in another fp32 order -- has been applied; it is kept only if the
        callable returns True (e.g. "the serving workload's tokens are the ones the built-in geometry samples"), otherwise the bit-identical
        winner is.  Returns the report (also kept as self.autotune_report)."""
        from . import autotune as at
        knobs = dict(self.knobs)
        tune0 = dict(self.tune)
        if self.decode_mode != "v2" or 2 * B > 16:
            rep = dict(best={}, skipped="the tile / pipeline variants serve the packed <= 16-row decode path")
        elif in_child:
            base = {k: v for k, v in self.tune.items() if v != self._TUNE.get(k)}
            rep = at.tune_in_child(self.L, B, ctx, steps, reps, min_gain, allow_reorder, self.dev.index or 0, base, knobs, timeout, log, green_only=green_only)
        else:
            rep = at.tune_decode(self, B, ctx, steps, reps, min_gain, allow_reorder, use_graph=self.dev.type == "cuda", log=log,
                                 tiles=tiles or at.TILE_VARIANTS, attn=at.ATTN_VARIANTS if attn is None else attn,
                                 epi=at.EPI_VARIANTS if epi is None else epi, green_only=green_only)

        def adopt(v):
            t, k = at.split_variant(v)
            self.apply_variant(dict(tune0, **t), dict(knobs, **k))
            with torch.inference_mode():
                self._prepare_tune()

        best, best_any = rep.get("best") or {}, rep.get("best_any") or {}
        adopted = best
        if validate is not None and best_any and best_any != best:
            adopt(best_any)
            try:
                ok = bool(validate())
            except Exception as e:
                ok = False
                rep["validate_error"] = f"{type(e).__name__}: {e}"[:200]
            rep["best_any_validated"] = ok
            if ok:
                adopted = best_any
        if adopted or (validate is not None and best_any and best_any != best):
            adopt(adopted)
        rep["adopted"] = adopted
        for key in [k for k in self._state if k[3] == 7]:  # the measurement's own state (KV cache of the synthetic context)
            del self._state[key]
        self.autotune_report = rep
        return rep

    # ------------------------------------------------------------------ conditioning (t3.py:92-100, cond_enc.py:64-97)
    def _perceiver_block(self, x1, x2):
        """AttentionBlock2.forward (perceiver.py:156-170): x1 (n1,1024) queries attend to x2 (n2,1024)."""
        f = lambda *s: torch.empty(*s, device=self.dev)
        n1, n2 = x1.shape[0], x2.shape[0]
        h1, h2 = f(n1, 1024), f(n2, 1024)
        ops.layernorm(x1, self.p_ln[0], self.p_ln[1], h1)
        ops.layernorm(x2, self.p_ln[0], self.p_ln[1], h2)
        q, kv = f(n1, 1024), f(n2, 2048)
        ops.linear(h1, self.p_q[0], q, bias=self.p_q[1])
        ops.linear(h2, self.p_kv[0], kv, bias=self.p_kv[1])
        n2p = (n2 + 3) // 4 * 4
        s, pr = f(1, 4, n1, n2p), f(1, 4, n1, n2p)
        kv8 = kv.view(1, n2, 8, 256)  # heads 0-3 = K heads, 4-7 = V heads
        ops.bmm(q.view(1, n1, 4, 256).permute(0, 2, 1, 3), kv8[:, :, :4].permute(0, 2, 1, 3), s[..., :n2])
        ops.softmax_rows(s, pr, 1.0 / 16.0, n2)
        o = f(n1, 1024)
        ops.bmm(pr[..., :n2], kv8[:, :, 4:].permute(0, 2, 1, 3), o.view(1, n1, 4, 256).permute(0, 2, 1, 3), nn=True)
        out = f(n1, 1024)
        ops.linear(o, self.p_out[0], out, bias=self.p_out[1], residual=x1)
        return out

    @ops.on_device
    def cond_embeds(self, cond):
        """T3.prepare_conditioning -> (34, 1024): [speaker | 32 perceiver latents | emotion]."""
        dev = self.dev
        out = torch.empty(34, 1024, device=dev)
        spk = cond["speaker_emb"].to(dev).float().view(1, 256)
        ops.linear(spk, self.spkr_w, out[0:1], bias=self.spkr_b)
        toks = cond["cond_prompt_speech_tokens"].to(dev).long().view(-1)
        n = toks.shape[0]
        pe = torch.empty(n, 1024, device=dev)
        ops.embed(toks, self.speech_emb, pe, table2=self.speech_pos, ids2=torch.arange(n, dtype=torch.int32, device=dev))
        pre = self._perceiver_block(self.pq, pe)
        out[1:33] = self._perceiver_block(pre, pre)
        emo = float(torch.as_tensor(cond["emotion_adv"]).reshape(-1)[0])
        ops.axpby(self.emo_w.view(1, -1), out[33:34], a=emo, b=0.0)
        return out

    # ------------------------------------------------------------------ the conditioning prefix of a voice (round 6: VoicePrefixCache)
    # Later prefills with the SAME conditioning tensors run over the text positions only (a third fewer rows in every prefill GEMM at 64 text tokens) and read the
    # prefix keys from the cache (cbx_flash_attn_kv_f32).
    def _layer_prefill_text(self, lw, x, ws, St, S, rows, kc, vc, pos, crow):
        """_layer_prefill over the text positions alone: queries from the q | k | v workspace, keys / values [cached prefix | text] from the KV cache."""
        ops.layernorm(x, lw["ln1"], None, ws["h"], 1e-5, rms=True)
        ops.linear(ws["h"], lw["wqkv"], ws["qkv"])
        ops.rope_kv(ws["qkv"], pos, self.cos, self.sin, kc, vc, self.H, cache_rows=crow)
        q4 = ws["qkv"].view(rows, St, 3, self.H, 64)
        ops.flash_attn(q4[:, :, 0], kc[:rows, :, :S].permute(0, 2, 1, 3), vc[:rows, :, :S].permute(0, 2, 1, 3), ws["att"].view(rows, St, self.H, 64), 0.125,
                       causal=True)
        ops.linear(ws["att"], lw["wo"], x, residual=x)
        ops.layernorm(x, lw["ln2"], None, ws["h"], 1e-5, rms=True)
        ops.linear(ws["h"], lw["wgu"], ws["g"], swiglu=True)
        ops.linear(ws["g"], lw["wd"], x, residual=x)

    # ------------------------------------------------------------------ one transformer layer
    def _layer_prefill(self, lw, x, ws, S, rows, kc, vc, pos, crow):
        M = rows * S
        ops.layernorm(x, lw["ln1"], None, ws["h"], 1e-5, rms=True)
        ops.linear(ws["h"], lw["wqkv"], ws["qkv"])
        ops.rope_kv(ws["qkv"], pos, self.cos, self.sin, kc, vc, self.H, cache_rows=crow)
        q4 = ws["qkv"].view(rows, S, 3, self.H, 64)
        ops.flash_attn(q4[:, :, 0], q4[:, :, 1], q4[:, :, 2], ws["att"].view(rows, S, self.H, 64), 0.125, causal=True)
        ops.linear(ws["att"], lw["wo"], x, residual=x)
        ops.layernorm(x, lw["ln2"], None, ws["h"], 1e-5, rms=True)
        ops.linear(ws["h"], lw["wgu"], ws["g"], swiglu=True)
        ops.linear(ws["g"], lw["wd"], x, residual=x)

    def _prefill_c(self, xf, ws, S, rows, st, pos, crow):
        """The prefill through cbx_t3_prefill (include/cbx.h): the launches of _layer_prefill x n_layers, sequenced in C."""
        import ctypes
        from ._lib import T3Layer, T3Prefill, check, lib
        p = lambda t: t.data_ptr()
        # built per call (30 x 6 pointers): a cached array would outlive a re-packed / re-loaded weight tensor
        arr = (T3Layer * self.L)()
        for i, lw in enumerate(self.layers):
            arr[i].ln1, arr[i].ln2, arr[i].wqkv, arr[i].wo, arr[i].wgu, arr[i].wd = p(lw["ln1"]), p(lw["ln2"]), p(lw["wqkv"]), p(lw["wo"]), p(lw["wgu"]), p(lw["wd"])
        d = T3Prefill()
        d.n_layers, d.rows, d.S, d.dim, d.ffn, d.n_heads, d.precision = self.L, rows, S, self.D, self.F, self.H, int(self.tune.get("prefill_prec") or 0)
        d.eps, d.attn_scale, d.layers = 1e-5, 0.125, arr
        d.x, d.h, d.qkv, d.att, d.g = p(xf), p(ws["h"]), p(ws["qkv"]), p(ws["att"]), p(ws["g"])
        d.positions, d.cache_rows, d.cos_t, d.sin_t, d.kc, d.vc = p(pos), p(crow), p(self.cos), p(self.sin), p(st["kc"]), p(st["vc"])
        d.kv_layer_stride, d.kv_row_stride, d.kv_head_stride = st["kc"].stride(0), st["kc"].stride(1), st["kc"].stride(2)
        check(lib.cbx_t3_prefill(ctypes.byref(d), torch.cuda.current_stream().cuda_stream), "cbx_t3_prefill")

    def _forward_decode(self, st):
        """One token for every row.  The residual stream x is only touched by add_rmsnorm, which folds the split-K
        partials of the previous projection, the residual add and the RMSNorm into one pass."""
        ws, x, tn = st["dws"], st["dws"]["x"], self.tune
        h, qkv, att, g, po, pd = ws["h"], ws["qkv"], ws["att"], ws["g"], ws["po"][: tn["o_ks"]], ws["pd"][: tn["d_ks"]]
        ops.embed(st["next_ids"], self.speech_emb, x, table2=self.speech_pos, ids2=st["next_pos_ids"])
        part = None
        for i, lw in enumerate(self.layers):
            ops.add_rmsnorm(x, part, lw["ln1"], h)
            ops.gemv(h, lw["wqkv"], qkv, nw=tn["qkv_nw"])
            ops.decode_attn_rope(qkv, st["positions"], self.cos, self.sin, st["kc"][i], st["vc"][i], att, 0.125, geom=st["da"])
            ops.gemv(att, lw["wo"], po, ksplit=tn["o_ks"], nw=4)
            ops.add_rmsnorm(x, po, lw["ln2"], h)
            ops.gemv(h, lw["wgu"], g, swiglu=True, nw=tn["gu_nw"])
            ops.gemv(g, lw["wd"], pd, ksplit=tn["d_ks"], nw=4)
            part = pd
        ops.add_rmsnorm(x, part, self.norm, h)
        ops.gemv(h, self.head, st["logits"], nw=tn["head_nw"])

    def _forward_decode_v2(self, st):
        """5 launches per layer and no standalone norm / reduce kernels.  Every GEMV operand lives in the lane-ordered packed layout
        (written that way by its producer); RMSNorm is folded into the q/k/v, gate/up and head GEMVs (x * norm_w on the way to the
        MFMA, rstd in the epilogue); the o projection (4 MB) adds the residual in its epilogue; the down projection (16.8 MB, needs all
        256 CUs) emits `d_ks` split-K partial images that the NEXT consumer (q/k/v of the following layer, or the head) sums into its x
        operand on the fly, writing the new residual stream to the other ping-pong image."""
        ws, tn = st["dws"], self.tune
        rows, dks = st["rows"], tn["d_ks2"]
        cur, nxt, qkv, att, g, pd = ws["x_pk"], ws["x2_pk"], ws["qkv"], ws["att_pk"], ws["g_pk"], ws["pd_pk"][:dks]
        pk = dict(w_packed=True, x_packed=True, M=rows, flags=self._gf())
        ops.embed(st["next_ids"], self.speech_emb, cur, table2=self.speech_pos, ids2=st["next_pos_ids"], out_packed=True)
        red = {}  # partial images pending on the residual stream
        qtc, odtc = self._tiles()
        qt, ot = (0 if qtc == 16 else qtc), (0 if odtc == 16 else odtc)
        qks, qct, hct = self._qks(), int(tn.get("qkv_ct") or 3), self._hct()
        for i, lw in enumerate(self.layers):
            if qks > 1:  # split-K partial sums over column-tile workgroups; the attention launch adds them and applies rstd (ABI v11)
                qp, sq = ws["qkv_parts"][:qks], ws["qkv_ssq"][:qks]
                ops.gemv(cur, lw["wqkv_pk"], qp, N=3 * self.D, K=self.D, nw=8, norm_w=lw["ln1"], col_tiles=qct, ksplit=qks, ssq_out=sq, **pk, **red)
            else:
                ops.gemv(cur, self._image(lw, "wqkv", qtc), qkv, N=3 * self.D, K=self.D, nw=8, norm_w=lw["ln1"], half_tile=qt, **pk, **red)
            if red:
                cur, nxt = nxt, cur  # the q/k/v GEMV wrote x + sum(partials) to the other image
            if qks > 1:
                ops.decode_attn_rope(qp, st["positions"], self.cos, self.sin, st["kc"][i], st["vc"][i], att, 0.125, out_packed=True, geom=st["da"],
                                     qkv_ssq=sq, rms_dim=self.D, rms_eps=1e-5)
            else:
                ops.decode_attn_rope(qkv, st["positions"], self.cos, self.sin, st["kc"][i], st["vc"][i], att, 0.125, out_packed=True, geom=st["da"])
            ops.gemv(att, self._image(lw, "wo", odtc), cur, N=self.D, K=self.D, nw=tn["o_nw2"], res=cur, out_packed=True, half_tile=ot, **pk)
            ops.gemv(cur, lw["wgu_pk"], g, N=self.F, K=self.D, swiglu=True, nw=tn["gu_nw"], norm_w=lw["ln2"], out_packed=True, **pk)
            if dks > 1:
                ops.gemv(g, self._image(lw, "wd", odtc), pd, N=self.D, K=self.F, ksplit=dks, nw=tn["d_nw2"], out_packed=True, half_tile=ot, **pk)
                red = dict(xpart=pd, x_out=nxt)
            else:  # the down projection adds the residual itself: no partial images, no fold in the next q/k/v GEMV
                ops.gemv(g, self._image(lw, "wd", odtc), cur, N=self.D, K=self.F, nw=tn["d_nw2"], res=cur, out_packed=True, half_tile=ot, **pk)
        if red:
            red["x_out"] = None
        ops.gemv(cur, self.head_pk, st["logits"], N=self.V, K=self.D, nw=8, norm_w=self.norm, col_tiles=hct, **red, **pk)

    def _forward(self, st):
        if self.decode_mode == "v2" and st["rows"] <= 16:  # (17 .. 64 rows on this 5-launch layer: measured -4.5 % at 32 rows, +2.4 % at 64 -- profiles/r06_wide_rows_decode_ab.jsonl -- not adopted)
            return self._forward_decode_v2(st)
        self._forward_decode(st)

    def _decode_step(self, st):
        if self._use_c_step(st):
            return self._decode_step_c(st)
        self._forward(st)
        self._sample(st)

    def _use_c_step(self, st):
        return self.c_step and self.decode_mode == "v2" and st["rows"] <= 16 and self.tune["d_ks2"] in (1, 2, 4)

    def _c_loop(self, st):
        """The cbx_t3_loop_t of this state's current geometry (include/cbx.h: the decode step captured in a hipGraph by the LIBRARY, replayed by
        cbx_t3_loop_run), created on first use; kept beside the step descriptor and dropped / cached with it (apply_variant)."""
        import ctypes
        from ._lib import check, lib
        self._c_step_desc(st)
        if len(st["cstep"]) == 3:
            h = ctypes.c_void_p()
            torch.cuda.synchronize()
            check(lib.cbx_t3_loop_create(ctypes.byref(st["cstep"][0]), torch.cuda.current_stream().cuda_stream, ctypes.byref(h)), "cbx_t3_loop_create")
            st["cstep"] = st["cstep"] + (_LoopHandle(h),)
        return st["cstep"][3].h

    def _run_c_loop(self, st, n_steps, poll_every):
        import ctypes
        from ._lib import check, lib
        ran = ctypes.c_int(0)
        check(lib.cbx_t3_loop_run(self._c_loop(st), int(n_steps), int(poll_every), torch.cuda.current_stream().cuda_stream, ctypes.byref(ran)), "cbx_t3_loop_run")
        return int(ran.value)

    def _decode_step_c(self, st):
        """The same token step through the stage-level C entry point cbx_t3_decode_step (include/cbx.h): one ctypes call enqueues the
        153 launches that _forward_decode_v2 + _sample issue one by one."""
        import ctypes
        from ._lib import check, lib
        self._c_step_desc(st)
        check(lib.cbx_t3_decode_step(ctypes.byref(st["cstep"][0]), torch.cuda.current_stream().cuda_stream), "cbx_t3_decode_step")

    def _c_step_desc(self, st):
        """Build (once per state and geometry) the cbx_t3_step_t of this state: st["cstep"] = (descriptor, layers, sampler[, loop handle])."""
        import ctypes
        from ._lib import SamplerParams, T3Layer, T3Step
        if "cstep" not in st:
            p = lambda t: t.data_ptr()
            ws, tn = st["dws"], self.tune
            layers = (T3Layer * self.L)()
            qtc, odtc = self._tiles()
            for i, lw in enumerate(self.layers):
                layers[i].ln1, layers[i].ln2 = p(lw["ln1"]), p(lw["ln2"])
                layers[i].wqkv, layers[i].wgu = p(self._image(lw, "wqkv", qtc)), p(lw["wgu_pk"])
                layers[i].wo, layers[i].wd = p(self._image(lw, "wo", odtc)), p(self._image(lw, "wd", odtc))
            sp = SamplerParams()
            for k, v in dict(logits=st["logits"], ld=st["logits"].stride(0), V=self.V, B=st["B"], cfg=1, order=0, eos_token=STOP_SPEECH,
                             dev_params=st["samp_dev"], seen=st["seen"], uniforms=st["uniforms"], max_steps=st["max_steps"], step=st["step"],
                             out_tokens=st["out_tokens"], done=st["done"], n_generated=st["n_generated"], next_ids=st["next_ids"],
                             next_pos_ids=st["next_pos_ids"], positions=st["positions"], ctx_lens=st["ctx_lens"]).items():
                setattr(sp, k, v.data_ptr() if torch.is_tensor(v) else v)
            d = T3Step()
            d.n_layers, d.rows, d.dim, d.ffn, d.n_heads, d.vocab = self.L, st["rows"], self.D, self.F, self.H, self.V
            d.o_nw, d.gu_nw, d.d_nw, d.d_ksplit, d.eps, d.attn_scale = tn["o_nw2"], tn["gu_nw"], tn["d_nw2"], tn["d_ks2"], 1e-5, 0.125
            d.half_tiles, d.qkv_tile = (0 if odtc == 16 else odtc), (0 if qtc == 16 else qtc)
            d.w_bf16 = int(self.head_pk.dtype == torch.bfloat16)
            d.layers = layers
            d.speech_emb, d.speech_pos, d.final_norm, d.head = p(self.speech_emb), p(self.speech_pos), p(self.norm), p(self.head_pk)
            d.cos_t, d.sin_t, d.kc, d.vc = p(self.cos), p(self.sin), p(st["kc"]), p(st["vc"])
            d.kv_row_stride, d.kv_head_stride = st["kc"].stride(1), st["kc"].stride(2)
            d.next_ids, d.next_pos_ids, d.positions = p(st["next_ids"]), p(st["next_pos_ids"]), p(st["positions"])
            d.x_a, d.x_b, d.qkv, d.att, d.g, d.pd = p(ws["x_pk"]), p(ws["x2_pk"]), p(ws["qkv"]), p(ws["att_pk"]), p(ws["g_pk"]), p(ws["pd_pk"])
            d.logits, d.ld_logits, d.sampler = p(st["logits"]), st["logits"].stride(0), ctypes.pointer(sp)
            da = st["da"]  # the step's own attention geometry + split-context workspace, GEMV flags (ABI v10: nothing process-wide)
            d.da_unroll, d.da_pipeline, d.da_split_min, d.gemv_flags = da.unroll, da.pipeline, da.split_min, self._gf()
            qks = self._qks()
            if qks > 1:
                d.qkv, d.qkv_ksplit, d.qkv_ct, d.qkv_ssq = p(ws["qkv_parts"]), qks, int(tn.get("qkv_ct") or 3), p(ws["qkv_ssq"])
            d.head_ct = self._hct()
            d.da_ws, d.da_cnt, d.da_pairs = ops._p(da.ws), ops._p(da.cnt), (da.max_pairs if da.ws is not None else 0)  # None: >= 128 (row, head) pairs never split
            st["cstep"] = (d, layers, sp)  # keep the host structures alive

    def _sample(self, st):
        # the sampling parameters are read from device memory (st["samp_dev"], one row per utterance): a request with other settings
        # replays the SAME captured decode graph
        ops.t3_sample(logits=st["logits"], ld=st["logits"].stride(0), V=self.V, B=st["B"], cfg=1, order=0, eos_token=STOP_SPEECH,
                      dev_params=st["samp_dev"], seen=st["seen"],
                      uniforms=st["uniforms"], max_steps=st["max_steps"], step=st["step"], out_tokens=st["out_tokens"],
                      done=st["done"], n_generated=st["n_generated"], next_ids=st["next_ids"], next_pos_ids=st["next_pos_ids"],
                      positions=st["positions"], ctx_lens=st["ctx_lens"])

    # ------------------------------------------------------------------ state / workspaces
    def release_state(self, slot):
        """Drop every decode state (KV cache, workspaces, captured graph) of `slot`: measure_decode / probe_decode park theirs in slot 7
        (≈ 2.6 GB at B = 8, L = 30), which a serving engine must get back after an in-process autotune."""
        for k in [k for k in self._state if k[3] == slot]:
            del self._state[k]

    def _get_state(self, B, max_ctx, max_steps, slot=0):
        key = (B, max_ctx, max_steps, slot)
        if key in self._state:
            return self._state[key]
        for k in [k for k in self._state if k[3] == slot]:  # one live configuration per slot (the KV cache dominates memory)
            del self._state[k]
        dev, rows = self.dev, 2 * B
        f = lambda *s: torch.empty(*s, device=dev)
        i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=dev)
        st = dict(B=B, rows=rows, max_ctx=max_ctx, max_steps=max_steps,
                  kc=torch.zeros(self.L, rows, self.H, max_ctx, 64, device=dev),
                  vc=torch.zeros(self.L, rows, self.H, max_ctx, 64, device=dev),
                  logits=f(rows, self.V), seen=torch.zeros(B, self.V, dtype=torch.uint8, device=dev),
                  uniforms=f(B, max_steps), step=i32(B), out_tokens=torch.zeros(B, max_steps, dtype=torch.int64, device=dev),
                  done=i32(B), n_generated=i32(B), next_ids=torch.zeros(rows, dtype=torch.int64, device=dev),
                  next_pos_ids=i32(rows), positions=i32(rows), ctx_lens=i32(rows),
                  dws=dict(x=f(rows, self.D), h=f(rows, self.D), qkv=f(rows, 3 * self.D), att=f(rows, self.D), g=f(rows, self.F),
                           po=f(8, rows, self.D), pd=f(8, rows, self.D),
                           # packed operand images of the v2 decode path (rows padded to whole 16-row tiles, pad rows stay 0)
                           x_pk=torch.zeros((rows + 15) // 16 * 16, self.D, device=dev),
                           att_pk=torch.zeros((rows + 15) // 16 * 16, self.D, device=dev),
                           x2_pk=torch.zeros((rows + 15) // 16 * 16, self.D, device=dev),
                           pd_pk=torch.zeros(4, (rows + 15) // 16 * 16, self.D, device=dev),
                           g_pk=torch.zeros((rows + 15) // 16 * 16, self.F, device=dev),
                           # split-K partial sums of the q/k/v projection + their sums of squares (tune qkv_ks, ABI v11)
                           qkv_parts=torch.zeros(4, rows, 3 * self.D, device=dev), qkv_ssq=torch.zeros(4, 16, device=dev)),
                  graph=None, samp_dev=torch.zeros(B, 8, device=dev),
                  # geometry + split-context workspace of this state's attention launches (a state = one stream of launches; two slots may be in flight)
                  da=ops.DecodeAttnGeom(dev, split=rows * self.H < 128))
        self._sync_geom(st)
        self._state[key] = st
        return st

    @ops.on_device
    def advance(self, handle, n_steps):
        """Enqueue up to `n_steps` further decode steps of an async generate() (replays of its captured graph; finished utterances are
        no-ops inside the sampler).  Returns the number of steps enqueued."""
        st = handle["st"]
        n = max(0, min(int(n_steps), handle["max_new_tokens"] - handle["next_i"]))
        if n and self.c_loop and self._use_c_step(st) and self.dev.type == "cuda" and st["graph"] is None:
            self._run_c_loop(st, n, 0)
            handle["next_i"] += n
            return n
        for _ in range(n):
            if st["graph"] is not None:
                st["graph"].replay()
            else:
                self._decode_step(st)
        handle["next_i"] += n
        return n

    @ops.on_device
    def peek(self, handle):
        """Tokens sampled so far (synchronises with the launch stream): (list of B 1-D LongTensors, list of B done flags)."""
        st, B = handle["st"], handle["B"]
        # (order matters when the decode is still running on ANOTHER stream -- engine.synthesize_stream(overlap=True): the sampler writes token, count, then
        # the done flag, so a done flag read FIRST implies the count read after it is final; a count that is still growing just means "not final yet")
        done = st["done"].tolist()
        n = st["n_generated"].tolist()
        toks = st["out_tokens"].cpu()
        return [toks[b, : n[b]].clone() for b in range(B)], [bool(d) for d in done]

    @ops.on_device
    def collect(self, handle):
        """Fetch the tokens of an (async) generate() call.  Must run on the stream the call was enqueued on."""
        st, B = handle["st"], handle["B"]
        n = st["n_generated"].tolist()
        toks = st["out_tokens"].cpu()
        return [toks[b, : n[b]].clone() for b in range(B)]

    # ------------------------------------------------------------------ T3.inference
    @ops.on_device
    @torch.inference_mode()
    def generate(self, conds, text_tokens, max_new_tokens=1000, temperature=0.8, top_p=1.0, min_p=0.05,
                 repetition_penalty=1.2, cfg_weight=0.5, uniforms=None, ban_eos=False, ban_from=0, use_graph=True, poll_every=16,
                 return_prefill_logits=False, debug_logits=False, async_mode=False, slot=0, run_steps=None):
        """conds: one T3 cond dict (shared voice) or a list of B; text_tokens: list of B 1-D LongTensors that already
        carry SOT/EOT (mtl_tts.py:319-322).  Returns a list of B 1-D LongTensors (EOS included if it was sampled).
        `slot` selects an independent set of workspaces / KV cache / decode graph (pipelined serving keeps two alive).
        Chunked use (streaming synthesis): `async_mode=True, run_steps=k` samples only the first k tokens and returns a handle;
        `advance(handle, n)` enqueues n more decode steps and `peek(handle)` fetches the tokens sampled so far."""
        dev, B = self.dev, len(text_tokens)
        assert B >= 1, "empty batch"
        if uniforms is not None:
            uniforms = torch.as_tensor(uniforms, dtype=torch.float32)
            assert uniforms.numel() % B == 0 and uniforms.numel() // B >= max_new_tokens, \
                f"uniforms must hold at least max_new_tokens={max_new_tokens} draws per utterance"
            uniforms = uniforms.view(B, -1)
        if B > self.MAX_BATCH:  # rows = 2B feeds the decode GEMV (M <= 64): larger batches run as consecutive sub-batches
            assert not (async_mode or debug_logits or return_prefill_logits), "sub-batching is only defined for the plain token path"
            out = []
            for lo in range(0, B, self.MAX_BATCH):
                hi = min(B, lo + self.MAX_BATCH)
                out += self.generate(conds if isinstance(conds, dict) else conds[lo:hi], text_tokens[lo:hi], max_new_tokens=max_new_tokens,
                                     temperature=temperature, top_p=top_p, min_p=min_p, repetition_penalty=repetition_penalty,
                                     cfg_weight=cfg_weight, uniforms=None if uniforms is None else uniforms[lo:hi], ban_eos=ban_eos,
                                     ban_from=ban_from, use_graph=use_graph, poll_every=poll_every, slot=slot)
            return out
        rows = 2 * B
        prec = self.tune.get("prefill_prec") or 0
        pre = self._voice_prefix(conds) if prec in (0, 1) else None  # the cached conditioning prefix of this voice: prefill the text positions only
        if pre is not None:
            ce = None
        elif isinstance(conds, dict):
            ce = [self.cond_embeds(conds)] * B
        else:
            ce = [self.cond_embeds(c) for c in conds]
        tl = [int(t.numel()) for t in text_tokens]
        s0 = [34 + n + 2 for n in tl]
        S = max(s0)
        max_ctx = (S + max_new_tokens + 63) // 64 * 64
        assert max_ctx <= self.max_pos, "context exceeds the RoPE table"
        st = self._get_state(B, max_ctx, max_new_tokens, slot)
        self._prepare_tune()
        # {cfg_weight, temperature, min_p, top_p, rep_penalty, top_k, ban_token, ban_from} per utterance (cbx_sampler_t.dev_params)
        st["samp_dev"].copy_(torch.tensor([float(cfg_weight), float(temperature), float(min_p), float(top_p), float(repetition_penalty), 0.0,
                                           float(STOP_SPEECH if ban_eos else -1), float(ban_from)]).repeat(B, 1), non_blocking=True)
        for k in ("seen", "step", "done", "n_generated", "out_tokens"):
            st[k].zero_()
        st["seen"][:, START_SPEECH] = 1
        if uniforms is None:
            st["uniforms"].uniform_()
        else:
            st["uniforms"].copy_(torch.as_tensor(uniforms, dtype=torch.float32).view(B, -1)[:, :max_new_tokens])

        # ---- prefill embeddings (prepare_input_embeds + second BOS, t3.py:102-130,305-313)
        P0 = 34 if pre is not None else 0  # prompt positions that are NOT computed in this call
        Sx = S - P0
        x = torch.zeros(rows, Sx, self.D, device=dev)
        bos = torch.full((2,), START_SPEECH, dtype=torch.int64, device=dev)
        zero2 = torch.zeros(2, dtype=torch.int32, device=dev)
        for b in range(B):
            ids = text_tokens[b].to(dev).long().view(-1)
            pos = torch.arange(tl[b], dtype=torch.int32, device=dev)
            for r, scale in ((b, 1.0), (B + b, 0.0)):
                if pre is None:
                    x[r, :34] = ce[b]
                ops.embed(ids, self.text_emb, x[r, 34 - P0:34 - P0 + tl[b]], table2=self.text_pos, ids2=pos, scale=scale)
                ops.embed(bos, self.speech_emb, x[r, 34 - P0 + tl[b]:s0[b] - P0], table2=self.speech_pos, ids2=zero2)
        xf = x.view(rows * Sx, self.D)
        pws = dict(h=torch.empty(rows * Sx, self.D, device=dev), qkv=torch.empty(rows * Sx, 3 * self.D, device=dev),
                   att=torch.empty(rows * Sx, self.D, device=dev), g=torch.empty(rows * Sx, self.F, device=dev))
        pos = torch.arange(P0, S, dtype=torch.int32, device=dev).repeat(rows)
        crow = torch.arange(rows, dtype=torch.int32, device=dev).repeat_interleave(Sx)
        # prefill_prec (tune / CBX_T3_TUNE="prefill_prec=6", opt-in, untimed): the prefill's plain projections (q/k/v, o, down) and its attention
        # on the bf16x6 split kernels (24 significand bits, fp32 range, accumulation error below the exact MFMA's own: DESIGN.md section 1)
        # instead of the exact fp32 MFMA; gate|up (SwiGLU epilogue) and every decode step stay exact
        if pre is not None:
            self._paste_voice_prefix(pre, st, rows)
            with ops.gemm_precision(prec):
                for i, lw in enumerate(self.layers):
                    self._layer_prefill_text(lw, xf, pws, Sx, S, rows, st["kc"][i], st["vc"][i], pos, crow)
        elif self.c_step and not ops.TIMER:  # the same launches through the stage-level C entry point cbx_t3_prefill (one ctypes call instead of 9 per layer)
            self._prefill_c(xf, pws, S, rows, st, pos, crow)
        else:
            with ops.gemm_precision(prec):
                for i, lw in enumerate(self.layers):
                    self._layer_prefill(lw, xf, pws, S, rows, st["kc"][i], st["vc"][i], pos, crow)
        if pre is None and prec in (0, 1):
            self._keep_voice_prefix(conds, st)
        last = torch.tensor([r * Sx + s0[r % B] - P0 - 1 for r in range(rows)], device=dev)
        hl = xf.index_select(0, last).contiguous()
        ops.layernorm(hl, self.norm, None, st["dws"]["h"], 1e-5, rms=True)
        ops.linear(st["dws"]["h"], self.head, st["logits"])
        prefill_logits = st["logits"].clone() if return_prefill_logits else None
        del x, xf, pws

        # ---- decode loop: positions / ctx_lens hold the state BEFORE the first sample (incremented by the sampler)
        s0t = torch.tensor(s0 + s0, dtype=torch.int32, device=dev)
        st["positions"].copy_(s0t - 1)
        st["ctx_lens"].copy_(s0t)
        step_logits = [st["logits"].clone()] if debug_logits else None
        self._sample(st)
        if debug_logits:
            use_graph = False
        c_loop = bool(use_graph and self.c_loop and self._use_c_step(st) and self.dev.type == "cuda" and max_new_tokens > 1)
        if c_loop:
            self._c_loop(st)  # (captures the step on first use: before the timed events below)
        if use_graph and not c_loop and st["graph"] is None and max_new_tokens > 1:
            torch.cuda.synchronize()
            saved = {k: st[k].clone() for k in ("seen", "step", "done", "n_generated", "out_tokens", "next_ids",
                                                "next_pos_ids", "positions", "ctx_lens", "logits")}
            g = torch.cuda.CUDAGraph()
            # thread_local: a capture on the T3-enqueue thread of synthesize_pipelined must not make the OTHER host thread's allocations / launches
            # (flow + vocoder of the previous batch, on their own stream) illegal
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._decode_step(st)
            for k, v in saved.items():  # the capture warm-up must not leak into the real sequence
                st[k].copy_(v)
            st["graph"] = g
        ev = None
        if self.time_decode:  # two HIP events around the decode loop on its launch stream (bench.py: in-run decode-step roofline)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        n_replays = 0
        last = max_new_tokens if run_steps is None else max(1, min(max_new_tokens, int(run_steps)))
        if c_loop and last > 1:  # the token loop in C: one ctypes call; EOS polled every `poll_every` steps unless the caller wants no synchronisation
            n_replays = self._run_c_loop(st, last - 1, 0 if (ban_eos or async_mode) else poll_every)
            if async_mode:
                last = 1 + n_replays
        for i in range(1, 1 if c_loop else last):
            n_replays += 1
            if use_graph and st["graph"] is not None:
                st["graph"].replay()
            elif debug_logits:  # forward and sampler split so that the raw logits of every step can be inspected
                self._forward(st)
                step_logits.append(st["logits"].clone())
                self._sample(st)
            else:
                self._decode_step(st)
            if not ban_eos and not async_mode and (i % poll_every == 0) and bool(st["done"].all()):
                break
        if ev is not None:
            ev[1].record()
            self.decode_events.append((ev[0], ev[1], n_replays, list(s0), rows))
        if async_mode:  # everything is enqueued on the current stream; no host synchronisation happened
            return dict(st=st, B=B, next_i=last, max_new_tokens=max_new_tokens)
        out = self.collect(dict(st=st, B=B))
        if debug_logits:
            return out, torch.stack(step_logits)  # (steps, 2B, V)
        return (out, prefill_logits) if return_prefill_logits else out
