"""ABI v9 decode variants (-m gpu): 12- / 4-column GEMV tiles, the down projection without partial images, the software-pipelined decode
attention, and the T3 engines running on them.

These variants were written at the END of round 3 without GPU access: they are verified on the SIMT emulator (tests/test_simt_kernels.py
runs the same bodies on the CPU) but had not yet run on an MI355X when they were committed.  The file name sorts last on purpose: under
`pytest -x` a surprise here cannot hide the results of the established suites.  Every variant is opt-in; the shipped defaults are covered
by the other files.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_ops_gpu import _close, _r, _unpack_operand

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
SAMP = dict(temperature=0.8, top_p=1.0, min_p=0.05, repetition_penalty=1.2, cfg_weight=0.5)


@pytest.mark.parametrize("M,N,K,ks,nw,tile,mode", [(16, 3072, 1024, 1, 8, 12, "rms_np2"), (16, 3072, 1024, 1, 8, 12, "rms"), (16, 1024, 4096, 1, 16, 4, "res"),
                                                    (16, 1024, 1024, 1, 8, 4, "res"), (9, 1024, 4096, 1, 8, 4, "plain"), (5, 40, 256, 2, 4, 12, "plain"),
                                                    (16, 1024, 4096, 1, 8, 4, "bf16")])
def test_gemv_narrow_tiles(dev, M, N, K, ks, nw, tile, mode):
    """12- and 4-column output tiles (ABI v9: cbx_gemv_t.half_tile = 12 / 4 + the matching packed image): q/k/v (N = 3072) resp. the o / down
    projections (N = 1024) on exactly 256 workgroups.  Same per-column arithmetic as the 16-column form: results bit for bit, in the plain,
    RMSNorm-folded, partial-sum-operand, residual-epilogue and bf16-weight forms."""
    from chatterbox_amd import ops
    x, w, r = _r((M, K), 1), _r((N, K), 2, 1 / math.sqrt(K)), _r((M, (N + 31) // 32 * 32), 3)
    nwt = 1 + 0.1 * _r((K,), 4)
    xp = ops.pack_gemv_weight(x.to(dev))
    bf = mode == "bf16"
    w16, wn = ops.pack_gemv_weight(w.to(dev), bf16=bf), ops.pack_gemv_weight(w.to(dev), half_tile=tile, bf16=bf)
    assert wn.shape[0] == (N + tile - 1) // tile * tile
    shape = (ks, M, N) if ks > 1 else (M, N)
    kw = dict(N=N, M=M, K=K, ksplit=ks, nw=nw, w_packed=True, x_packed=True)
    tol = 3e-5 * max(1.0, math.sqrt(K / 256))
    if mode == "res":
        ra, rb = ops.pack_gemv_weight(r.to(dev)), ops.pack_gemv_weight(r.to(dev))
        ops.gemv(xp, w16, ra, res=ra, out_packed=True, **kw)
        ops.gemv(xp, wn, rb, res=rb, out_packed=True, half_tile=tile, **kw)
        assert torch.equal(ra, rb)
        _close(_unpack_operand(rb, M, N), r[:, :N] + F.linear(x, w), tol, f"{tile}-column gemv + residual")
        return
    extra = {}
    ref = F.linear(x, w)
    if mode.startswith("rms"):
        extra = dict(norm_w=nwt.to(dev))
        xs = x
        if mode == "rms_np2":
            parts = _r((2, M, K), 5, 0.3)
            pp = torch.stack([ops.pack_gemv_weight(parts[j].to(dev)) for j in range(2)])
            extra.update(xpart=pp, x_out=torch.zeros_like(xp))
            xs = x + parts[0] + parts[1]
        ref = F.linear(xs * torch.rsqrt((xs * xs).mean(-1, keepdim=True) + 1e-5) * nwt, w)
    if bf:
        ref = F.linear(x, w.bfloat16().float())
    a, b = torch.zeros(shape, device=dev), torch.zeros(shape, device=dev)
    ops.gemv(xp, w16, a, **kw, **extra)
    if "x_out" in extra:
        extra["x_out"] = torch.zeros_like(xp)
    ops.gemv(xp, wn, b, half_tile=tile, **kw, **extra)
    assert torch.equal(a, b), f"{tile}-column tiles differ from the 16-column form"
    _close(b.sum(0) if ks > 1 else b, ref, 2 * tol, f"{tile}-column gemv ({mode})")
    if "x_out" in extra:
        _close(_unpack_operand(extra["x_out"], M, K), xs, 1e-6, "x_out = x + partial images")


@pytest.mark.parametrize("rows,H,rope,split_min", [(3, 16, True, 512), (1, 12, False, 1), (16, 16, True, 512)])
def test_decode_attn_rope_pipelined_equals_plain(dev, rows, H, rope, split_min):
    """cbx_set_decode_attn_pipeline(1) (ABI v9: the next step's K / V rows are requested before the current step is multiplied, two register
    sets) against the plain form: same arithmetic in the same order, so outputs and appended cache rows are equal bit for bit -- contexts of
    1 .. 70 (fewer rows than one step, exactly one step, the ragged tail of the second register set), a few long ones, 4 and 8 rows per
    lane group and step, the one-workgroup and the split-context grids."""
    from chatterbox_amd import ops
    from oracle import ref_torch as O
    ops.ensure_decode_attn_workspace(dev)
    maxp = 640
    kc0, vc0 = _r((rows, H, maxp, 64), 1), _r((rows, H, maxp, 64), 2)
    cos, sin = O.rope_cos_sin(torch.arange(maxp), O.llama3_inv_freq())
    cd, sd_ = (cos.to(dev), sin.to(dev)) if rope else (None, None)
    try:
        ops.lib.cbx_set_decode_attn_split_min(split_min)
        for u in (4, 8):
            ops.lib.cbx_set_decode_attn_unroll(u)
            for n in list(range(0, 70, 1 if rows < 16 else 9)) + [127, 128, 129, 255, 300, 639]:
                pos = torch.tensor([(n + 37 * r) % maxp for r in range(rows)], dtype=torch.int32)
                qkv = _r((rows, 3 * H * 64), 100 + n)
                res = []
                # 2 / 3: non-temporal K / V loads, plain / pipelined; 4 .. 7: the same four with the speculative first step (positions 0 .. 63 requested
                # before positions[row] has arrived) -- the cache past each row's context is poisoned with NaN: whatever is read there must not matter
                for pipe in (0, 1) + ((2, 3, 4, 5, 6, 7) if u == 4 else ()):
                    ops.lib.cbx_set_decode_attn_pipeline(pipe)
                    kc, vc, out = kc0.clone(), vc0.clone(), torch.zeros(rows, H * 64, device=dev)
                    for r in range(rows):
                        kc[r, :, int(pos[r]):], vc[r, :, int(pos[r]):] = float("nan"), float("nan")
                    kc, vc = kc.to(dev), vc.to(dev)
                    ops.decode_attn_rope(qkv.to(dev), pos.to(dev), cd, sd_, kc, vc, out, 0.125)
                    kc, vc = kc.cpu(), vc.cpu()
                    assert all(bool(torch.isnan(c[r, :, int(pos[r]) + 1:]).all()) and bool(torch.isfinite(c[r, :, : int(pos[r]) + 1]).all())
                               for c in (kc, vc) for r in range(rows)), "exactly one row appended per (row, head)"
                    res.append((out.cpu(), torch.nan_to_num(kc), torch.nan_to_num(vc)))
                for other in res[1:]:
                    for a, b, what in zip(res[0], other, ("output", "k cache", "v cache")):
                        assert torch.equal(a, b) and bool(torch.isfinite(a).all()), f"pipelined / non-temporal / speculative decode attention differs in the {what} (U = {u}, context {n + 1})"
                if n in (0, 63, 64, 65, 300):  # and the result itself against torch
                    q, k, v = (qkv.view(rows, 3, H, 64)[:, i] for i in range(3))
                    if rope:
                        c, s = cos[pos.long()][:, None], sin[pos.long()][:, None]
                        q, k = q * c + O._rot_half(q) * s, k * c + O._rot_half(k) * s
                    for r in range(rows):
                        m = int(pos[r])
                        kk, vv = torch.cat([kc0[r, :, :m], k[r][:, None]], 1), torch.cat([vc0[r, :, :m], v[r][:, None]], 1)
                        _close(res[1][0][r].view(H, 64), F.scaled_dot_product_attention(q[r].reshape(H, 1, 64), kk, vv)[:, 0], 2e-5, f"pipelined, ctx {m + 1}")
    finally:
        ops.lib.cbx_set_decode_attn_pipeline(0)
        ops.lib.cbx_set_decode_attn_unroll(4)
        ops.lib.cbx_set_decode_attn_split_min(512)


@pytest.mark.parametrize("tune", ["qkv_tc=12,od_tc=4,d_ks2=1", "qkv_tc=12", "od_tc=4,d_ks2=1,d_nw2=8", "pair_ogu=1", "pair_ogu=1,pair_dq=1,qkv_tc=12,od_tc=4,d_ks2=1,d_nw2=8", "chain=1,qkv_tc=12,od_tc=4,d_ks2=1,d_nw2=8"])
def test_t3_decode_tile_variants_sample_the_reference_tokens(dev, tune, monkeypatch):
    """The round-3 decode geometries (CBX_T3_TUNE: 12-column q/k/v tiles, 4-column o / down tiles, down projection without partial images)
    against the golden tokens of the reference (t3_l2: 2 layers, 64 steps) on the hipGraph + C-step path, and against the default geometry
    on a ragged batch."""
    from chatterbox_amd import synth
    from chatterbox_amd.t3 import T3Engine
    g = np.load(os.path.join(GOLD, "t3_l2.npz"))
    steps, n_text = int(g["steps"]), int(g["n_text"])
    sd = synth.t3_state_dict(2, 0)
    monkeypatch.setenv("CBX_T3_TUNE", tune)
    eng = T3Engine(sd, dev)
    assert all(eng.tune[kv.split("=")[0]] == int(kv.split("=")[1]) for kv in tune.split(","))
    u = torch.from_numpy(g["uniforms"])[None]
    toks = eng.generate(synth.t3_cond(), [synth.text_tokens(n_text)], max_new_tokens=steps, uniforms=u, ban_eos=True, **SAMP)
    assert toks[0].tolist() == g["tokens"].tolist()
    monkeypatch.delenv("CBX_T3_TUNE")
    tt = [synth.text_tokens(12, seed=1), synth.text_tokens(20, seed=2), synth.text_tokens(7, seed=3)]
    u3 = synth.rand((3, 20), seed=3)
    kw = dict(max_new_tokens=20, uniforms=u3, ban_eos=True, **SAMP)
    ra, rb = eng.generate(synth.t3_cond(), tt, **kw), T3Engine(sd, dev).generate(synth.t3_cond(), tt, **kw)
    assert [t.tolist() for t in ra] == [t.tolist() for t in rb]
    if "pair_" in tune or "chain" in tune:  # cbx_gemv_pair_f32 / cbx_gemv_chain_f32 inside the replayed graph: counters re-armed after every launch, no consumer ever timed out
        assert not any(bool(st["dws"]["pair_ws"].any()) for st in eng._state.values())




def test_stream_with_growing_chunks_matches_the_oracle_schedule(dev):
    """synthesize_stream(chunk_growth=2): rounds at 9 -> 13 -> 21 -> 30 tokens (engine.stream_token_schedule), against the same schedule
    restated on the CPU oracle (tests/test_stream_gpu.py::_oracle_stream) and, beyond the vocoder's receptive field, the one-shot synthesis."""
    from chatterbox_amd.engine import stream_token_schedule
    from oracle import ref_torch as O
    from test_stream_gpu import _oracle_stream, _setup
    N, P, first, chunk, look, fade = 30, 8, 6, 4, 3, 240
    assert stream_token_schedule(N, first, chunk, look, 2.0) == [9, 13, 21, 30]
    eng, s3_sd, texts, cond, ref, z, phase, noise, kw = _setup(dev, N, P)
    rounds = list(eng.synthesize_stream(texts, cond, ref, first_chunk=first, chunk=chunk, chunk_growth=2.0, lookahead=look, fade=fade, **kw))
    assert [r["n_tokens"][0] for r in rounds] == [9, 13, 21, 30] and rounds[-1]["final"] == [True, True]
    full, toks = eng.synthesize(texts, cond, ref, drop_last_token=True, **kw)
    for b in range(2):
        streamed = torch.cat([r["wavs"][b] for r in rounds])
        assert streamed.numel() == (N - 1) * 960 == full[b].numel()
        pieces = _oracle_stream(O, s3_sd, toks[b], ref, z[b:b + 1], phase[b:b + 1], noise[b:b + 1], first, chunk, look, fade, 3, growth=2.0)
        assert [p.numel() for p in pieces] == [r["wavs"][b].numel() for r in rounds]
        rmse = (streamed - torch.cat(pieces)).pow(2).mean().sqrt().item()
        assert rmse <= 2e-3, f"utt {b}: streamed vs oracle-streamed RMSE {rmse:.3e}"


@pytest.mark.parametrize("M,N,K,ks,tile,res", [(16, 1024, 4096, 1, 4, True), (16, 1024, 4096, 2, 8, False), (9, 64, 2048, 1, 0, False)])
def test_gemv_deep_batches_equal_plain(dev, M, N, K, ks, tile, res):
    """cbx_set_gemv_deep_batches(1): an 8-wave plain packed GEMV whose waves own >= 256 of K requests 8 K blocks per load batch instead of 4
    (the partial-free down projection: K = 4096 over 8 waves).  Same blocks in the same order: bit-identical."""
    from chatterbox_amd import ops
    x, w, r = _r((M, K), 1), _r((N, K), 2, 1 / math.sqrt(K)), _r((M, (N + 31) // 32 * 32), 3)
    xp, wp = ops.pack_gemv_weight(x.to(dev)), ops.pack_gemv_weight(w.to(dev), half_tile=tile)
    kw = dict(N=N, M=M, K=K, ksplit=ks, nw=8, w_packed=True, x_packed=True, half_tile=tile)
    outs = []
    try:
        for deep in (0, 1):
            ops.lib.cbx_set_gemv_deep_batches(deep)
            if res:
                o = ops.pack_gemv_weight(r.to(dev))
                ops.gemv(xp, wp, o, res=o, out_packed=True, **kw)
            else:
                o = torch.zeros((ks, M, N) if ks > 1 else (M, N), device=dev)
                ops.gemv(xp, wp, o, **kw)
            outs.append(o.cpu())
    finally:
        ops.lib.cbx_set_gemv_deep_batches(0)
    assert torch.equal(outs[0], outs[1])
    got = _unpack_operand(outs[1], M, N) - r[:, :N] if res else (outs[1].sum(0) if ks > 1 else outs[1])
    _close(got, F.linear(x, w), 6e-5 * max(1.0, math.sqrt(K / 256)), "deep-batch gemv")


def test_t3_prefill_on_the_bf16x6_kernels_samples_the_reference_tokens(dev, monkeypatch):
    """CBX_T3_TUNE="prefill_prec=6" (opt-in): the prefill's q/k/v, o and down projections and its attention on the bf16x6 split kernels
    (24 significand bits, fp32 exponent range) instead of the exact fp32 MFMA; decode stays exact.  Golden tokens of the reference (t3_l2,
    64 steps) and teacher-forced logits within the usual 1e-3."""
    from chatterbox_amd import synth
    from chatterbox_amd.t3 import T3Engine
    g = np.load(os.path.join(GOLD, "t3_l2.npz"))
    steps, n_text = int(g["steps"]), int(g["n_text"])
    monkeypatch.setenv("CBX_T3_TUNE", "prefill_prec=6")
    eng = T3Engine(synth.t3_state_dict(2, 0), dev)
    assert eng.tune["prefill_prec"] == 6
    u = torch.from_numpy(g["uniforms"])[None]
    toks, logits = eng.generate(synth.t3_cond(), [synth.text_tokens(n_text)], max_new_tokens=steps, uniforms=u, ban_eos=True, debug_logits=True, **SAMP)
    idx = torch.from_numpy(g["logit_idx"]).long()
    err = (logits.cpu()[:, :, idx] - torch.from_numpy(g["logits_sub"])).abs().max().item()
    assert err <= 1e-3, f"teacher-forced logits max-abs {err:.3e}"
    assert toks[0].tolist() == g["tokens"].tolist()


def test_decode_autotuner_adopts_only_bit_identical_geometries(dev):
    """T3Engine.autotune (chatterbox_amd/autotune.py): the candidates are timed in a child process on a 2-layer model of the real width; whatever
    is adopted samples the reference's golden tokens (t3_l2: 64 steps) through the hipGraph path, and every candidate row carries either a time
    and an identity verdict or an error -- a reordering candidate is never the adopted one."""
    from chatterbox_amd import autotune as at, ops, synth
    from chatterbox_amd.t3 import T3Engine
    g = np.load(os.path.join(GOLD, "t3_l2.npz"))
    steps, n_text = int(g["steps"]), int(g["n_text"])
    eng = T3Engine(synth.t3_state_dict(2, 0), dev)
    try:
        rep = eng.autotune(B=8, ctx=128, steps=16, reps=2, timeout=240.0)
        assert "error" not in rep, rep
        rows = [r for r in rep["candidates"] if "variant" in r]
        assert len(rows) >= len(at.TILE_VARIANTS) + len(at.ATTN_VARIANTS) and all(("ms_per_token" in r) != ("error" in r) for r in rows), rows
        chained = [r for r in rows if r["variant"].get("chain")]
        assert len(chained) == len(at.CHAIN_VARIANTS) and all("twin_identical" in r or "error" in r for r in chained), chained
        assert all(r.get("twin_identical") for r in chained), f"a chained launch ended its measured run in another state than its separate launches: {chained}"
        narrow = [r for r in rows if r["variant"] in (dict(qkv_tc=12), dict(od_tc=4), dict(qkv_tc=12, od_tc=4))]
        assert all(r.get("identical") for r in narrow), narrow  # same per-column arithmetic as the 16- / 8-column tiles
        best = rep["best"]
        if best:
            assert next(r for r in rows if r["variant"] == best)["identical"] and not best.get("chain")  # (the partial-free geometry reorders)
            assert all(eng.tune[k] == v for k, v in at.split_variant(best)[0].items())
        u = torch.from_numpy(g["uniforms"])[None]
        toks = eng.generate(synth.t3_cond(), [synth.text_tokens(n_text)], max_new_tokens=steps, uniforms=u, ban_eos=True, **SAMP)
        assert toks[0].tolist() == g["tokens"].tolist(), f"adopted geometry {best}"
    finally:
        eng.apply_variant(dict(T3Engine._TUNE), dict(at.LIB_KNOBS))  # the attention knobs are process-wide


_EPI_BODIES = [("test_gemv_decode", (16, 3072, 1024, 1, 8)), ("test_gemv_decode", (40, 1024, 1024, 2, 4)), ("test_gemv_decode", (6, 64, 256, 1, 4)),
               ("test_gemv_swiglu", ()), ("test_gemv_packed_rms_fused", (16, 3072, 1024, False, 8)),
               ("test_gemv_packed_residual_epilogue", (16, 1024, 1024, 16)), ("test_gemv_packed_residual_epilogue", (7, 1024, 1024, 8)),
               ("test_gemv_layernorm_fused", (5, 2304, 768, 4, False)), ("test_gemv_layernorm_fused", (16, 4096, 1024, 0, True)),
               ("test_gemv_half_tile", (16, 1024, 1024, 1, 8, True)), ("test_gemv_narrow_tiles", (16, 1024, 4096, 1, 16, 4, "res")),
               ("test_gemv_narrow_tiles", (16, 3072, 1024, 1, 8, 12, "rms_np2")), ("test_gemv_deep_batches_equal_plain", (16, 1024, 4096, 1, 4, True))]


@pytest.mark.parametrize("name,args", _EPI_BODIES, ids=[f"{n}{list(a)}" for n, a in _EPI_BODIES])
def test_gemv_epilogue_prefetch_equals_plain(dev, name, args, monkeypatch):
    """cbx_set_gemv_epilogue_prefetch(1): the residual element, the bias and the LayerNorm-fold constants of a GEMV's epilogue are requested
    with the first weight batch instead of after the reduction.  Every GEMV launch of the established bodies (bias, split-K, swiglu, RMSNorm
    and LayerNorm folds, in-place packed residual, narrow tiles, deep batches) runs twice -- knob off, knob on -- from the same memory state:
    outputs bit-identical, and the body's own comparison against torch holds with the knob on."""
    import sys
    import test_ops_gpu
    from chatterbox_amd import ops
    real, n = ops.gemv, [0]

    def both(x, w, out, **kw):
        keep = [(t, t.clone()) for t in (out, kw.get("x_out")) if t is not None]
        ops.lib.cbx_set_gemv_epilogue_prefetch(0)
        real(x, w, out, **kw)
        plain = [t.clone() for t, _ in keep]
        for t, c in keep:
            t.copy_(c)
        ops.lib.cbx_set_gemv_epilogue_prefetch(1)
        r = real(x, w, out, **kw)
        for (t, _), p in zip(keep, plain):
            assert torch.equal(t.cpu(), p.cpu()), f"{name}{list(args)}: launch {n[0]} differs with the epilogue operands prefetched"
        n[0] += 1
        return r

    monkeypatch.setattr(ops, "gemv", both)
    try:
        body = getattr(test_ops_gpu, name, None) or getattr(sys.modules[__name__], name)
        body(dev, *args)
    finally:
        ops.lib.cbx_set_gemv_epilogue_prefetch(0)
    assert n[0] >= 1


@pytest.mark.parametrize("B,T,H,lens", [(2, 150, 2, (150, 70)), (1, 64, 8, None), (3, 33, 1, (33, 1, 0)), (1, 300, 2, (257,))])
def test_flash_relpos_equals_materialised_scores(dev, B, T, H, lens):
    """cbx_flash_relpos_f32 (conformer rel-pos attention without the (T, T) / (T, 2T-1) score tensors) against fp64 torch and against the
    materialised path it replaces (bmm, bmm, softmax_relpos, bmm): ragged key lengths incl. an empty row, T across query-tile (128), key-tile
    (64) and 32-row position-block boundaries."""
    from chatterbox_amd import ops
    q4 = _r((B, T, 4, H, 64), 1, 0.5)
    pp = _r((2 * T - 1, H * 64), 2, 0.5)
    kl = None if lens is None else torch.tensor(lens, dtype=torch.int32)
    qu, qv, k, v = (q4[:, :, i].permute(0, 2, 1, 3).double() for i in range(4))  # (B, H, T, 64)
    p = pp.view(2 * T - 1, H, 64).permute(1, 0, 2).double()                     # (H, 2T-1, 64)
    ac = qu @ k.transpose(-1, -2)
    bd_full = qv @ p.transpose(-1, -2)[None]                                     # (B, H, T, 2T-1)
    idx = (T - 1 - torch.arange(T)[:, None] + torch.arange(T)[None, :])          # [i][j] -> T-1-i+j
    bd = torch.gather(bd_full, 3, idx.expand(B, H, T, T))
    sc = (ac + bd) * 0.125
    if kl is not None:
        sc = sc.masked_fill(torch.arange(T)[None, None, None, :] >= kl[:, None, None, None], float("-inf"))
    pr = torch.nan_to_num(torch.softmax(sc, -1))  # an empty row: all keys masked -> zeros
    ref = (pr @ v).permute(0, 2, 1, 3)                                           # (B, T, H, 64)

    q4d, ppd, kld = q4.to(dev), pp.to(dev), None if kl is None else kl.to(dev)
    out = torch.full((B, T, H, 64), float("nan"), device=dev)
    ops.flash_relpos(q4d, ppd, out, 0.125, key_lens=kld)
    _close(out, ref, 2e-5, "flash rel-pos attention vs fp64")

    Tp, Pp = (T + 3) // 4 * 4, (2 * T - 1 + 3) // 4 * 4
    f = lambda *s: torch.zeros(*s, device=dev)
    acd, bdd, prd, att = f(B, H, T, Tp), f(B, H, T, Pp), f(B, H, T, Tp), f(B, T, H, 64)
    ops.bmm(q4d[:, :, 0].permute(0, 2, 1, 3), q4d[:, :, 2].permute(0, 2, 1, 3), acd[..., :T])
    ops.bmm(q4d[:, :, 1].permute(0, 2, 1, 3), ppd.view(1, 2 * T - 1, H, 64).permute(0, 2, 1, 3).expand(B, H, 2 * T - 1, 64), bdd[..., : 2 * T - 1])
    ops.softmax_relpos(acd[..., :T], bdd, prd, 0.125, key_lens=kld)
    ops.bmm(prd[..., :T], q4d[:, :, 3].permute(0, 2, 1, 3), att.permute(0, 2, 1, 3), nn=True)
    _close(out, att.cpu().double(), 2e-5, "flash rel-pos attention vs the materialised path")


def test_encoder_flash_relpos_modes_match_the_materialised_encoder(dev):
    """FlowEngine.encode through cbx_flash_relpos_f32: by default ("auto") a batch whose rel-pos score tensors would exceed ENC_SCORE_BYTES takes
    the flash form (instead of being walked in row groups), CBX_ENC_FLASH=1 always does.  Ragged batch of 3 through encoder + CFM: the mel equals
    the materialised encoder's to rounding and the oracle's at the usual tolerances."""
    from chatterbox_amd import ops, synth
    from chatterbox_amd.s3gen import FlowEngine
    from oracle import ref_torch as O
    from test_models_gpu import _s3_inputs
    sd = synth.s3gen_state_dict(0, n_mid=2, n_enc=2, n_up_enc=1)
    eng = FlowEngine(sd, dev)
    P, Ns = 10, [14, 9, 3]
    ref, toks, lens = _s3_inputs(P, Ns)
    z = synth.randn((3, 80, 2 * (P + max(Ns))), seed=5).transpose(1, 2).contiguous()
    eng.ENC_FLASH = "0"
    mel = eng.inference(toks, lens, ref, z=z, n_steps=3).cpu()
    calls, real = [0], ops.flash_relpos

    def counted(*a, **k):
        calls[0] += 1
        return real(*a, **k)

    ops.flash_relpos = counted
    try:
        for mode, cap, want in (("auto", 32 << 30, 0), ("auto", 1, 3), ("1", 32 << 30, 3)):  # 2 + 1 conformer layers in this small model
            calls[0] = 0
            eng.ENC_FLASH, eng.ENC_SCORE_BYTES = mode, cap
            mel2 = eng.inference(toks, lens, ref, z=z, n_steps=3).cpu()
            assert calls[0] == want, f"CBX_ENC_FLASH={mode}, cap {cap}: {calls[0]} flash launches"
            assert (mel - mel2).abs().max() <= 2e-5, f"flash rel-pos encoder ({mode}, cap {cap}): {(mel - mel2).abs().max():.3e}"
    finally:
        ops.flash_relpos = real
    for b, n in enumerate(Ns):
        om = O.flow_inference(sd, toks[b:b + 1, :n], torch.tensor([n]), ref, z[b:b + 1, : 2 * (P + n)].transpose(1, 2), 3)
        err = (mel2[b, : 2 * n] - om[0].t()).abs()
        assert err.mean() <= 1e-4 and err.max() <= 1e-3, f"utt {b}: mel L1 {err.mean():.3e} max {err.max():.3e}"


@pytest.mark.parametrize("M,tile", [(16, 8), (16, 0), (9, 4), (2, 12)])
def test_gemv_pair_equals_the_two_launches(dev, M, tile):
    """cbx_gemv_pair_f32: the o projection (+ residual, in place, packed) and the RMSNorm-folded gate | up SwiGLU GEMV that reads it, in ONE
    launch whose consumer workgroups request their weights before they wait for the producers.  Bit-identical to the two cbx_gemv_f32
    launches (same arithmetic, same order), twice in a row on the same counters (they re-arm themselves), error word untouched."""
    from chatterbox_amd import ops
    D, Fh = 1024, 2048
    att, x0 = _r((M, D), 1), _r((M, D), 2)
    wo, wg, wu, ln2 = _r((D, D), 3, 1 / math.sqrt(D)), _r((Fh, D), 4, 0.03), _r((Fh, D), 5, 0.03), 1 + 0.1 * _r((D,), 6)
    pk = dict(w_packed=True, x_packed=True, M=M)
    attp = ops.pack_gemv_weight(att.to(dev))
    wop = ops.pack_gemv_weight(wo.to(dev), half_tile=tile)
    wgu = ops.pack_gemv_weight(torch.cat([wg, wu]).to(dev), swiglu=True)
    o_kw = lambda cur: dict(N=D, K=D, nw=8, res=cur, out_packed=True, half_tile=tile, **pk)
    gu_kw = dict(N=Fh, K=D, swiglu=True, nw=8, norm_w=ln2.to(dev), out_packed=True, **pk)
    rows16 = (M + 15) // 16 * 16

    cur1, g1 = ops.pack_gemv_weight(x0.to(dev)), torch.zeros(rows16, Fh, device=dev)
    ops.gemv(attp, wop, cur1, **o_kw(cur1))
    ops.gemv(cur1, wgu, g1, **gu_kw)

    sync = torch.zeros(16, dtype=torch.int32, device=dev)
    for rep in range(2):
        cur2, g2 = ops.pack_gemv_weight(x0.to(dev)), torch.full((rows16, Fh), float("nan"), device=dev)
        ops.gemv_pair((attp, wop, cur2, o_kw(cur2)), (cur2, wgu, g2, gu_kw), sync)
        assert torch.equal(cur2.cpu(), cur1.cpu()), f"residual stream differs (launch {rep})"
        assert torch.equal(_unpack_operand(g2.cpu(), M, Fh), _unpack_operand(g1.cpu(), M, Fh)), f"SwiGLU output differs (launch {rep})"
        assert sync.cpu().tolist() == [0] * 16, f"counters re-armed, no time-out: {sync.cpu().tolist()}"
    h = x0 + F.linear(att, wo)
    hn = h * torch.rsqrt((h * h).mean(-1, keepdim=True) + 1e-5) * ln2
    _close(_unpack_operand(g2.cpu(), M, Fh), F.silu(F.linear(hn, wg)) * F.linear(hn, wu), 6e-5, "pair: SwiGLU(RMSNorm(x + att Wo^T))")


@pytest.mark.parametrize("M,dtile,qtile", [(16, 4, 12), (5, 4, 0), (16, 8, 0)])
def test_gemv_pair_down_and_next_qkv(dev, M, dtile, qtile):
    """cbx_gemv_pair_f32 with the plain consumer: the down projection of a layer (K = 4096 over 8 waves = four load batches, residual added in
    place) and the RMSNorm-folded q/k/v GEMV of the next layer that reads it -- bit-identical to the two launches, counters re-armed."""
    from chatterbox_amd import ops
    D, Fh = 1024, 4096
    g, x0 = _r((M, Fh), 1, 0.3), _r((M, D), 2)
    wd, wq, ln1 = _r((D, Fh), 3, 1 / math.sqrt(Fh)), _r((3 * D, D), 4, 1 / math.sqrt(D)), 1 + 0.1 * _r((D,), 5)
    pk = dict(w_packed=True, x_packed=True, M=M)
    gp = ops.pack_gemv_weight(g.to(dev))
    wdp, wqp = ops.pack_gemv_weight(wd.to(dev), half_tile=dtile), ops.pack_gemv_weight(wq.to(dev), half_tile=qtile)
    d_kw = lambda cur: dict(N=D, K=Fh, nw=8, res=cur, out_packed=True, half_tile=dtile, **pk)
    q_kw = dict(N=3 * D, K=D, nw=8, norm_w=ln1.to(dev), half_tile=qtile, **pk)
    cur1, q1 = ops.pack_gemv_weight(x0.to(dev)), torch.zeros(M, 3 * D, device=dev)
    ops.gemv(gp, wdp, cur1, **d_kw(cur1))
    ops.gemv(cur1, wqp, q1, **q_kw)
    sync = torch.zeros(16, dtype=torch.int32, device=dev)
    for rep in range(2):
        cur2, q2 = ops.pack_gemv_weight(x0.to(dev)), torch.full((M, 3 * D), float("nan"), device=dev)
        ops.gemv_pair((gp, wdp, cur2, d_kw(cur2)), (cur2, wqp, q2, q_kw), sync)
        assert torch.equal(cur2.cpu(), cur1.cpu()) and torch.equal(q2.cpu(), q1.cpu()), f"pair differs from the two launches (launch {rep})"
        assert sync.cpu().tolist() == [0] * 16, f"counters re-armed, no time-out: {sync.cpu().tolist()}"
    h = x0 + F.linear(g, wd)
    _close(q2, F.linear(h * torch.rsqrt((h * h).mean(-1, keepdim=True) + 1e-5) * ln1, wq), 1e-4, "pair: RMSNorm(x + g Wd^T) Wqkv^T")


@pytest.mark.parametrize("M,odtile,qtile", [(16, 4, 12), (3, 8, 0), (16, 0, 0)])
def test_gemv_chain_equals_the_four_launches(dev, M, odtile, qtile, D=1024, Fh=2048):
    """cbx_gemv_chain_f32: o projection (+ residual) -> RMSNorm + gate | up + SwiGLU -> down projection (+ residual) -> RMSNorm + q/k/v of the next
    layer in ONE launch of four roles, each waiting on the arrival counters of the one in front of it after its first weight batch is in flight.
    Bit-identical to the four cbx_gemv_f32 launches, three times in a row on the same counters."""
    from chatterbox_amd import ops
    att, x0 = _r((M, D), 1), _r((M, D), 2)
    wo, wg, wu = _r((D, D), 3, 1 / math.sqrt(D)), _r((Fh, D), 4, 0.03), _r((Fh, D), 5, 0.03)
    wd, wq = _r((D, Fh), 6, 1 / math.sqrt(Fh)), _r((3 * D, D), 7, 1 / math.sqrt(D))
    ln2, ln1 = 1 + 0.1 * _r((D,), 8), 1 + 0.1 * _r((D,), 9)
    pk = dict(w_packed=True, x_packed=True, M=M)
    attp = ops.pack_gemv_weight(att.to(dev))
    wop, wdp = ops.pack_gemv_weight(wo.to(dev), half_tile=odtile), ops.pack_gemv_weight(wd.to(dev), half_tile=odtile)
    wgu, wqp = ops.pack_gemv_weight(torch.cat([wg, wu]).to(dev), swiglu=True), ops.pack_gemv_weight(wq.to(dev), half_tile=qtile)
    rows16 = (M + 15) // 16 * 16

    def four(cur, g, q):
        res_kw = dict(nw=8, res=cur, out_packed=True, half_tile=odtile, **pk)
        return [(attp, wop, cur, dict(N=D, K=D, **res_kw)),
                (cur, wgu, g, dict(N=Fh, K=D, swiglu=True, nw=8, norm_w=ln2.to(dev), out_packed=True, **pk)),
                (g, wdp, cur, dict(N=D, K=Fh, **res_kw)),
                (cur, wqp, q, dict(N=3 * D, K=D, nw=8, norm_w=ln1.to(dev), half_tile=qtile, **pk))]

    cur1, g1, q1 = ops.pack_gemv_weight(x0.to(dev)), torch.zeros(rows16, Fh, device=dev), torch.zeros(M, 3 * D, device=dev)
    for x, w, out, kw in four(cur1, g1, q1):
        ops.gemv(x, w, out, **kw)
    sync = torch.zeros(64, dtype=torch.int32, device=dev)
    for rep in range(3):
        cur2, g2, q2 = ops.pack_gemv_weight(x0.to(dev)), torch.zeros(rows16, Fh, device=dev), torch.full((M, 3 * D), float("nan"), device=dev)
        ops.gemv_chain(four(cur2, g2, q2), sync)
        assert torch.equal(cur2.cpu(), cur1.cpu()) and torch.equal(g2.cpu(), g1.cpu()) and torch.equal(q2.cpu(), q1.cpu()), f"chain differs from the four launches (launch {rep})"
        assert not sync.cpu().any(), f"counters re-armed, no time-out: {sync.cpu().tolist()}"
    h = x0 + F.linear(att, wo)
    rms = lambda t, w: t * torch.rsqrt((t * t).mean(-1, keepdim=True) + 1e-5) * w
    hn = rms(h, ln2)
    h2 = h + F.linear(F.silu(F.linear(hn, wg)) * F.linear(hn, wu), wd)
    _close(q2, F.linear(rms(h2, ln1), wq), 2e-4, "chain vs torch")


@pytest.mark.parametrize("name", ["turbo_l2", "nano_l12"])
def test_t3_turbo_chained_decode_samples_the_reference_tokens(dev, name, monkeypatch):
    """CBX_TURBO_TUNE="chain=1,od_tc=4,d_ks=1,d_nw=8": the GPT-2 decode step as attention + ONE launch per layer (cbx_gemv_chain_f32 in its GPT-2
    form: biases, LayerNorm-folded roles, gelu_new) through the hipGraph path, against the golden tokens of the reference's inference_turbo
    (Turbo d = 1024, 2 layers; Nano d = 768, 12 layers); the counters' error word stays clean."""
    from chatterbox_amd import synth
    from chatterbox_amd.t3_turbo import T3TurboEngine
    g = np.load(os.path.join(GOLD, name + ".npz"))
    L, d, steps, n_text = int(g["n_layers"]), int(g["d"]), int(g["steps"]), int(g["n_text"])
    monkeypatch.setenv("CBX_TURBO_TUNE", "chain=1,od_tc=4,d_ks=1,d_nw=8")
    eng = T3TurboEngine(synth.t3_turbo_state_dict(L, d, 0), dev)
    assert eng.tune["chain"] == 1
    u = torch.from_numpy(g["uniforms"])[None]
    toks = eng.generate(synth.t3_cond(prompt_len=375), [synth.turbo_text_tokens(n_text)], max_gen_len=steps, uniforms=u, ban_eos=True,
                        temperature=0.8, top_k=1000, top_p=0.95, repetition_penalty=1.2)
    assert toks[0].tolist() == g["tokens"].tolist()
    assert not any(bool(st["dws"]["pair_ws"].any()) for st in eng._state.values())


def test_gemv_chain_at_the_nano_width(dev):
    """The same at d = 768 / 3072 (Nano's GPT-2 small): three K blocks per wave, i.e. a load batch whose fourth slot is idle."""
    test_gemv_chain_equals_the_four_launches(dev, 5, 4, 0, D=768, Fh=3072)
