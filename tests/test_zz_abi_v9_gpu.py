"""Decode-step geometry variants (-m gpu): 12- / 4-column GEMV tiles, the down projection without partial images, the software-pipelined /
non-temporal / speculative decode attention, the GEMV epilogue prefetch and deep load batches, the flash rel-pos encoder attention, and the T3
engines running on them -- each against the plain form BIT FOR BIT and against torch / the reference's golden tokens.

History: written at the end of round 3 without GPU access (SIMT-emulator evidence only); the first hardware run (GPUTEST_r03) failed
`test_decode_attn_rope_pipelined_equals_plain` -- hipcc's default -ffp-contract=fast had rounded one template instantiation's dot product
differently (v_pk_mul + adds instead of an fma chain).  The library is built with -ffp-contract=on since (chatterbox_amd/build.py), the geometry
travels per call (ABI v10: cbx_decode_attn_t, cbx_gemv_t.flags) and every variant bench.py may adopt is on an allow-list
(chatterbox_amd/decode_green.json) that `test_green_variant_*` re-checks on the hardware.  The file name still sorts last on purpose.
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


def _first_diff(a, b):
    """'row r, head h, element e: a vs b (max |d| ...)' of the first differing element of two (rows, H * 64) / cache tensors."""
    d = (a != b) & ~(torch.isnan(a) & torch.isnan(b))
    if not bool(d.any()):
        return "equal"
    i = int(torch.nonzero(d.flatten())[0])
    idx = np.unravel_index(i, tuple(a.shape))
    return (f"{int(d.sum())} of {d.numel()} elements differ, first at {tuple(int(x) for x in idx)}: {a.flatten()[i].item()!r} vs {b.flatten()[i].item()!r}, "
            f"max |d| {torch.nan_to_num(a - b).abs().max().item():.3e}, non-finite: {int((~torch.isfinite(a)).sum())} / {int((~torch.isfinite(b)).sum())}")


@pytest.mark.parametrize("pipe,u", [(1, 4), (1, 8), (2, 4), (3, 4), (4, 4), (5, 4), (6, 4), (7, 4)])
@pytest.mark.parametrize("rows,H,rope,split_min", [(3, 16, True, 512), (1, 12, False, 1), (16, 16, True, 512)])
def test_decode_attn_rope_pipelined_equals_plain(dev, rows, H, rope, split_min, pipe, u):
    """cbx_decode_attn_t.pipeline (bit 0: the next step's K / V rows are requested before the current step is multiplied, two register sets;
    bit 1: non-temporal K / V loads; bit 2: positions 0 .. 63 requested before positions[row] has arrived) and .unroll (4 / 8 / 16 rows per lane
    group and step) against the plain form (pipeline 0, the SAME unroll: another unroll folds the online softmax in other groups): same arithmetic in the same order, so outputs and appended cache rows are
    equal bit for bit -- contexts of 1 .. 70 (fewer rows than one step, exactly one step, the ragged tail of the second register set), a few
    long ones, the one-workgroup and the split-context grids.  The cache past each row's context is poisoned with NaN: whatever a variant
    reads there must not matter.  One test per (shape, mode): a failure names the mode, the context, the row and the size of the difference."""
    from chatterbox_amd import ops
    from oracle import ref_torch as O
    maxp = 640
    kc0, vc0 = _r((rows, H, maxp, 64), 1), _r((rows, H, maxp, 64), 2)
    cos, sin = O.rope_cos_sin(torch.arange(maxp), O.llama3_inv_freq())
    cd, sd_ = (cos.to(dev), sin.to(dev)) if rope else (None, None)
    plain = ops.DecodeAttnGeom(dev, unroll=u, pipeline=0, split_min=split_min)
    var = ops.DecodeAttnGeom(dev, unroll=u, pipeline=pipe, split_min=split_min)  # its own workspace: nothing shared between the two forms
    for n in list(range(0, 70, 1 if rows < 16 else 9)) + [127, 128, 129, 255, 300, 639]:
        pos = torch.tensor([(n + 37 * r) % maxp for r in range(rows)], dtype=torch.int32)
        qkv = _r((rows, 3 * H * 64), 100 + n)
        res = []
        for geom in (plain, var):
            kc, vc, out = kc0.clone(), vc0.clone(), torch.zeros(rows, H * 64, device=dev)
            for r in range(rows):
                kc[r, :, int(pos[r]):], vc[r, :, int(pos[r]):] = float("nan"), float("nan")
            kc, vc = kc.to(dev), vc.to(dev)
            ops.decode_attn_rope(qkv.to(dev), pos.to(dev), cd, sd_, kc, vc, out, 0.125, geom=geom)
            kc, vc = kc.cpu(), vc.cpu()
            assert all(bool(torch.isnan(c[r, :, int(pos[r]) + 1:]).all()) and bool(torch.isfinite(c[r, :, : int(pos[r]) + 1]).all())
                       for c in (kc, vc) for r in range(rows)), f"pipeline {geom.pipeline}, unroll {geom.unroll}, context {n + 1}: exactly one row appended per (row, head)"
            res.append((out.cpu(), torch.nan_to_num(kc), torch.nan_to_num(vc)))
        for a, b, what in zip(res[0], res[1], ("output", "k cache", "v cache")):
            assert torch.equal(a, b) and bool(torch.isfinite(a).all()), \
                f"pipeline {pipe}, unroll {u}, contexts {[int(p) + 1 for p in pos]}: {what} differs from the plain kernel: {_first_diff(a, b)}"
        if n in (0, 63, 64, 65, 300):  # and the result itself against torch
            q, k, v = (qkv.view(rows, 3, H, 64)[:, i] for i in range(3))
            if rope:
                c, s_ = cos[pos.long()][:, None], sin[pos.long()][:, None]
                q, k = q * c + O._rot_half(q) * s_, k * c + O._rot_half(k) * s_
            for r in range(rows):
                m = int(pos[r])
                kk, vv = torch.cat([kc0[r, :, :m], k[r][:, None]], 1), torch.cat([vc0[r, :, :m], v[r][:, None]], 1)
                _close(res[1][0][r].view(H, 64), F.scaled_dot_product_attention(q[r].reshape(H, 1, 64), kk, vv)[:, 0], 2e-5, f"pipeline {pipe}, ctx {m + 1}")


def test_decode_attn_split_workspaces_are_caller_owned(dev):
    """Two split-context attention launches in flight on two streams, each with its OWN cbx_decode_attn_t workspace (the process-wide
    one could be shared by concurrent launches): both equal the same launches run one after the other."""
    from chatterbox_amd import ops
    rows, H, maxp = 1, 12, 1536
    kc0, vc0 = _r((2, rows, H, maxp, 64), 1).to(dev), _r((2, rows, H, maxp, 64), 2).to(dev)
    qkv = _r((2, rows, 3 * H * 64), 3).to(dev)
    pos = torch.tensor([[1200], [900]], dtype=torch.int32, device=dev)
    geoms = [ops.DecodeAttnGeom(dev, split_min=512), ops.DecodeAttnGeom(dev, split_min=512)]
    assert geoms[0].ws.data_ptr() != geoms[1].ws.data_ptr()

    def run(i, kc, vc, out):
        ops.decode_attn_rope(qkv[i], pos[i], None, None, kc, vc, out, 0.125, geom=geoms[i])

    serial = []
    for i in range(2):
        kc, vc, out = kc0[i].clone(), vc0[i].clone(), torch.zeros(rows, H * 64, device=dev)
        run(i, kc, vc, out)
        serial.append(out.cpu())
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    for rep in range(20):
        outs, keep = [], []
        for i in range(2):
            kc, vc, out = kc0[i].clone(), vc0[i].clone(), torch.zeros(rows, H * 64, device=dev)
            keep.append((kc, vc))
            outs.append(out)
        torch.cuda.synchronize()
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                run(i, keep[i][0], keep[i][1], outs[i])
        torch.cuda.synchronize()
        for i in range(2):
            assert torch.equal(outs[i].cpu(), serial[i]), f"round {rep}, stream {i}: {_first_diff(outs[i].cpu(), serial[i])}"
    for i in range(2):  # and against torch
        m = int(pos[i, 0])
        q, k, v = (qkv[i].cpu().view(rows, 3, H, 64)[:, j] for j in range(3))
        kk, vv = torch.cat([kc0[i, 0, :, :m].cpu(), k[0][:, None]], 1), torch.cat([vc0[i, 0, :, :m].cpu(), v[0][:, None]], 1)
        _close(serial[i][0].view(H, 64), F.scaled_dot_product_attention(q[0].reshape(H, 1, 64), kk, vv)[:, 0], 2e-5, f"split attention, ctx {m + 1}")


@pytest.mark.parametrize("M,N,K,ct,ks,np_", [(16, 3072, 1024, 3, 4, 2), (16, 3072, 1024, 3, 4, 4), (16, 3072, 1024, 2, 2, 0), (16, 8194, 1024, 2, 1, 2), (16, 8194, 1024, 4, 1, 0),
                                               (5, 3072, 1024, 3, 4, 2), (16, 3072, 1024, 1, 1, 2), (9, 200, 2048, 3, 2, 0), (16, 3072, 1024, 3, 1, 4)])
def test_gemv_col_tiles_and_split_k(dev, M, N, K, ct, ks, np_):
    """cbx_gemv_t.col_tiles (ABI v11): the RMSNorm-folded packed GEMV with `ct` column tiles per workgroup sharing every x register and 1 / ks of K
    per workgroup.  ks == 1: bit-identical to the one-tile kernel (same MFMA order per tile, same 8-way LDS reduction) and equal to torch;
    ks > 1: UN-normalised partial sums + per-slice sums of squares whose fixed-order fold (what cbx_decode_attn_t.qkv_nparts does) equals torch;
    with partial images (np_) the reduced residual stream x_out is exact.  Ragged N (8194 = 513 tiles, 200 = 13 tiles), M < 16."""
    from chatterbox_amd import ops
    x, w, nwt = _r((M, K), 1), _r((N, K), 2, 1 / math.sqrt(K)), 1 + 0.1 * _r((K,), 3)
    parts = _r((max(np_, 1), M, K), 5, 0.3)
    xs = x + sum(parts[j] for j in range(np_)) if np_ else x
    ref = F.linear(xs * torch.rsqrt((xs * xs).mean(-1, keepdim=True) + 1e-5) * nwt, w)
    xp, wp = ops.pack_gemv_weight(x.to(dev)), ops.pack_gemv_weight(w.to(dev))
    extra = dict(norm_w=nwt.to(dev))
    mk_extra = lambda: dict(extra, xpart=torch.stack([ops.pack_gemv_weight(parts[j].to(dev)) for j in range(np_)]), x_out=torch.zeros_like(xp)) if np_ else dict(extra)
    kw = dict(N=N, M=M, K=K, nw=8, w_packed=True, x_packed=True)
    tol = 3e-5 * max(1.0, math.sqrt(K / 256))
    if ks == 1:
        a, b = torch.zeros(M, N, device=dev), torch.full((M, N), float("nan"), device=dev)
        ea, eb = mk_extra(), mk_extra()
        ops.gemv(xp, wp, a, **kw, **ea)
        ops.gemv(xp, wp, b, col_tiles=ct, **kw, **eb)
        assert torch.equal(a, b), f"col_tiles = {ct} differs from the one-tile kernel: {_first_diff(a.cpu(), b.cpu())}"
        _close(b, ref, 2 * tol, f"gemv col_tiles = {ct}")
        if np_:
            assert torch.equal(ea["x_out"], eb["x_out"])
            _close(_unpack_operand(eb["x_out"], M, K), xs, 1e-6, "x_out = x + partial images")
        return
    out, ssq = torch.full((ks, M, N), float("nan"), device=dev), torch.full((ks, 16), float("nan"), device=dev)
    e = mk_extra()
    ops.gemv(xp, wp, out, col_tiles=ct, ksplit=ks, ssq_out=ssq, **kw, **e)
    o, q = out.cpu(), ssq.cpu()
    assert bool(torch.isfinite(o).all()) and bool(torch.isfinite(q[:, :M]).all())
    acc, sq = o[0].clone(), q[0, :M].clone()
    for j in range(1, ks):  # the consumer's fixed-order fold
        acc += o[j]
        sq += q[j, :M]
    _close(sq, (xs * xs).sum(-1), 1e-4 * K / 256, "sum of squares over the K slices")
    _close(acc * torch.rsqrt(sq / K + 1e-5)[:, None], ref, 2 * tol, f"split-K col_tiles = {ct}, ksplit = {ks}")
    if np_:
        _close(_unpack_operand(e["x_out"], M, K), xs, 1e-6, "x_out = x + partial images (written slice by slice)")


@pytest.mark.parametrize("M,N,K,ct,np_", [(1, 6563, 1024, 2, 2), (5, 6563, 768, 2, 0), (16, 3072, 1024, 3, 4)])
def test_gemv_col_tiles_layernorm_form(dev, M, N, K, ct, np_):
    """cbx_gemv_t.col_tiles with ln_cw / ln_cb (GPT-2: ln_f folded into the head GEMV of Turbo / Nano, K = 1024 / 768): bit-identical to the one-tile
    kernel's LayerNorm form and equal to torch."""
    from chatterbox_amd import ops
    x, w, lw, lb, bias = _r((M, K), 1), _r((N, K), 2, 1 / math.sqrt(K)), 1 + 0.1 * _r((K,), 3), 0.1 * _r((K,), 4), _r((N,), 6)
    parts = _r((max(np_, 1), M, K), 5, 0.3)
    xs = x + sum(parts[j] for j in range(np_)) if np_ else x
    ref = F.linear(F.layer_norm(xs, (K,), lw, lb, 1e-5), w, bias)
    xp, wp = ops.pack_gemv_weight(x.to(dev)), ops.pack_gemv_weight(w.to(dev))
    cw, cb = F.linear(lw[None], w)[0], F.linear(lb[None], w)[0] + bias
    mk = lambda: dict(norm_w=lw.to(dev), ln_cw=cw.to(dev), ln_cb=cb.to(dev),
                      **(dict(xpart=torch.stack([ops.pack_gemv_weight(parts[j].to(dev)) for j in range(np_)]), x_out=torch.zeros_like(xp)) if np_ else {}))
    kw = dict(N=N, M=M, K=K, nw=8, w_packed=True, x_packed=True)
    a, b = torch.zeros(M, N, device=dev), torch.full((M, N), float("nan"), device=dev)
    ops.gemv(xp, wp, a, **kw, **mk())
    ops.gemv(xp, wp, b, col_tiles=ct, **kw, **mk())
    assert torch.equal(a, b), f"col_tiles = {ct} (LayerNorm form) differs from the one-tile kernel: {_first_diff(a.cpu(), b.cpu())}"
    _close(b, ref, 1e-4 * max(1.0, math.sqrt(K / 256)), "LayerNorm-folded gemv on column tiles")


@pytest.mark.parametrize("rows,H,nparts,pipe", [(16, 16, 4, 0), (16, 16, 4, 7), (5, 16, 2, 1), (2, 12, 4, 0)])
def test_decode_attn_folds_qkv_partial_sums(dev, rows, H, nparts, pipe):
    """cbx_decode_attn_t.qkv_nparts (ABI v11): the attention launch adds the split-K partial sums of the q/k/v row in fixed order and applies
    rstd = rsqrt(sum ssq / dim + eps) -- against the same launch on the finished row (outputs to 1e-6: the device rsqrt may differ from the
    host's by an ulp, which a sharp softmax amplifies), appended cache rows included; one-workgroup and split-context grids, plain and pipelined
    forms."""
    from chatterbox_amd import ops
    from oracle import ref_torch as O
    maxp, D = 640, H * 64
    kc0, vc0 = _r((rows, H, maxp, 64), 1), _r((rows, H, maxp, 64), 2)
    cos, sin = O.rope_cos_sin(torch.arange(maxp), O.llama3_inv_freq())
    parts, ssq = _r((nparts, rows, 3 * D), 3), torch.zeros(nparts, 16)
    ssq[:, :rows] = _r((nparts, rows), 4).abs() * D / nparts
    tot, sq = parts[0].clone(), ssq[0].clone()
    for j in range(1, nparts):
        tot += parts[j]
        sq += ssq[j]
    full = tot * torch.rsqrt(sq[:rows] / D + 1e-5)[:, None]
    for ctx in (1, 40, 65, 300, 639):
        pos = torch.tensor([(ctx - 1 + 37 * r) % maxp for r in range(rows)], dtype=torch.int32)
        res = []
        for src in ("parts", "row"):
            geom = ops.DecodeAttnGeom(dev, pipeline=pipe, split_min=256)
            kc, vc, out = kc0.clone().to(dev), vc0.clone().to(dev), torch.zeros(rows, D, device=dev)
            if src == "parts":
                ops.decode_attn_rope(parts.to(dev), pos.to(dev), cos.to(dev), sin.to(dev), kc, vc, out, 0.125, geom=geom, qkv_ssq=ssq.to(dev), rms_dim=D)
            else:
                ops.decode_attn_rope(full.to(dev), pos.to(dev), cos.to(dev), sin.to(dev), kc, vc, out, 0.125, geom=geom)
            res.append((out.cpu(), kc.cpu(), vc.cpu()))
        for a, b, what in zip(res[0], res[1], ("output", "k cache", "v cache")):
            _close(a, b, 1e-5, f"attention on q/k/v partial sums vs the finished row: {what}, ctx {ctx}")


@pytest.mark.parametrize("tune", ["qkv_tc=12,od_tc=4,d_ks2=1", "qkv_tc=12", "od_tc=4,d_ks2=1,d_nw2=8", "od_tc=4", "qkv_ks=4,qkv_ct=3", "qkv_ks=4,qkv_ct=3,head_ct=2",
                                  "qkv_ks=4,qkv_ct=3,head_ct=2,half_tiles=0,d_ks2=4", "qkv_ks=2,qkv_ct=2,head_ct=4,od_tc=4,d_ks2=1,d_nw2=8"])
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
    """cbx_gemv_t.flags & CBX_GEMV_DEEP: an 8-wave plain packed GEMV whose waves own >= 256 of K requests 8 K blocks per load batch instead of 4
    (the partial-free down projection: K = 4096 over 8 waves).  Same blocks in the same order: bit-identical."""
    from chatterbox_amd import ops
    x, w, r = _r((M, K), 1), _r((N, K), 2, 1 / math.sqrt(K)), _r((M, (N + 31) // 32 * 32), 3)
    xp, wp = ops.pack_gemv_weight(x.to(dev)), ops.pack_gemv_weight(w.to(dev), half_tile=tile)
    kw = dict(N=N, M=M, K=K, ksplit=ks, nw=8, w_packed=True, x_packed=True, half_tile=tile)
    outs = []
    for deep in (0, 1):
        fl = ops.gemv_flags(deep=deep)
        if res:
            o = ops.pack_gemv_weight(r.to(dev))
            ops.gemv(xp, wp, o, res=o, out_packed=True, flags=fl, **kw)
        else:
            o = torch.zeros((ks, M, N) if ks > 1 else (M, N), device=dev)
            ops.gemv(xp, wp, o, flags=fl, **kw)
        outs.append(o.cpu())
    assert torch.equal(outs[0], outs[1]), _first_diff(outs[0], outs[1])
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
    and an identity verdict (logits of single steps over ragged contexts 1 .. 640 + the timed run's final logits) or an error -- a reordering
    candidate is never the adopted one; the attention / epilogue / tile variants that claim the same arithmetic ARE identical on the hardware."""
    from chatterbox_amd import autotune as at, synth
    from chatterbox_amd.t3 import T3Engine
    g = np.load(os.path.join(GOLD, "t3_l2.npz"))
    steps, n_text = int(g["steps"]), int(g["n_text"])
    eng = T3Engine(synth.t3_state_dict(2, 0), dev)
    rep = eng.autotune(B=8, ctx=128, steps=16, reps=2, timeout=300.0)
    assert "error" not in rep, rep
    rows = [r for r in rep["candidates"] if "variant" in r]
    assert len(rows) >= len(at.TILE_VARIANTS) + len(at.ATTN_VARIANTS) and all(("ms_per_token" in r) != ("error" in r) for r in rows), rows
    same_arith = [r for r in rows if "ms_per_token" in r and not any(k in r["variant"] for k in ("d_ks2", "d_nw2", "da_u", "qkv_ks"))]
    assert all(r.get("identical") for r in same_arith), [r for r in same_arith if not r.get("identical")]
    best = rep["best"]
    if best:
        assert next(r for r in rows if r["variant"] == best)["identical"]  # (the partial-free geometry reorders)
        assert all(eng.tune[k] == v for k, v in at.split_variant(best)[0].items()) and all(eng.knobs[k] == v for k, v in at.split_variant(best)[1].items())
    u = torch.from_numpy(g["uniforms"])[None]
    toks = eng.generate(synth.t3_cond(), [synth.text_tokens(n_text)], max_new_tokens=steps, uniforms=u, ban_eos=True, **SAMP)
    assert toks[0].tolist() == g["tokens"].tolist(), f"adopted geometry {best}"


def _green():
    from chatterbox_amd import autotune as at
    return [dict(v) for v in sorted(at.green_variants())]


@pytest.mark.parametrize("variant", _green(), ids=lambda v: ",".join(f"{k}={x}" for k, x in sorted(v.items())) or "r03-base")
def test_green_variant_is_bit_identical_and_samples_the_reference_tokens(dev, variant):
    """Every geometry on the allow-list bench.py may run or adopt (chatterbox_amd/decode_green.json; an entry = the keys in which a full geometry
    differs from the frozen round-3 base, autotune.BASE_TUNE / BASE_KNOBS), on THIS hardware: (a) its logits over the ragged probe contexts
    {1, 38, 63, 64, 65, 225, 640} at B = 8 and B = 1 (split grid) equal the base geometry's bit for bit -- or, for a geometry that sums a
    projection in another order (split-K factors, wave counts), to 2e-4 of the logit scale --, (b) it samples the reference's golden tokens
    (t3_l2, 64 steps, hipGraph + C-step path)."""
    from chatterbox_amd import autotune as at, synth
    from chatterbox_amd.t3 import T3Engine
    g = np.load(os.path.join(GOLD, "t3_l2.npz"))
    steps, n_text = int(g["steps"]), int(g["n_text"])
    sd = synth.t3_state_dict(2, 0)
    base, eng = T3Engine(sd, dev), T3Engine(sd, dev)
    base.apply_variant(dict(at.BASE_TUNE), dict(at.BASE_KNOBS))
    t, k = at.split_variant(variant)
    eng.apply_variant(dict(at.BASE_TUNE, **t), dict(at.BASE_KNOBS, **k))
    assert at.canon(eng.tune, eng.knobs) == at.canon(variant)
    reorders = any(x in variant for x in ("d_ks2", "d_nw2", "o_nw2", "gu_nw", "da_u", "qkv_ks"))
    for B in (8, 1):
        a, b = base.probe_decode(B=B).cpu(), eng.probe_decode(B=B).cpu()
        if reorders:
            assert (a - b).abs().max() <= 2e-4 * max(1.0, float(a.abs().max())), f"B = {B}: {_first_diff(a, b)}"
        else:
            assert torch.equal(a, b), f"{variant}, B = {B}: probe logits differ from the base geometry: {_first_diff(a, b)}"
    u = torch.from_numpy(g["uniforms"])[None]
    toks = eng.generate(synth.t3_cond(), [synth.text_tokens(n_text)], max_new_tokens=steps, uniforms=u, ban_eos=True, **SAMP)
    assert toks[0].tolist() == g["tokens"].tolist(), f"{variant}: golden tokens"


def test_the_default_geometry_is_on_the_allow_list():
    """What T3Engine runs out of the box (T3Engine._TUNE + autotune.LIB_KNOBS) must itself be a hardware-verified geometry."""
    from chatterbox_amd import autotune as at
    from chatterbox_amd.t3 import T3Engine
    assert at.canon(T3Engine._TUNE, at.LIB_KNOBS) in at.green_variants(), at.canon(T3Engine._TUNE, at.LIB_KNOBS)


def test_t3_prefill_through_the_c_entry_point_equals_the_python_sequence(dev, layers=2, lens=(12, 20, 7), steps=6):
    """cbx_t3_prefill (stage-level C seam of T3.inference's prefill, t3.py:303-335): the KV cache and the prefill logits of a ragged batch are
    bit-identical to issuing the same launches one by one from Python (T3Engine._layer_prefill), and the sampled tokens follow."""
    from chatterbox_amd import synth
    from chatterbox_amd.t3 import T3Engine
    sd = synth.t3_state_dict(layers, 0)
    tt = [synth.text_tokens(n, seed=i + 1) for i, n in enumerate(lens)]
    u = synth.rand((len(lens), steps), seed=3)
    kw = dict(max_new_tokens=steps, uniforms=u, ban_eos=True, return_prefill_logits=True, **SAMP)
    res = []
    for c_step in (True, False):
        eng = T3Engine(sd, dev)
        eng.c_step = c_step
        toks, logits = eng.generate(synth.t3_cond(), tt, **kw)
        st = next(iter(eng._state.values()))
        res.append(([t.tolist() for t in toks], logits.cpu(), st["kc"].cpu().clone(), st["vc"].cpu().clone()))
    assert res[0][0] == res[1][0]
    for a, b, what in zip(res[0][1:], res[1][1:], ("prefill logits", "k cache", "v cache")):
        assert torch.equal(a, b), f"cbx_t3_prefill vs the Python launch sequence: {what}: {_first_diff(a, b)}"


def test_t3_cached_voice_prefix_prefill_equals_the_full_prefill(dev, layers=2, steps=6):
    """Round 6: after the first prefill with a voice the K / V of its 34 conditioning positions are kept; the next prefills with the same conditioning tensors run over
    the text positions only and read the prefix keys from the KV cache (cbx_flash_attn_kv_f32).  The prompt is causal, so nothing may change: KV cache, prefill logits
    and tokens of the second / third call (another batch shape) are bit-identical to an engine that never shares; an in-place edit of a conditioning tensor is a miss."""
    from chatterbox_amd import synth
    from chatterbox_amd.t3 import T3Engine
    sd = synth.t3_state_dict(layers, 0)
    cond = synth.t3_cond()
    a, b = T3Engine(sd, dev), T3Engine(sd, dev)
    b.share_prefix = False
    for lens in ((12, 20, 7), (12, 20, 7), (31, 5)):
        tt = [synth.text_tokens(n, seed=i + 1) for i, n in enumerate(lens)]
        u = synth.rand((len(lens), steps), seed=3)
        kw = dict(max_new_tokens=steps, uniforms=u, ban_eos=True, return_prefill_logits=True, **SAMP)
        hit = a._voice_prefix(cond) is not None
        res = []
        for eng in (a, b):
            toks, logits = eng.generate(cond, tt, **kw)
            st = eng._state[next(k for k in eng._state if k[0] == len(lens))]
            S = 34 + max(lens) + 2
            res.append(([t.tolist() for t in toks], logits.cpu(), st["kc"][:, :, :, :S].cpu().clone(), st["vc"][:, :, :, :S].cpu().clone()))
        assert res[0][0] == res[1][0], f"tokens, lens {lens} (prefix cached: {hit})"
        for x, y, what in zip(res[0][1:], res[1][1:], ("prefill logits", "k cache", "v cache")):
            assert torch.equal(x, y), f"text-only prefill behind the cached prefix vs the full prefill: {what}, lens {lens}: {_first_diff(x, y)}"
    assert len(a._prefix_cache) == 1 and not b._prefix_cache and a._voice_prefix(cond) is not None
    assert a._voice_prefix(dict(cond)) is not None, "a new dict over the same tensors is the same voice"
    cond["speaker_emb"].mul_(1.0)  # in-place write: the version counter moves
    assert a._voice_prefix(cond) is None, "an edited conditioning tensor must not hit"
    assert a._voice_prefix(synth.t3_cond()) is None, "equal content in other tensors is a miss (identity, not a device-side compare: no sync in generate())"
    with torch.inference_mode():  # conditioning tensors made under inference_mode (the API layer) carry no version counter: identity + host content
        ci = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in synth.t3_cond().items()}
    t1 = a.generate(ci, tt, **kw)[0]
    assert a._voice_prefix(ci) is not None
    t2 = a.generate(ci, tt, **kw)[0]
    assert [t.tolist() for t in t1] == [t.tolist() for t in t2]


def test_two_engines_with_different_geometries_in_one_process(dev):
    """ABI v10: the decode geometry travels per call (cbx_decode_attn_t, cbx_gemv_t.flags, cbx_t3_step_t.da_* / gemv_flags) -- nothing is
    process-wide.  Two T3Engines with different geometries, their decode graphs captured one after the other and replayed INTERLEAVED, each
    produce the tokens they produce alone (the reference's golden tokens); a graph captured before the other engine changed ITS geometry keeps
    running its own kernels."""
    from chatterbox_amd import synth
    from chatterbox_amd.t3 import T3Engine
    g = np.load(os.path.join(GOLD, "t3_l2.npz"))
    steps, n_text = int(g["steps"]), int(g["n_text"])
    sd = synth.t3_state_dict(2, 0)
    ea, eb = T3Engine(sd, dev), T3Engine(sd, dev)
    ea.apply_variant(dict(ea.tune, qkv_tc=12), dict(ea.knobs, da_pipe=7, pre_epi=1))
    eb.apply_variant(dict(eb.tune, od_tc=4, qkv_ks=0, head_ct=0), dict(eb.knobs, da_pipe=0, da_u=8, pre_epi=0))
    u = torch.from_numpy(g["uniforms"])[None]
    kw = dict(max_new_tokens=steps, uniforms=u, ban_eos=True, async_mode=True, run_steps=1, **SAMP)
    ha = ea.generate(synth.t3_cond(), [synth.text_tokens(n_text)], slot=0, **kw)
    hb = eb.generate(synth.t3_cond(), [synth.text_tokens(n_text)], slot=0, **kw)
    sa, sb = next(iter(ea._state.values())), next(iter(eb._state.values()))
    assert (sa["da"].pipeline, sa["da"].unroll, ea._gf()) == (7, 0, 1) and (sb["da"].pipeline, sb["da"].unroll, eb._gf()) == (0, 8, 0)
    assert sa["da"].ws is None or sb["da"].ws is None or sa["da"].ws.data_ptr() != sb["da"].ws.data_ptr()
    for _ in range(steps - 1):  # interleaved replays of the two captured graphs
        ea.advance(ha, 1)
        eb.advance(hb, 1)
    ta, tb = ea.collect(ha)[0].tolist(), eb.collect(hb)[0].tolist()
    assert ta == g["tokens"].tolist() and tb == g["tokens"].tolist()


_EPI_BODIES = [("test_gemv_decode", (16, 3072, 1024, 1, 8)), ("test_gemv_decode", (40, 1024, 1024, 2, 4)), ("test_gemv_decode", (6, 64, 256, 1, 4)),
               ("test_gemv_swiglu", ()), ("test_gemv_packed_rms_fused", (16, 3072, 1024, False, 8)),
               ("test_gemv_packed_residual_epilogue", (16, 1024, 1024, 16)), ("test_gemv_packed_residual_epilogue", (7, 1024, 1024, 8)),
               ("test_gemv_layernorm_fused", (5, 2304, 768, 4, False)), ("test_gemv_layernorm_fused", (16, 4096, 1024, 0, True)),
               ("test_gemv_half_tile", (16, 1024, 1024, 1, 8, True)), ("test_gemv_narrow_tiles", (16, 1024, 4096, 1, 16, 4, "res")),
               ("test_gemv_narrow_tiles", (16, 3072, 1024, 1, 8, 12, "rms_np2")), ("test_gemv_deep_batches_equal_plain", (16, 1024, 4096, 1, 4, True))]


@pytest.mark.parametrize("name,args", _EPI_BODIES, ids=[f"{n}{list(a)}" for n, a in _EPI_BODIES])
def test_gemv_epilogue_prefetch_equals_plain(dev, name, args, monkeypatch):
    """cbx_gemv_t.flags & CBX_GEMV_PRE_EPI: the residual element, the bias and the LayerNorm-fold constants of a GEMV's epilogue are requested
    with the first weight batch instead of after the reduction.  Every GEMV launch of the established bodies (bias, split-K, swiglu, RMSNorm
    and LayerNorm folds, in-place packed residual, narrow tiles, deep batches) runs twice -- knob off, knob on -- from the same memory state:
    outputs bit-identical, and the body's own comparison against torch holds with the knob on."""
    import sys
    import test_ops_gpu
    from chatterbox_amd import ops
    real, n = ops.gemv, [0]

    def both(x, w, out, **kw):
        keep = [(t, t.clone()) for t in (out, kw.get("x_out")) if t is not None]
        fl = int(kw.pop("flags", 0))
        real(x, w, out, flags=fl & ~ops.GEMV_PRE_EPI, **kw)
        plain = [t.clone() for t, _ in keep]
        for t, c in keep:
            t.copy_(c)
        r = real(x, w, out, flags=fl | ops.GEMV_PRE_EPI, **kw)
        for (t, _), p in zip(keep, plain):
            assert torch.equal(t.cpu(), p.cpu()), f"{name}{list(args)}: launch {n[0]} differs with the epilogue operands prefetched: {_first_diff(t.cpu(), p.cpu())}"
        n[0] += 1
        return r

    monkeypatch.setattr(ops, "gemv", both)
    body = getattr(test_ops_gpu, name, None) or getattr(sys.modules[__name__], name)
    body(dev, *args)
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
    calls, real, seam_calls, real_seam = [0], ops.flash_relpos, [0], eng._encode_c

    def counted(*a, **k):
        calls[0] += 1
        return real(*a, **k)

    def counted_seam(*a, **k):
        seam_calls[0] += 1
        return real_seam(*a, **k)

    ops.flash_relpos, eng._encode_c = counted, counted_seam
    try:
        # the Python launch sequencing (c_seam off) counts kernel-level flash launches; the stage-level C entry point cbx_s3gen_encode (the engines' default
        # since round 5) serves exactly the calls that take the flash form: one call instead of the 3 launches
        for seam in (False, True):
            eng.c_seam = seam
            for mode, cap, want in (("auto", 32 << 30, 0), ("auto", 1, 3), ("1", 32 << 30, 3)):  # 2 + 1 conformer layers in this small model
                calls[0] = seam_calls[0] = 0
                eng.ENC_FLASH, eng.ENC_SCORE_BYTES = mode, cap
                mel2 = eng.inference(toks, lens, ref, z=z, n_steps=3).cpu()
                if seam:
                    assert (calls[0], seam_calls[0]) == (0, 1 if want else 0), f"c_seam, CBX_ENC_FLASH={mode}, cap {cap}: {calls[0]} launches, {seam_calls[0]} seam calls"
                else:
                    assert (calls[0], seam_calls[0]) == (want, 0), f"CBX_ENC_FLASH={mode}, cap {cap}: {calls[0]} flash launches, {seam_calls[0]} seam calls"
                assert (mel - mel2).abs().max() <= 2e-5, f"flash rel-pos encoder ({mode}, cap {cap}, c_seam {seam}): {(mel - mel2).abs().max():.3e}"
    finally:
        ops.flash_relpos = real
    for b, n in enumerate(Ns):
        om = O.flow_inference(sd, toks[b:b + 1, :n], torch.tensor([n]), ref, z[b:b + 1, : 2 * (P + n)].transpose(1, 2), 3)
        err = (mel2[b, : 2 * n] - om[0].t()).abs()
        assert err.mean() <= 1e-4 and err.max() <= 1e-3, f"utt {b}: mel L1 {err.mean():.3e} max {err.max():.3e}"
