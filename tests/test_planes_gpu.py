"""Plane-format operands (-m gpu; ABI v7: gemm_planes.hip, attention_planes.hip, the planes LayerNorm): every entry point through the C
ABI against an fp64 torch reference of the same op on the same seeded inputs.

Tolerance: the f16x3 arithmetic carries 22 significand bits per operand and drops the l*l product; measured at the exact-fp32 MFMA's error
level (tests/test_ops_gpu.py) -- the fp32 GEMM tolerance 3e-5 * (1 + |ref|) * sqrt(K / 256) is asserted, against fp64.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TILES = list(range(0, 20)) + [21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33, 34, 35, 36, 37, 41, 42, 43, 44, 45]  # 21 .. 25: the loader-wave forms, 26 .. 28: 16-wave workgroups  # 0 = the dispatcher's own choice, 1..17 = the menu of gemm_planes.hip


def _r(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def _close(got, ref, tol, what=""):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    err = (got - ref).abs()
    bad = err > tol * (1.0 + ref.abs())
    assert not bad.any(), f"{what}: max err {err.max():.3e} (ref max {ref.abs().max():.3e}), {int(bad.sum())} / {bad.numel()} over tol {tol}"


def _planes_exact(x):
    """What a planes tensor holds for fp32 x: h + l / 2048 in fp64."""
    h = x.half()
    l = ((x.double() - h.double()) * 2048.0).half()
    return h.double() + l.double() / 2048.0


def test_split_planes_roundtrip_and_range_flag(dev):
    from chatterbox_amd import ops
    x = _r((300, 320), 1) * torch.logspace(-3, 3, 320)[None]
    P = ops.split_planes(x.to(dev))
    assert torch.equal(P.float().double().cpu(), _planes_exact(x).float().double()) or (P.float().cpu().double() - _planes_exact(x)).abs().max() < 1e-12
    err = (P.float().cpu().double() - x.double()).abs()
    # 22 significand bits wherever h is a normal fp16 (|x| >= 6.1e-5); below that the second plane's subnormal spacing / 2048 bounds the error
    assert bool((err <= x.abs().double() * 2.0 ** -21 + 3e-11).all()), f"plane pair error {float((err / x.abs().double().clamp_min(1e-30)).max()):.3e}"
    # a column range of a wider tensor
    W = ops.Planes(300, 512, dev, zero=True)
    ops.split_planes(x.to(dev)[:, :80], W.cols(256, 80))
    assert torch.equal(W.cols(256, 80).float().cpu(), P.cols(0, 80).float().cpu()) and float(W.cols(0, 256).float().abs().max()) == 0.0
    ops.enable_range_flag(dev)
    assert not ops.range_flag_tripped()
    big = x.clone()
    big[17, 5] = 7e4
    ops.split_planes(big.to(dev))
    assert ops.range_flag_tripped() and not ops.range_flag_tripped()


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("persist", [1, 8, 0])
def test_gemm_planes_linear_tiles(dev, tile, persist):
    """Every tile shape of the menu: ragged M and N, bias + GELU + residual, fp32 and plane outputs; persist = 8: eight persistent
    workgroups walk all tiles (the DMA stream crosses tile boundaries, counted waits behind an epilogue), 0: one tile per workgroup."""
    from chatterbox_amd import ops
    try:
        ops.lib.cbx_set_planes_tile(tile)
        ops.lib.cbx_set_planes_persist(persist)
        for (M, N, K) in [(1000, 1536, 256), (333, 80, 256), (129, 258, 512), (2048, 256, 1024)]:
            x, w, b, r = _r((M, K), 1), _r((N, K), 2, 1 / math.sqrt(K)), _r((N,), 3), _r((M, N), 4)
            xP, wP = ops.split_planes(x.to(dev)), ops.split_planes(w.to(dev))
            ref0 = F.linear(_planes_exact(x), _planes_exact(w), b.double())
            tol = 3e-5 * max(1.0, math.sqrt(K / 256))
            for act, fn in ((ops.NONE, lambda t: t), (ops.GELU_ERF, F.gelu), (ops.SILU, F.silu)):
                out = r.clone().to(dev)
                outP = ops.Planes(M, N, dev, zero=True)
                ops.linear_planes(xP, wP, out=out, outp=outP, bias=b.to(dev), act=act, residual=out)
                ref = fn(ref0) + r.double()
                _close(out, ref, tol, f"tile {tile} {M}x{N}x{K} act {act} (fp32 out)")
                _close(outP.float(), ref, tol, f"tile {tile} {M}x{N}x{K} act {act} (plane out)")
                assert (outP.float() - out).abs().max() <= 2.0 ** -21 * out.abs().max(), "plane output = split of the fp32 output"
    finally:
        ops.lib.cbx_set_planes_tile(0)
        ops.lib.cbx_set_planes_persist(1)


@pytest.mark.parametrize("tile", [0, 2, 3, 7, 11, 12, 15])
@pytest.mark.parametrize("persist", [8, 0])
def test_gemm_planes_conv_and_swapped_product(dev, tile, persist):
    """Causal Conv1d as implicit GEMM on planes (3 taps, left pad, batches, ragged lens), a column-range A operand, and the swapped
    (V^T) product with a per-batch W operand."""
    from chatterbox_amd import ops
    try:
        ops.lib.cbx_set_planes_tile(tile)
        ops.lib.cbx_set_planes_persist(persist)
        B, T, cin, N = 3, 517, 320, 256
        x, w, b = _r((B, T, cin), 1), _r((N, cin, 3), 2, 1 / math.sqrt(3 * cin)), _r((N,), 3)
        ref = F.conv1d(F.pad(_planes_exact(x).transpose(1, 2), (2, 0)), _planes_exact(w), b.double()).transpose(1, 2)
        wp = w.permute(0, 2, 1).reshape(N, 3 * cin).contiguous()  # tap-major packing (weights.pack_conv)
        wide = ops.Planes(B * T, 512, dev, zero=True)  # A is a column range (64 .. 384) of a wider planes tensor
        ops.split_planes(x.reshape(B * T, cin).to(dev), wide.cols(64, cin))
        out, outP = torch.empty(B, T, N, device=dev), ops.Planes(B * T, N, dev)
        ops.conv1d_planes(wide.cols(64, cin), ops.split_planes(wp.to(dev)), B=B, T=T, taps=3, cin=cin, out=out, outp=outP, bias=b.to(dev), pad_left=2)
        _close(out, ref, 6e-5, f"conv3 tile {tile}")
        _close(outP.float().view(B, T, N), ref, 6e-5, f"conv3 planes tile {tile}")
        # ragged input lengths: rows >= lens[z] read as zero
        lens = torch.tensor([517, 100, 3], dtype=torch.int32)
        xm = x.clone()
        for z in range(B):
            xm[z, lens[z]:] = 0
        refm = F.conv1d(F.pad(_planes_exact(xm).transpose(1, 2), (2, 0)), _planes_exact(w), b.double()).transpose(1, 2)
        ops.conv1d_planes(wide.cols(64, cin), ops.split_planes(wp.to(dev)), B=B, T=T, taps=3, cin=cin, out=out, bias=b.to(dev), pad_left=2,
                          lens=lens.to(dev))
        _close(out, refm, 6e-5, f"conv3 lens tile {tile}")
        # swapped product: VT[z] (D x T) = Wv (D x K) @ h[z]^T, written as planes with row stride >= T rounded up to 8
        D, K, Tp = 512, 256, (T + 7) // 8 * 8
        wv, h = _r((D, K), 5, 1 / 16), _r((B * T, K), 6)
        vt = ops.Planes(B * D, Tp, dev, zero=True)
        hP = ops.split_planes(h.to(dev))
        ops.gemm_planes(ops.split_planes(wv.to(dev)), hP, M=D, N=T - 1 if T % 2 else T, K=K, nz1=B, w_s1=T * hP.ld, P=vt, p_s1=D * vt.ld)
        n = T - 1 if T % 2 else T
        refv = torch.einsum("dk,ztk->zdt", _planes_exact(wv), _planes_exact(h).view(B, T, K))
        got = vt.float().view(B, D, Tp)
        _close(got[:, :, :n], refv[:, :, :n], 3e-5, f"swapped product tile {tile}")
        assert float(got[:, :, n:].abs().max()) == 0.0, "pad columns stay untouched"
    finally:
        ops.lib.cbx_set_planes_tile(0)
        ops.lib.cbx_set_planes_persist(1)


@pytest.mark.parametrize("tile", [0, 1, 4, 9, 14, 21])
@pytest.mark.parametrize("persist", [1, 8, 0])
def test_gemm_planes_transposed_column_range(dev, tile, persist):
    """to_q | to_k | to_v of a transformer block as ONE launch (ABI v8): columns below pt_n0 go to the q | k planes, the v columns are stored
    transposed per row group of T (the V^T operand of cbx_flash_attn_planes).  Against fp64 on the exact plane values, against the two-launch
    form (q | k Linear + swapped V^T product), pad columns of V^T untouched, ragged last row tile."""
    from chatterbox_amd import ops
    try:
        ops.lib.cbx_set_planes_tile(tile)
        ops.lib.cbx_set_planes_persist(persist)
        for Z, T in ((3, 200), (2, 1000), (1, 36)):
            M, K, N, n0 = Z * T, 256, 1536, 1024
            Tp = (T + 7) // 8 * 8
            h, w = _r((M, K), 1), _r((N, K), 2, 1 / math.sqrt(K))
            hP, wP = ops.split_planes(h.to(dev)), ops.split_planes(w.to(dev))
            qkP, vtP = ops.Planes(M, n0, dev), ops.Planes(Z * 512, Tp, dev, zero=True)
            ops.gemm_planes(hP, wP, M=M, N=N, K=K, P=qkP, PT=vtP, pt_n0=n0, pt_T=T, pt_zs=512 * vtP.ld)
            ref = F.linear(_planes_exact(h), _planes_exact(w))
            _close(qkP.float(), ref[:, :n0], 3e-5, f"q|k columns, tile {tile} Z {Z} T {T}")
            vt = vtP.float().view(Z, 512, Tp)
            _close(vt[:, :, :T], ref[:, n0:].view(Z, T, 512).transpose(1, 2), 3e-5, f"V^T, tile {tile} Z {Z} T {T}")
            assert float(vt[:, :, T:].abs().max() if Tp > T else 0.0) == 0.0, "pad keys of V^T stay zero"
            # the two-launch form computes the same products (cross terms in the other order): equal to rounding of the low accumulator
            qk2, vt2 = ops.Planes(M, n0, dev), ops.Planes(Z * 512, Tp, dev, zero=True)
            ops.linear_planes(hP, wP.rows_view(0, n0), outp=qk2)
            ops.gemm_planes(wP.rows_view(n0, 512), hP, M=512, N=T, K=K, nz1=Z, w_s1=T * hP.ld, P=vt2, p_s1=512 * vt2.ld)
            assert torch.equal(qkP.t, qk2.t), "q | k planes identical to the separate Linear"
            assert (vtP.float() - vt2.float()).abs().max() <= 1e-6, "V^T equal to the swapped product"
    finally:
        ops.lib.cbx_set_planes_tile(0)
        ops.lib.cbx_set_planes_persist(1)


DF_SHAPES = {"full": [(1000, 1536, 256), (2048, 256, 1024), (333, 320, 512)], "small": [(260, 384, 256), (200, 256, 512)]}


@pytest.mark.parametrize("tile,plain", [(41, 32), (42, 35)])
@pytest.mark.parametrize("persist", [2, 8, 1])
def test_gemm_planes_deferred_epilogue_equals_plain(dev, tile, plain, persist, shapes="full"):
    """The deferred-epilogue forms (round 6: a finished tile is folded and its epilogue rides on the next tile's K loop) against their plain twins, BIT FOR BIT:
    plane outputs with bias and every activation (q | k | v, ff1), the transposed column range with group boundaries inside a tile (pt_T below and above the tile
    height), fp32 outputs with an in-place residual (out-projection, ff2).  persist = 2 / 8: workgroups walk many tiles (folded tiles ride), 1: the chip-sized grid
    (on these shapes: one tile per workgroup -- everything goes through the drain)."""
    from chatterbox_amd import ops
    dv = dev
    try:
        ops.lib.cbx_set_planes_persist(persist)
        for (M, N, K) in DF_SHAPES[shapes]:
            x, w, b, r = _r((M, K), 1), _r((N, K), 2, 1 / math.sqrt(K)), _r((N,), 3), _r((M, N), 4)
            xP, wP = ops.split_planes(x.to(dv)), ops.split_planes(w.to(dv))
            for act in (ops.NONE, ops.GELU_ERF, ops.SILU):
                got, want = ops.Planes(M, N, dv, zero=True), ops.Planes(M, N, dv, zero=True)
                ops.gemm_planes(xP, wP, M=M, N=N, K=K, P=got, bias=b.to(dv), act=act, tile=tile)
                ops.gemm_planes(xP, wP, M=M, N=N, K=K, P=want, bias=b.to(dv), act=act, tile=plain)
                assert torch.equal(got.t, want.t), f"planes, tile {tile} vs {plain}, {M}x{N}x{K} act {act}"
                assert float(want.float().abs().max()) > 0
            for res in (False, True):
                got, want = r.clone().to(dv), r.clone().to(dv)
                ops.gemm_planes(xP, wP, M=M, N=N, K=K, C=got, ldc=N, R=got if res else None, ldr=N if res else 0, bias=b.to(dv), tile=tile)
                ops.gemm_planes(xP, wP, M=M, N=N, K=K, C=want, ldc=N, R=want if res else None, ldr=N if res else 0, bias=b.to(dv), tile=plain)
                assert torch.equal(got, want), f"fp32 out, tile {tile} vs {plain}, {M}x{N}x{K} residual {res}"
        for Z, T in ((5, 200), (10, 100), (3, 36)) if shapes == "full" else ((2, 132), (5, 36)):
            M, K, N, n0 = Z * T, 256, 1536, 1024
            Tp = (T + 7) // 8 * 8
            h, w = _r((M, K), 5), _r((N, K), 6, 1 / math.sqrt(K))
            hP, wP = ops.split_planes(h.to(dv)), ops.split_planes(w.to(dv))
            outs = []
            for t in (tile, plain):
                qkP, vtP = ops.Planes(M, n0, dv, zero=True), ops.Planes(Z * 512, Tp, dv, zero=True)
                ops.gemm_planes(hP, wP, M=M, N=N, K=K, P=qkP, PT=vtP, pt_n0=n0, pt_T=T, pt_zs=512 * vtP.ld, tile=t)
                outs.append((qkP.t, vtP.t))
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), f"q | k | V^T, tile {tile} vs {plain}, Z {Z} T {T}"
            assert float(outs[1][1].float().abs().max()) > 0
    finally:
        ops.lib.cbx_set_planes_persist(1)


def test_gemm_planes_transposed_rejects_bad_arguments(dev):
    from chatterbox_amd import ops
    h, w = ops.split_planes(_r((200, 256), 1).to(dev)), ops.split_planes(_r((1536, 256), 2).to(dev))
    qk, vt = ops.Planes(200, 1024, dev), ops.Planes(512, 104, dev, zero=True)
    with pytest.raises(RuntimeError, match="pt_T"):
        ops.gemm_planes(h, w, M=200, N=1536, K=256, P=qk, PT=vt, pt_n0=1024, pt_T=50, pt_zs=512 * vt.ld)   # 50 % 4 != 0
    with pytest.raises(RuntimeError, match="pt_n0"):
        ops.gemm_planes(h, w, M=200, N=1536, K=256, P=qk, PT=vt, pt_n0=1000, pt_T=100, pt_zs=512 * vt.ld)  # not a multiple of 256
    with pytest.raises(RuntimeError, match="plain Linear"):
        ops.gemm_planes(h, w, M=200, N=1536, K=256, P=qk, PT=vt, pt_n0=1024, pt_T=100, pt_zs=512 * vt.ld, act=ops.SILU)


def test_layernorm_planes(dev):
    from chatterbox_amd import ops
    M, C = 1003, 256
    x, w, b, tb = _r((M, C), 1) * 3 + 0.5, 1 + 0.1 * _r((C,), 2), 0.1 * _r((C,), 3), _r((C,), 4)
    out = ops.Planes(M, C, dev)
    ops.layernorm_planes(x.to(dev), w.to(dev), b.to(dev), out)
    ref = F.layer_norm(x.double(), (C,), w.double(), b.double(), 1e-5)
    _close(out.float(), ref, 2e-6, "layernorm planes")
    ops.layernorm_planes(x.to(dev), w.to(dev), b.to(dev), out, act=ops.MISH, post_add=tb.to(dev))
    _close(out.float(), F.mish(ref) + tb.double(), 3e-6, "layernorm + mish + time bias planes")
    plain = torch.empty(M, C, device=dev)
    ops.layernorm(x.to(dev), w.to(dev), b.to(dev), plain, 1e-5, act=ops.MISH, post_add=tb.to(dev))
    assert (out.float() - plain).abs().max() <= 2.0 ** -21 * plain.abs().max(), "planes = split of the fp32 LayerNorm kernel's result"


@pytest.fixture(params=[2, 1, 3, 4, 5, 6])
def attn_version(request):
    from chatterbox_amd import ops
    ops.lib.cbx_set_attn_planes_version(request.param)
    yield request.param
    ops.lib.cbx_set_attn_planes_version(0)  # the library's automatic choice


@pytest.mark.parametrize("Z,T,lens", [(2, 1000, None), (3, 517, [517, 130, 64]), (1, 64, None), (2, 200, [1, 199]), (1, 300, [0])])
def test_flash_attn_planes(dev, attn_version, Z, T, lens):
    """q, k as column ranges of one planes tensor, V^T from the swapped layout; ragged key lengths; vs fp64 softmax attention on the
    values the planes hold."""
    from chatterbox_amd import ops
    H, Tp = 8, (T + 7) // 8 * 8
    q, k, v = _r((Z, T, H, 64), 1) * 1.5, _r((Z, T, H, 64), 2) * 1.5, _r((Z, T, H, 64), 3)
    qk = ops.Planes(Z * T, 1024, dev)
    ops.split_planes(q.reshape(Z * T, 512).to(dev), qk.cols(0, 512))
    ops.split_planes(k.reshape(Z * T, 512).to(dev), qk.cols(512, 512))
    vt = ops.Planes(Z * 512, Tp, dev, zero=True)
    vtt = torch.zeros(Z, 512, Tp)
    vtt[:, :, :T] = v.reshape(Z, T, 512).transpose(1, 2)
    ops.split_planes(vtt.reshape(Z * 512, Tp).to(dev), vt)
    out = ops.Planes(Z * T, 512, dev)
    kl = None if lens is None else torch.tensor(lens, dtype=torch.int32, device=dev)
    ops.flash_attn_planes(qk.cols(0, 512), qk.cols(512, 512), vt, out, Z=Z, H=H, T=T, vt_sb=512 * vt.ld, scale=0.125, key_lens=kl)
    qe, ke, ve = _planes_exact(q), _planes_exact(k), _planes_exact(v)
    s = torch.einsum("zqhd,zkhd->zhqk", qe, ke) * 0.125
    if lens is not None:
        for z, n in enumerate(lens):
            s[z, :, :, n:] = -math.inf
    ref = torch.einsum("zhqk,zkhd->zqhd", torch.softmax(s, -1), ve).reshape(Z * T, 512)
    _close(out.float(), ref, 2e-5, f"flash attention planes Z={Z} T={T} lens={lens}")


def test_flash_attn_planes_forced_rescale(dev, attn_version):
    """A key whose score jumps far above everything seen before, placed in a late tile: the running-maximum rescale branch must fire
    and scale O, l exactly once (guide rule 26: the branch is data dependent and rare on random inputs)."""
    from chatterbox_amd import ops
    Z, T, H = 1, 512, 8
    q, k, v = _r((Z, T, H, 64), 1) * 0.3, _r((Z, T, H, 64), 2) * 0.3, _r((Z, T, H, 64), 3)
    k[0, 300] = q[0, 10] * 40.0  # raw score 40 |q|^2 for query 10 (and large for its neighbours in direction), tile 4
    qk = ops.Planes(Z * T, 1024, dev)
    ops.split_planes(q.reshape(Z * T, 512).to(dev), qk.cols(0, 512))
    ops.split_planes(k.reshape(Z * T, 512).to(dev), qk.cols(512, 512))
    vt = ops.split_planes(v.reshape(Z, T, 512).transpose(1, 2).reshape(512, T).contiguous().to(dev))
    out = ops.Planes(Z * T, 512, dev)
    ops.flash_attn_planes(qk.cols(0, 512), qk.cols(512, 512), vt, out, Z=Z, H=H, T=T, vt_sb=512 * vt.ld, scale=0.125)
    s = torch.einsum("zqhd,zkhd->zhqk", _planes_exact(q), _planes_exact(k)) * 0.125
    ref = torch.einsum("zhqk,zkhd->zqhd", torch.softmax(s, -1), _planes_exact(v)).reshape(Z * T, 512)
    _close(out.float(), ref, 2e-5, "rescale branch")


def test_estimator_planes_path_matches_fp32_operand_path(dev, monkeypatch):
    """The plane-format CFM estimator against the fp32-operand f16x3 path (CBX_PLANES = 0) of the same engine: the same arithmetic on the
    same operand values, so the mels agree far inside the parity tolerance; ragged batch of two."""
    from chatterbox_amd import synth
    from chatterbox_amd.s3gen import FlowEngine
    sd = synth.s3gen_state_dict(0, n_mid=2, n_enc=1, n_up_enc=1)
    eng = FlowEngine(sd, dev, precision=16)
    P, N = 24, 41
    ref = synth.s3gen_ref(n_prompt_tokens=P)
    toks = torch.stack([synth.speech_tokens(N, seed=1), synth.speech_tokens(N, seed=2)])
    lens = torch.tensor([N, N - 13])
    z = synth.randn((2, 2 * (P + N), 80), seed=5).to(dev)
    assert eng.use_planes
    mel_p = eng.inference(toks, lens, ref, z=z, n_steps=3).cpu()
    eng.use_planes = False
    mel_f = eng.inference(toks, lens, ref, z=z, n_steps=3).cpu()
    for b, n in enumerate(lens.tolist()):
        d = (mel_p[b, : 2 * n] - mel_f[b, : 2 * n]).abs()
        assert d.mean() <= 2e-6 and d.max() <= 3e-5, f"row {b}: mean {d.mean():.2e} max {d.max():.2e}"


@pytest.mark.parametrize("M,K,res", [(16000 + 37, 512, True), (333, 1024, True), (64, 256, False), (1000, 512, True)])
def test_gemm_planes_layernorm_epilogue(dev, M, K, res):
    """cbx_gemm_pl_t.ln_w (ABI v13): the row-spanning 64 x 256 tile writes the fp32 row (bias + residual) AND nn.LayerNorm of that row in plane format -- against
    fp64 on the values the planes hold, against the two launches it replaces (plain Linear: bit-identical fp32 row; cbx_layernorm_planes_f32 of that row: the
    LayerNorm tolerance), ragged M, with and without a residual."""
    from chatterbox_amd import ops
    N = 256
    x, w, b = _r((M, K), 1), _r((N, K), 2, 1 / math.sqrt(K)), _r((N,), 3)
    r = _r((M, N), 4) * 3.0 + 0.5  # a residual stream with a non-zero row mean
    lw, lb = 1 + 0.1 * _r((N,), 5), 0.1 * _r((N,), 6)
    xP, wP = ops.split_planes(x.to(dev)), ops.split_planes(w.to(dev))
    ref = F.linear(_planes_exact(x), _planes_exact(w), b.double()) + (r.double() if res else 0)
    out = r.clone().to(dev) if res else torch.empty(M, N, device=dev)
    lnP = ops.Planes(M, N, dev, zero=True)
    ops.linear_planes(xP, wP, out=out, bias=b.to(dev), residual=out if res else None, ln=(lw.to(dev), lb.to(dev)), lnp=lnP)
    _close(out, ref, 3e-5 * max(1.0, math.sqrt(K / 256)), f"fp32 row {M}x{K}")
    ln_ref = F.layer_norm(out.double().cpu(), (N,), lw.double(), lb.double(), 1e-5)
    _close(lnP.float(), ln_ref, 3e-6, f"LayerNorm epilogue {M}x{K}")
    # the two launches it replaces
    out2 = r.clone().to(dev) if res else torch.empty(M, N, device=dev)
    ops.linear_planes(xP, wP, out=out2, bias=b.to(dev), residual=out2 if res else None)
    assert torch.equal(out2, out), "the fp32 row must not depend on the tile form"
    ln2 = ops.Planes(M, N, dev, zero=True)
    ops.layernorm_planes(out2, lw.to(dev), lb.to(dev), ln2, 1e-5)
    assert (ln2.float() - lnP.float()).abs().max() <= 4e-6, f"vs layernorm_planes: {(ln2.float() - lnP.float()).abs().max():.3e}"


def test_range_flag_words_attribute_a_trip_to_the_batch_that_raised_it(dev):
    """ops.select_range_flag (round 5): launches report an operand outside the fp16 range into the flag word that was registered when they were ENQUEUED -- the
    throughput schedule gives consecutive batches alternating words, so a trip is attributed to the right batch although their launches overlap on the GPU."""
    from chatterbox_amd import ops
    ops.enable_range_flag(dev)
    try:
        ok, big = _r((64, 256), 1).to(dev), _r((64, 256), 2).to(dev)
        big[3, 5] = 7e4
        ops.select_range_flag(dev, 0)
        ops.split_planes(ok)
        ops.select_range_flag(dev, 1)
        ops.split_planes(big)
        ops.select_range_flag(dev, 0)
        ops.split_planes(ok)
        assert not ops.range_flag_tripped(dev, 0), "word 0 saw only in-range operands"
        assert ops.range_flag_tripped(dev, 1) and not ops.range_flag_tripped(dev, 1), "word 1 tripped once; reading clears it"
        assert not ops.range_flag_tripped(dev), "the registered word (0) is clean"
    finally:
        ops.select_range_flag(dev, 0)
