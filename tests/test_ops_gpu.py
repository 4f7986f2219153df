"""Kernel-level parity (-m gpu): every C-ABI op against the CPU oracle / plain torch fp32 on the same seeded inputs.

Tolerances: fp32-exact mode.  GEMM-class ops: |err| <= 2e-5 * (1 + |ref|) * sqrt(K/256) (fp32 accumulation-order
noise between an fmaf chain and MKL's blocked sgemm); softmax/attention: 2e-5 abs on O(1) outputs.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(got, ref, tol, what=""):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    err = (got - ref).abs()
    lim = tol * (1.0 + ref.abs())
    bad = err > lim
    assert not bad.any(), f"{what}: max err {err.max():.3e} (ref max {ref.abs().max():.3e}), {int(bad.sum())} / {bad.numel()} over tol {tol}"


def _r(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("M,N,K", [(1, 8, 16), (16, 1024, 1024), (37, 80, 192), (300, 1536, 256), (129, 257, 320), (1000, 64, 1024)])
def test_linear(dev, M, N, K):
    from chatterbox_amd import ops
    x, w, b, r = _r((M, K), 1), _r((N, K), 2, 1 / math.sqrt(K)), _r((N,), 3), _r((M, N), 4)
    for act, fn in ((ops.NONE, lambda t: t), (ops.SILU, F.silu), (ops.GELU_ERF, F.gelu), (ops.MISH, F.mish),
                    (ops.GELU_TANH, lambda t: F.gelu(t, approximate="tanh")), (ops.ELU, F.elu)):
        out = torch.empty(M, N, device=dev)
        ops.linear(x.to(dev), w.to(dev), out, bias=b.to(dev), act=act, residual=r.to(dev))
        _close(out, fn(F.linear(x, w, b)) + r, 3e-5 * max(1.0, math.sqrt(K / 256)), f"linear act={act}")


def test_linear_small_grid_form_is_bit_identical(dev):
    """Round 6: a handful of workgroups walking a long K take 64-wide K tiles (cbx_gemm_f32's dispatcher, 128 workgroups or fewer, K >= 1024).  The k order per output
    element does not change: the rows of a 65-row call are bit-identical to the same rows inside a 1200-row call (a grid that stays on the 16-wide K tiles)."""
    from chatterbox_amd import ops
    for N, K in ((1024, 4096), (3072, 1024)):
        x, w, b, r = _r((1200, K), 1).to(dev), _r((N, K), 2, 1 / math.sqrt(K)).to(dev), _r((N,), 3).to(dev), _r((1200, N), 4).to(dev)
        big, small = torch.empty(1200, N, device=dev), torch.empty(65, N, device=dev)
        ops.linear(x, w, big, bias=b, act=ops.GELU_TANH, residual=r)
        ops.linear(x[:65], w, small, bias=b, act=ops.GELU_TANH, residual=r[:65])
        assert torch.equal(small, big[:65]), f"N {N} K {K}: {(small - big[:65]).abs().max().item():.3e}"


def test_gelu_erf_accuracy(dev):
    """The branch-free erf of the GELU epilogue (cbx_common.h: cbx_gelu_erf) against fp64 on a dense grid over [-8, 8], the branch
    seam |x| / sqrt 2 = 1, the clamp at 4 and denormal-small inputs: |error| <= 1.5e-7 max(1, |gelu|), i.e. the rounding level of F.gelu in fp32."""
    from chatterbox_amd import ops
    x = torch.cat([torch.linspace(-8, 8, 1 << 20), torch.linspace(1.40, 1.43, 4096), torch.linspace(5.6, 5.7, 4096),
                   torch.tensor([0.0, 1e-30, -1e-30, 1e-8, 30.0, -30.0, 1e4, -1e4])])
    x = x[: x.numel() // 8 * 8].reshape(-1, 8).contiguous()
    out = torch.empty_like(x, device=dev)
    ops.act(x.to(dev), out, ops.GELU_ERF)
    ref = F.gelu(x.double())
    err = (out.cpu().double() - ref).abs()
    assert (err <= 1.5e-7 * ref.abs().clamp(min=1.0)).all(), f"gelu max abs err {err.max():.3e}"
    big = x.abs() > 6
    assert torch.equal(out.cpu()[big & (x > 0)], x[big & (x > 0)]) and (out.cpu()[big & (x < 0)].abs() == 0).all(), "saturation"
    f32 = (F.gelu(x) .double() - ref).abs().max()
    assert err.max() <= 2.5 * max(float(f32), 6e-8), f"not worse than 2.5x torch's own fp32 gelu ({f32:.3e})"


def test_linear_strided_accumulate_and_second_output(dev):
    from chatterbox_amd import ops
    M, N, K = 200, 96, 64
    xb, w = _r((M, K + 8), 1), _r((N, K), 2, 0.1)
    alpha_p = 1.0 + 0.2 * _r((N,), 5)
    cb = _r((M, N + 4), 6)
    out, out2 = cb.clone().to(dev), torch.empty(M, N, device=dev)
    ops.linear(xb.to(dev)[:, :K], w.to(dev), out[:, :N], alpha=1.0 / 3, beta=1.0, out2=out2, act2=ops.SNAKE,
               act2_param=alpha_p.to(dev))
    ref = cb[:, :N] + F.linear(xb[:, :K], w) / 3
    _close(out[:, :N], ref, 3e-5, "accumulate")
    _close(out[:, N:], cb[:, N:], 0.0, "untouched pad")
    a = alpha_p[None]
    _close(out2, ref + (1.0 / (a + 1e-9)) * torch.sin(ref * a) ** 2, 5e-5, "snake second output")


def test_swiglu(dev):
    from chatterbox_amd import ops, weights
    for M in (16, 200):
        D, Fh = 256, 512
        x, g, u = _r((M, D), 1), _r((Fh, D), 2, 0.06), _r((Fh, D), 3, 0.06)
        out = torch.empty(M, Fh, device=dev)
        ops.linear(x.to(dev), weights.pack_swiglu(g, u).to(dev), out, swiglu=True)
        _close(out, F.silu(F.linear(x, g)) * F.linear(x, u), 3e-5, "swiglu")


@pytest.mark.parametrize("cin,cout,k,dil,stride,pad,T", [(32, 48, 3, 1, 1, 1, 77), (80, 512, 7, 1, 1, 3, 60), (64, 64, 11, 5, 1, 25, 300),
                                                         (32, 256, 30, 1, 15, 7, 1201), (32, 128, 6, 1, 3, 1, 241), (320, 256, 3, 1, 1, 2, 100)])
def test_conv1d(dev, cin, cout, k, dil, stride, pad, T):
    from chatterbox_amd import ops, weights
    B = 3
    x, w, b = _r((B, cin, T), 1), _r((cout, cin, k), 2, 1 / math.sqrt(cin * k)), _r((cout,), 3)
    causal = (pad == k - 1 and dil == 1 and cin == 320)
    ref = F.conv1d(F.pad(x, (pad, 0)) if causal else x, w, b, stride=stride, dilation=dil, padding=0 if causal else pad)
    Tout = ref.shape[2]
    out = torch.empty(B, Tout, cout, device=dev)
    ops.conv1d(x.transpose(1, 2).contiguous().to(dev), weights.pack_conv(w).to(dev), out, taps=k, cin=cin, bias=b.to(dev),
               dil=dil, stride=stride, pad_left=pad)
    _close(out.transpose(1, 2), ref, 4e-5, "conv1d")


def test_conv1d_ragged_and_upsample(dev):
    from chatterbox_amd import ops, weights
    B, C, T = 3, 32, 50
    lens = torch.tensor([50, 31, 7], dtype=torch.int32)
    x, w, b = _r((B, C, T), 1), _r((C, C, 4), 2, 0.1), _r((C,), 3)
    out = torch.empty(B, T, C, device=dev)
    # look-ahead conv (right pad 3) must see zeros beyond each row's own length
    ops.conv1d(x.transpose(1, 2).contiguous().to(dev), weights.pack_conv(w).to(dev), out, taps=4, cin=C, bias=b.to(dev),
               pad_left=0, lens=lens.to(dev))
    for i in range(B):
        n = int(lens[i])
        ref = F.conv1d(F.pad(x[i:i + 1, :, :n], (0, 3)), w, b)
        _close(out[i, :n].t(), ref[0], 3e-5, f"ragged row {i}")
    # nearest x2 upsample + left pad 4 + k5 (Upsample1D of the conformer encoder)
    w5 = _r((C, C, 5), 4, 0.1)
    ref = F.conv1d(F.pad(x.repeat_interleave(2, dim=2), (4, 0)), w5, b)
    out = torch.empty(B, 2 * T, C, device=dev)
    ops.conv1d(x.transpose(1, 2).contiguous().to(dev), weights.pack_conv(w5).to(dev), out, taps=5, cin=C, bias=b.to(dev),
               pad_left=4, up=2)
    _close(out.transpose(1, 2), ref, 3e-5, "upsample conv")


@pytest.mark.parametrize("cin,cout,k,s,p", [(64, 32, 16, 8, 4), (32, 16, 11, 5, 3), (32, 64, 7, 3, 2)])
def test_conv_transpose(dev, cin, cout, k, s, p):
    from chatterbox_amd import ops, weights
    B, T = 2, 41
    x, w, b = _r((B, cin, T), 1), _r((cin, cout, k), 2, 0.1), _r((cout,), 3)
    ref = F.conv_transpose1d(x, w, b, stride=s, padding=p)
    wp, bp = weights.pack_conv_transpose(w, b, s, p)
    out = torch.empty(B, T, s * cout, device=dev)
    ops.conv1d(x.transpose(1, 2).contiguous().to(dev), wp.to(dev), out, taps=3, cin=cin, bias=bp.to(dev), pad_left=1)
    _close(out.view(B, T * s, cout).transpose(1, 2), ref, 3e-5, "conv_transpose")


def test_bmm(dev):
    from chatterbox_amd import ops
    Z1, Z2, M, K, N = 2, 3, 150, 64, 149
    a, b = _r((Z1, M, Z2, K), 1), _r((Z1, N, Z2, K), 2)
    out = torch.empty(Z1, Z2, M, 152, device=dev)
    ops.bmm(a.to(dev).permute(0, 2, 1, 3), b.to(dev).permute(0, 2, 1, 3), out[..., :N])
    _close(out[..., :N], torch.einsum("zmhk,znhk->zhmn", a, b), 3e-5, "bmm nt")
    # nn: P (Z1,Z2,M,Kp) @ V (Z1,Z2,Kp,64) with Kp = 150 (not a multiple of 16), V strided
    pr, v = _r((Z1, Z2, M, 152), 3), _r((Z1, 150, Z2, 64), 4)
    pr[..., 150:] = 0
    o = torch.empty(Z1, M, Z2, 64, device=dev)
    ops.bmm(pr.to(dev)[..., :150], v.to(dev).permute(0, 2, 1, 3), o.permute(0, 2, 1, 3), nn=True)
    _close(o, torch.einsum("zhmk,zkhd->zmhd", pr[..., :150], v), 3e-5, "bmm nn")


def test_layernorm_rmsnorm(dev):
    from chatterbox_amd import ops
    from oracle import ref_torch as O
    for C, eps in ((256, 1e-5), (512, 1e-12), (1024, 1e-5), (80, 1e-5)):
        x, w, b, pa = _r((333, C), 1, 3.0) + 0.5, 1 + 0.1 * _r((C,), 2), 0.1 * _r((C,), 3), _r((C,), 4)
        out = torch.empty(333, C, device=dev)
        ops.layernorm(x.to(dev), w.to(dev), b.to(dev), out, eps)
        _close(out, F.layer_norm(x, (C,), w, b, eps), 2e-5, "layernorm")
        ops.layernorm(x.to(dev), w.to(dev), b.to(dev), out, eps, act=ops.MISH, post_add=pa.to(dev))
        _close(out, F.mish(F.layer_norm(x, (C,), w, b, eps)) + pa, 2e-5, "layernorm+mish+add")
        ops.layernorm(x.to(dev), w.to(dev), None, out, 1e-5, rms=True)
        _close(out, O.rms_norm(x, w), 2e-5, "rmsnorm")


@pytest.mark.parametrize("Tq,Tk,causal", [(200, 200, False), (1000, 1000, False), (103, 103, True), (64, 64, True), (130, 130, False)])
def test_flash_attn(dev, Tq, Tk, causal):
    from chatterbox_amd import ops
    Z, H = 3, 4
    qkv = _r((Z, Tq, 3, H, 64), 1)
    lens = torch.tensor([Tk, max(1, Tk - 37), max(1, Tk // 3)], dtype=torch.int32)
    q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
    if causal:
        ref = F.scaled_dot_product_attention(q, k, v, is_causal=True)
        kl = None
    else:
        bias = torch.zeros(Z, 1, 1, Tk)
        for z in range(Z):
            bias[z, ..., int(lens[z]):] = -1e10
        ref = F.scaled_dot_product_attention(q, k, v, attn_mask=bias)
        kl = lens.to(dev)
    d = qkv.to(dev)
    out = torch.empty(Z, Tq, H, 64, device=dev)
    ops.flash_attn(d[:, :, 0], d[:, :, 1], d[:, :, 2], out, 0.125, key_lens=kl, causal=causal)
    _close(out, ref.transpose(1, 2), 2e-5, "flash_attn")


def test_flash_attn_keys_from_the_kv_cache(dev):
    """cbx_flash_attn_kv_f32 (ABI v15): K / V with their own head strides -- views of a [row][head][max_ctx][64] cache -- and causal attention with fewer queries than
    keys (the queries are the LAST Tq positions: the prefill of T3's text positions behind a cached conditioning prefix).  Bit-identical to the same attention over the
    token-major copy of the keys (cbx_flash_attn_f32) restricted to those queries, and close to torch's SDPA."""
    from chatterbox_amd import ops
    Z, H, Tk, Tq, ctx = 3, 4, 103, 69, 128
    qkv = _r((Z, Tk, 3, H, 64), 1)
    d = qkv.to(dev)
    kc, vc = torch.zeros(Z, H, ctx, 64, device=dev), torch.zeros(Z, H, ctx, 64, device=dev)
    kc[:, :, :Tk], vc[:, :, :Tk] = d[:, :, 1].transpose(1, 2), d[:, :, 2].transpose(1, 2)
    full = torch.empty(Z, Tk, H, 64, device=dev)
    ops.flash_attn(d[:, :, 0], d[:, :, 1], d[:, :, 2], full, 0.125, causal=True)
    out = torch.empty(Z, Tq, H, 64, device=dev)
    ops.flash_attn(d[:, Tk - Tq:, 0], kc[:, :, :Tk].permute(0, 2, 1, 3), vc[:, :, :Tk].permute(0, 2, 1, 3), out, 0.125, causal=True)
    assert torch.equal(out, full[:, Tk - Tq:]), "queries behind a prefix, keys from the cache == the same rows of the full causal attention"
    q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
    ref = F.scaled_dot_product_attention(q, k, v, is_causal=True)
    _close(out, ref.transpose(1, 2)[:, Tk - Tq:], 2e-5, "flash_attn over the KV cache")


# ---- split-bf16 modes (cbx_gemm_t.precision 3 / 6, cbx_flash_attn_split_f32): fp32 operands rebuilt from bf16 planes on
# the 16x faster bf16 matrix cores.  Stated tolerances: precision 6 = the fp32 tolerances above; precision 3 = 1e-4.
_SPLIT_TOL = {6: 3e-5, 16: 3e-5, 3: 1e-4}  # 16 = f16x3 (two fp16 planes, second one scaled): fp32-level like 6


@pytest.mark.parametrize("prec", [3, 6, 16])
def test_split_gemm_linear(dev, prec):
    from chatterbox_amd import ops
    with ops.gemm_precision(prec):
        for (M, N, K) in [(1000, 256, 256), (129, 257, 320), (300, 1536, 256), (4000, 80, 1000), (513, 1024, 80), (33, 100, 36)]:
            x, w, b, r = _r((M, K), 1), _r((N, K), 2, 1 / math.sqrt(K)), _r((N,), 3), _r((M, N), 4)
            out = torch.empty(M, N, device=dev)
            ops.linear(x.to(dev), w.to(dev), out, bias=b.to(dev), act=ops.GELU_ERF, residual=r.to(dev))
            _close(out, F.gelu(F.linear(x, w, b)) + r, _SPLIT_TOL[prec] * max(1.0, math.sqrt(K / 256)), f"split linear {M}x{N}x{K}")
        # accumulate (beta), second output, strided views with untouched pad columns
        M, N, K = 200, 96, 64
        xb, w, ap, cb = _r((M, K + 8), 1), _r((N, K), 2, 0.1), 1.0 + 0.2 * _r((N,), 5), _r((M, N + 4), 6)
        out, out2 = cb.clone().to(dev), torch.empty(M, N, device=dev)
        ops.linear(xb.to(dev)[:, :K], w.to(dev), out[:, :N], alpha=1.0 / 3, beta=1.0, out2=out2, act2=ops.SNAKE, act2_param=ap.to(dev))
        ref = cb[:, :N] + F.linear(xb[:, :K], w) / 3
        _close(out[:, :N], ref, _SPLIT_TOL[prec], "split accumulate")
        _close(out[:, N:], cb[:, N:], 0.0, "untouched pad")
        _close(out2, ref + (1.0 / (ap[None] + 1e-9)) * torch.sin(ref * ap[None]) ** 2, 2 * _SPLIT_TOL[prec], "split second output")


@pytest.mark.parametrize("prec", [3, 6, 16])
def test_split_gemm_conv(dev, prec):
    from chatterbox_amd import ops, weights
    B = 3
    with ops.gemm_precision(prec):
        for (cin, cout, k, dil, stride, pad, T) in [(32, 48, 3, 1, 1, 1, 77), (64, 64, 11, 5, 1, 25, 300), (32, 256, 30, 1, 15, 7, 1201),
                                                    (320, 256, 3, 1, 1, 2, 100), (512, 512, 3, 1, 1, 1, 64), (80, 512, 7, 1, 1, 3, 60)]:
            x, w, b = _r((B, cin, T), 1), _r((cout, cin, k), 2, 1 / math.sqrt(cin * k)), _r((cout,), 3)
            causal = (pad == k - 1 and dil == 1 and cin == 320)
            ref = F.conv1d(F.pad(x, (pad, 0)) if causal else x, w, b, stride=stride, dilation=dil, padding=0 if causal else pad)
            out = torch.empty(B, ref.shape[2], cout, device=dev)
            ops.conv1d(x.transpose(1, 2).contiguous().to(dev), weights.pack_conv(w).to(dev), out, taps=k, cin=cin, bias=b.to(dev),
                       dil=dil, stride=stride, pad_left=pad)
            _close(out.transpose(1, 2), ref, 1.5 * _SPLIT_TOL[prec], f"split conv cin{cin} k{k}")  # cin 80: falls back to exact
        C, T = 32, 50
        lens = torch.tensor([50, 31, 7], dtype=torch.int32)
        x, w, b = _r((B, C, T), 1), _r((C, C, 4), 2, 0.1), _r((C,), 3)
        out = torch.empty(B, T, C, device=dev)
        ops.conv1d(x.transpose(1, 2).contiguous().to(dev), weights.pack_conv(w).to(dev), out, taps=4, cin=C, bias=b.to(dev), pad_left=0,
                   lens=lens.to(dev))
        for i in range(B):
            n = int(lens[i])
            _close(out[i, :n].t(), F.conv1d(F.pad(x[i:i + 1, :, :n], (0, 3)), w, b)[0], _SPLIT_TOL[prec], f"split ragged row {i}")
        w5 = _r((C, C, 5), 4, 0.1)
        ref = F.conv1d(F.pad(x.repeat_interleave(2, dim=2), (4, 0)), w5, b)
        out = torch.empty(B, 2 * T, C, device=dev)
        ops.conv1d(x.transpose(1, 2).contiguous().to(dev), weights.pack_conv(w5).to(dev), out, taps=5, cin=C, bias=b.to(dev), pad_left=4, up=2)
        _close(out.transpose(1, 2), ref, _SPLIT_TOL[prec], "split upsample conv")


@pytest.mark.parametrize("prec", [3, 6, 16])
@pytest.mark.parametrize("Tq,Tk,causal", [(200, 200, False), (1000, 1000, False), (103, 103, True), (64, 64, True), (130, 130, False)])
def test_split_flash_attn(dev, Tq, Tk, causal, prec):
    from chatterbox_amd import ops
    Z, H = 3, 4
    qkv = _r((Z, Tq, 3, H, 64), 1)
    lens = torch.tensor([Tk, max(1, Tk - 37), max(1, Tk // 3)], dtype=torch.int32)
    q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
    if causal:
        ref, kl = F.scaled_dot_product_attention(q, k, v, is_causal=True), None
    else:
        bias = torch.zeros(Z, 1, 1, Tk)
        for z in range(Z):
            bias[z, ..., int(lens[z]):] = -1e10
        ref, kl = F.scaled_dot_product_attention(q, k, v, attn_mask=bias), lens.to(dev)
    d = qkv.to(dev)
    out = torch.empty(Z, Tq, H, 64, device=dev)
    with ops.gemm_precision(prec):
        ops.flash_attn(d[:, :, 0], d[:, :, 1], d[:, :, 2], out, 0.125, key_lens=kl, causal=causal)
    _close(out, ref.transpose(1, 2), {6: 2e-5, 16: 2e-5, 3: 1e-4}[prec], f"split flash_attn p{prec}")


def test_layernorm_folded_into_linear(dev):
    """cbx_row_stats_f32 + cbx_gemm_t.ln_*: LayerNorm applied to the A operand inside the f16x3 Linear == layernorm() followed by
    linear() (same statistics kernel code, same normalisation expression), and within the split tolerance of torch; unsupported
    configurations are refused loudly, never silently un-normalised."""
    from chatterbox_amd import ops
    from chatterbox_amd._lib import CbxError
    K = 256
    for (M, N) in [(16000, 1536), (1000, 1024), (333, 80)]:
        x, w, b = _r((M, K + 4), 1, 3.0) + 1.5, _r((N, K), 2, 1 / math.sqrt(K)), _r((N,), 3)
        g, be = 1.0 + 0.3 * _r((K,), 4), _r((K,), 5)
        xd, wd, gd, bd = x.to(dev)[:, :K], w.to(dev), g.to(dev), be.to(dev)
        stats = ops.row_stats(xd, torch.empty(M, 2, device=dev))
        mu, var = x[:, :K].mean(1), x[:, :K].var(1, unbiased=False)
        _close(stats[:, 0], mu, 2e-6, "row mean")
        _close(stats[:, 1], (var + 1e-5).rsqrt(), 2e-6, "row rstd")
        fused, plain, h = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev), torch.empty(M, K, device=dev)
        with ops.gemm_precision(16):
            assert ops.ln_fusable(M, K)
            ops.linear(xd, wd, fused, bias=b.to(dev), act=ops.GELU_ERF, ln=(stats, gd, bd))
            ops.layernorm(xd, gd, bd, h, 1e-5)
            ops.linear(h, wd, plain, bias=b.to(dev), act=ops.GELU_ERF)
        _close(fused, plain.cpu(), 2e-6, f"folded LN == LN + linear {M}x{N}")
        _close(fused, F.gelu(F.linear(F.layer_norm(x[:, :K], (K,), g, be, 1e-5), w, b)), _SPLIT_TOL[16], "folded LN vs torch")
    with ops.gemm_precision(6):  # only the f16x3 kernel carries the fold: any other mode must refuse, not ignore it
        assert not ops.ln_fusable(1000, K)
        with pytest.raises(CbxError):
            ops.linear(xd, wd, fused, ln=(stats, gd, bd))
    with ops.gemm_precision(16), pytest.raises(CbxError):
        ops.linear(xd[:20], wd, fused[:20], ln=(stats, gd, bd))  # M <= 32: the skinny exact kernel


def test_f16x3_accuracy_and_range_flag(dev):
    """precision 16 against an fp64 product: as close as the exact fp32 MFMA kernel over 8 decades of operand scale; operands beyond
    the fp16 range raise the device flag (and only they do)."""
    from chatterbox_amd import ops
    M, N, K = 512, 256, 512
    ops.enable_range_flag(dev)
    assert not ops.range_flag_tripped()
    for sa, sw in [(1.0, 0.05), (300.0, 1.0), (1e-3, 1e-2), (3e4 / 5, 1.0)]:
        x, w = _r((M, K), 1, sa), _r((N, K), 2, sw)
        ref = x.double() @ w.double().t()
        errs = {}
        for prec in (1, 16, 3):
            out = torch.empty(M, N, device=dev)
            with ops.gemm_precision(prec):
                ops.linear(x.to(dev), w.to(dev), out)
            errs[prec] = float((out.cpu().double() - ref).abs().mean() / ref.abs().mean())
        assert errs[16] <= 1.5 * errs[1] + 1e-9, (sa, sw, errs)
        assert errs[16] < 0.2 * errs[3], (sa, sw, errs)
    assert not ops.range_flag_tripped()
    x = _r((M, K), 1)
    x[17, 33] = 7.0e4
    with ops.gemm_precision(16):
        ops.linear(x.to(dev), _r((N, K), 2).to(dev), torch.empty(M, N, device=dev))
    assert ops.range_flag_tripped()
    assert not ops.range_flag_tripped()  # reading clears it
    q = _r((1, 128, 3, 2, 64), 3)
    q[0, 5, 1, 0, 7] = -1e5  # a key
    d, out = q.to(dev), torch.empty(1, 128, 2, 64, device=dev)
    with ops.gemm_precision(16):
        ops.flash_attn(d[:, :, 0], d[:, :, 1], d[:, :, 2], out, 0.125)
    assert ops.range_flag_tripped()


def test_decode_attn(dev):
    from chatterbox_amd import ops
    rows, H, maxp = 6, 16, 700
    kc, vc, q = _r((rows, H, maxp, 64), 1), _r((rows, H, maxp, 64), 2), _r((rows, H * 64), 3)
    ctx = torch.tensor([1, 5, 64, 129, 333, 700], dtype=torch.int32)
    out = torch.empty(rows, H * 64, device=dev)
    ops.decode_attn(q.to(dev), kc.to(dev), vc.to(dev), out, ctx.to(dev), 0.125)
    for r in range(rows):
        n = int(ctx[r])
        ref = F.scaled_dot_product_attention(q[r].view(H, 1, 64), kc[r, :, :n], vc[r, :, :n])
        _close(out[r].view(H, 64), ref[:, 0], 2e-5, f"decode_attn row {r}")


def test_softmax_relpos(dev):
    from chatterbox_amd import ops
    Z1, Z2, T = 2, 8, 75
    ac, bd = _r((Z1, Z2, T, T), 1, 3.0), _r((Z1, Z2, T, 2 * T - 1), 2, 3.0)
    lens = torch.tensor([T, 40], dtype=torch.int32)
    p = torch.full((Z1, Z2, T, 76), 7.0, device=dev)
    ops.softmax_relpos(ac.to(dev), bd.to(dev), p, 0.125, key_lens=lens.to(dev))
    idx = T - 1 - torch.arange(T)[:, None] + torch.arange(T)[None]
    sc = (ac + torch.gather(bd, 3, idx.expand(Z1, Z2, T, T))) / 8
    m = torch.arange(T)[None, :] >= lens[:, None]
    ref = torch.softmax(sc.masked_fill(m[:, None, None], float("-inf")), -1).masked_fill(m[:, None, None], 0)
    _close(p[..., :T], ref, 1e-5, "softmax_relpos")
    assert float(p[..., T:].abs().max()) == 0.0


def test_rope_and_kv_append(dev):
    from chatterbox_amd import ops
    from oracle import ref_torch as O
    rows, H, maxp = 5, 16, 64
    qkv = _r((rows, 3 * H * 64), 1)
    pos = torch.tensor([0, 3, 17, 40, 63], dtype=torch.int32)
    cos, sin = O.rope_cos_sin(torch.arange(maxp), O.llama3_inv_freq())
    kc, vc = torch.zeros(rows, H, maxp, 64, device=dev), torch.zeros(rows, H, maxp, 64, device=dev)
    d = qkv.clone().to(dev)
    ops.rope_kv(d, pos.to(dev), cos.to(dev), sin.to(dev), kc, vc, H)
    q, k, v = (qkv.view(rows, 3, H, 64)[:, i] for i in range(3))
    c, s = cos[pos.long()][:, None], sin[pos.long()][:, None]
    qr, kr = q * c + O._rot_half(q) * s, k * c + O._rot_half(k) * s
    _close(d.view(rows, 3, H, 64)[:, 0], qr, 1e-6, "rope q")
    _close(d.view(rows, 3, H, 64)[:, 1], kr, 1e-6, "rope k")
    for r in range(rows):
        _close(kc[r, :, int(pos[r])], kr[r], 1e-6, "k cache")
        _close(vc[r, :, int(pos[r])], v[r], 0.0, "v cache")


def test_elementwise(dev):
    from chatterbox_amd import ops
    x, a = _r((100, 64), 1, 2.0), 1 + 0.2 * _r((64,), 2)
    out = torch.empty(100, 64, device=dev)
    ops.act(x.to(dev), out, ops.SNAKE, param=a.to(dev))
    _close(out, x + (1 / (a + 1e-9)) * torch.sin(x * a) ** 2, 3e-6, "snake")
    ops.act(x.to(dev), out, ops.LRELU, slope=0.1)
    _close(out, F.leaky_relu(x, 0.1), 0.0, "lrelu")
    tab, tab2 = _r((50, 64), 3), _r((20, 64), 4)
    ids, ids2 = torch.tensor([3, 49, 0, 7], dtype=torch.int64), torch.tensor([0, 1, 19, 5], dtype=torch.int32)
    o = torch.empty(4, 64, device=dev)
    ops.embed(ids.to(dev), tab.to(dev), o, table2=tab2.to(dev), ids2=ids2.to(dev))
    _close(o, tab[ids] + tab2[ids2.long()], 0.0, "embed")
    # CFM Euler + CFG on the packed estimator input
    B, T = 2, 33
    xin, v = _r((2 * B, T, 320), 5), _r((2 * B, T, 80), 6)
    dx = xin.clone().to(dev)
    ops.cfm_euler(dx, v.to(dev), B, T, 80, 0.07, 0.7)
    xn = xin[:B, :, :80] + 0.07 * (1.7 * v[:B] - 0.7 * v[B:])
    _close(dx[:B, :, :80], xn, 2e-6, "euler cond rows")
    _close(dx[B:, :, :80], xn, 2e-6, "euler uncond rows")
    _close(dx[:, :, 80:], xin[:, :, 80:], 0.0, "euler leaves mu/spk/cond")


@pytest.mark.parametrize("top_p,min_p", [(1.0, 0.05), (0.9, 0.0), (0.8, 0.05)])
def test_sampler(dev, top_p, min_p):
    from chatterbox_amd import ops
    from oracle import ref_torch as O
    B, V, steps = 3, 8194, 6
    logits = _r((steps, 2 * B, V), 1, 2.0)
    u = torch.rand(B, steps, generator=torch.Generator().manual_seed(5))
    d = dict(seen=torch.zeros(B, V, dtype=torch.uint8, device=dev), step=torch.zeros(B, dtype=torch.int32, device=dev),
             out_tokens=torch.zeros(B, steps, dtype=torch.int64, device=dev), done=torch.zeros(B, dtype=torch.int32, device=dev),
             n_generated=torch.zeros(B, dtype=torch.int32, device=dev), next_ids=torch.zeros(2 * B, dtype=torch.int64, device=dev),
             next_pos_ids=torch.zeros(2 * B, dtype=torch.int32, device=dev), positions=torch.full((2 * B,), 9, dtype=torch.int32, device=dev),
             ctx_lens=torch.full((2 * B,), 10, dtype=torch.int32, device=dev))
    d["seen"][:, O.START_SPEECH] = 1
    ud = u.to(dev)
    gen = [[O.START_SPEECH] for _ in range(B)]
    for s in range(steps):
        ld = logits[s].to(dev)
        ops.t3_sample(logits=ld, ld=V, V=V, B=B, cfg=1, cfg_weight=0.5, temperature=0.8, min_p=min_p, top_p=top_p,
                      rep_penalty=1.2, top_k=0, order=0, ban_token=O.STOP_SPEECH, eos_token=O.STOP_SPEECH, uniforms=ud,
                      max_steps=steps, **d)
        toks = d["out_tokens"][:, s].cpu()
        for b in range(B):
            l = O.process_logits(logits[s, b], logits[s, B + b], torch.tensor(gen[b]), 0.5, 0.8, min_p, top_p, 1.2)
            pr = torch.softmax(l, -1)
            pr[O.STOP_SPEECH] = 0
            ref = O.sample_inverse_cdf(pr, u[b, s])
            assert int(toks[b]) == ref, f"step {s} utt {b}: got {int(toks[b])} want {ref}"
            gen[b].append(ref)
    assert d["positions"].tolist() == [9 + steps] * (2 * B) and d["ctx_lens"].tolist() == [10 + steps] * (2 * B)
    assert d["next_pos_ids"].tolist() == [steps] * (2 * B) and d["step"].tolist() == [steps] * B


def test_sampler_all_surviving_ids_banned(dev):
    """ban_from on a peaked distribution: the arg-max is a banned id and min-p prunes everything else, so no probability mass is
    left.  Defined result (include/cbx.h): the allowed id with the largest CFG-combined raw logit; never an out-of-range token."""
    from chatterbox_amd import ops
    B, V, steps = 2, 8194, 1
    logits = _r((2 * B, V), 1, 0.5)
    logits[:, 7000] = 30.0            # banned arg-max (>= ban_from) in both CFG rows
    logits[0, 1234] = logits[B + 0, 1234] = 9.0   # best allowed id of utterance 0
    logits[1, 42] = logits[B + 1, 42] = 8.0       # ... of utterance 1
    d = dict(seen=torch.zeros(B, V, dtype=torch.uint8, device=dev), step=torch.zeros(B, dtype=torch.int32, device=dev),
             out_tokens=torch.zeros(B, steps, dtype=torch.int64, device=dev), done=torch.zeros(B, dtype=torch.int32, device=dev),
             n_generated=torch.zeros(B, dtype=torch.int32, device=dev), next_ids=torch.zeros(2 * B, dtype=torch.int64, device=dev),
             next_pos_ids=torch.zeros(2 * B, dtype=torch.int32, device=dev), positions=torch.zeros(2 * B, dtype=torch.int32, device=dev),
             ctx_lens=torch.zeros(2 * B, dtype=torch.int32, device=dev))
    guard = d["seen"].clone()
    ops.t3_sample(logits=logits.to(dev), ld=V, V=V, B=B, cfg=1, cfg_weight=0.5, temperature=0.8, min_p=0.05, top_p=1.0, rep_penalty=1.2,
                  top_k=0, order=0, ban_token=6562, eos_token=6562, ban_from=6561, uniforms=torch.full((B, steps), 0.5, device=dev),
                  max_steps=steps, **d)
    assert d["out_tokens"][:, 0].tolist() == [1234, 42]
    guard[0, 1234] = guard[1, 42] = 1
    assert torch.equal(d["seen"], guard) and d["next_ids"].tolist() == [1234, 42, 1234, 42]


def test_hift_source_stft_istft(dev):
    from chatterbox_amd import ops
    from oracle import ref_torch as O
    B, T = 2, 40
    f0 = torch.rand(B, T, generator=torch.Generator().manual_seed(1)) * 300
    f0[:, 5:9] = 3.0  # unvoiced stretch
    phase = (torch.rand(B, 9, 1, generator=torch.Generator().manual_seed(2)) * 2 - 1) * math.pi
    phase[:, 0] = 0
    noise = _r((B, 9, 480 * T), 3)
    sd = {"mel2wav.m_source.l_linear.weight": _r((1, 9), 4, 0.5), "mel2wav.m_source.l_linear.bias": torch.tensor([0.05])}
    ref = O.source_module(sd, f0, phase, noise)[:, 0]
    s = torch.empty(B, 480 * T, device=dev)
    cum = torch.empty(B, 9, T, dtype=torch.float64, device=dev)
    ops.hift_source(f0.to(dev), phase.to(dev), noise.to(dev), sd["mel2wav.m_source.l_linear.weight"].to(dev), 0.05, s, cum)
    _close(s, ref, 2e-5, "hift source")
    # STFT
    win = torch.hann_window(16, periodic=True)
    sp = torch.stft(ref, 16, 4, 16, window=win, return_complex=True)
    spec = torch.empty(B, 120 * T + 1, 32, device=dev)
    ops.hift_stft(ref.to(dev), spec)
    _close(spec[:, :, :9], sp.real.transpose(1, 2), 1e-5, "stft re")
    _close(spec[:, :, 9:18], sp.imag.transpose(1, 2), 1e-5, "stft im")
    assert float(spec[:, :, 18:].abs().max()) == 0.0
    # iSTFT head
    x = _r((B, 18, 120 * T + 1), 5)
    x[:, :9] = x[:, :9] * 1.5 - 1.0
    mag, ph = torch.exp(x[:, :9]).clip(max=1e2), torch.sin(x[:, 9:])
    wref = torch.istft(torch.complex(mag * torch.cos(ph), mag * torch.sin(ph)), 16, 4, 16, window=win).clamp(-0.99, 0.99)
    xd = torch.zeros(B, 120 * T + 1, 32, device=dev)
    xd[:, :, :18] = x.transpose(1, 2).to(dev)
    wav = torch.empty(B, 480 * T, device=dev)
    ops.hift_istft(xd, wav)
    _close(wav, wref, 1e-5, "istft")
    ops.hift_istft(xd, wav, fade_n=480)
    _close(wav, O.trim_fade(wref), 1e-5, "istft + trim_fade")


@pytest.mark.parametrize("M,N,K,ks,nw", [(16, 3072, 1024, 1, 8), (16, 1024, 1024, 4, 4), (16, 1024, 4096, 8, 4), (16, 8194, 1024, 1, 4),
                                         (2, 1024, 1024, 4, 4), (6, 64, 256, 1, 4), (40, 1024, 1024, 2, 4), (64, 3072, 1024, 1, 8)])
def test_gemv_decode(dev, M, N, K, ks, nw):
    from chatterbox_amd import ops
    x, w, b = _r((M, K), 1), _r((N, K), 2, 1 / math.sqrt(K)), _r((N,), 3)
    if ks == 1:
        out = torch.empty(M, N, device=dev)
        ops.gemv(x.to(dev), w.to(dev), out, bias=b.to(dev), nw=nw)
        _close(out, F.linear(x, w, b), 3e-5 * math.sqrt(K / 256), "gemv")
    else:
        part = torch.empty(ks, M, N, device=dev)
        ops.gemv(x.to(dev), w.to(dev), part, ksplit=ks, nw=nw)
        _close(part.sum(0), F.linear(x, w), 3e-5 * math.sqrt(K / 256), "gemv split-K")
        # consumer: residual + reduce + RMSNorm
        from oracle import ref_torch as O
        res, g = _r((M, N), 4), 1 + 0.1 * _r((N,), 5)
        xr, h = res.clone().to(dev), torch.empty(M, N, device=dev)
        ops.add_rmsnorm(xr, part, g.to(dev), h)
        ref = res + F.linear(x, w)
        _close(xr, ref, 3e-5 * math.sqrt(K / 256), "add (residual stream)")
        _close(h, O.rms_norm(ref, g), 5e-5 * math.sqrt(K / 256), "rmsnorm of the sum")


def test_gemv_swiglu(dev):
    from chatterbox_amd import ops, weights
    for M in (16, 33):
        D, Fh = 1024, 512
        x, g, u = _r((M, D), 1), _r((Fh, D), 2, 0.03), _r((Fh, D), 3, 0.03)
        out = torch.empty(M, Fh, device=dev)
        ops.gemv(x.to(dev), weights.pack_swiglu(g, u).to(dev), out, swiglu=True, nw=8)
        _close(out, F.silu(F.linear(x, g)) * F.linear(x, u), 5e-5, "gemv swiglu")


def test_decode_attn_rope_fused(dev):
    """Fused RoPE + cache append + attention == rope_kv followed by decode_attn == oracle."""
    from chatterbox_amd import ops
    from oracle import ref_torch as O
    rows, H, maxp = 6, 16, 512
    kc0, vc0, qkv = _r((rows, H, maxp, 64), 1), _r((rows, H, maxp, 64), 2), _r((rows, 3 * H * 64), 3)
    pos = torch.tensor([0, 4, 63, 128, 300, 511], dtype=torch.int32)
    cos, sin = O.rope_cos_sin(torch.arange(maxp), O.llama3_inv_freq())
    kc, vc, out = kc0.clone().to(dev), vc0.clone().to(dev), torch.empty(rows, H * 64, device=dev)
    ops.decode_attn_rope(qkv.to(dev), pos.to(dev), cos.to(dev), sin.to(dev), kc, vc, out, 0.125)
    q, k, v = (qkv.view(rows, 3, H, 64)[:, i] for i in range(3))
    c, s = cos[pos.long()][:, None], sin[pos.long()][:, None]
    qr, kr = q * c + O._rot_half(q) * s, k * c + O._rot_half(k) * s
    for r in range(rows):
        n = int(pos[r])
        kk = torch.cat([kc0[r, :, :n], kr[r][:, None]], 1)
        vv = torch.cat([vc0[r, :, :n], v[r][:, None]], 1)
        ref = F.scaled_dot_product_attention(qr[r].view(H, 1, 64), kk, vv)
        _close(out[r].view(H, 64), ref[:, 0], 2e-5, f"fused decode attention row {r}")
        _close(kc[r, :, n], kr[r], 1e-6, "k appended")
        _close(vc[r, :, n], v[r], 0.0, "v appended")
        _close(kc[r, :, :n], kc0[r, :, :n], 0.0, "cache untouched")


@pytest.mark.parametrize("split_min", [1, 512])
@pytest.mark.parametrize("rows,H,rope", [(1, 12, False), (1, 16, True), (2, 16, True), (5, 12, False)])
def test_decode_attn_rope_split_context(dev, rows, H, rope, split_min):
    """rows * heads < 128 (Turbo / Nano at small batch): the context of a (row, head) is walked by up to 8 workgroups and merged by
    the last to arrive.  Every context length 1..40 and a few long ones, launched back to back (the arrival counters reset
    themselves), against torch SDPA; and against the one-workgroup form of the same kernel."""
    from chatterbox_amd import ops
    from oracle import ref_torch as O
    ops.ensure_decode_attn_workspace(dev)
    ops.lib.cbx_set_decode_attn_split_min(split_min)  # 1: every context is split; 512 (the default): only the long ones, per row
    maxp = 1024
    kc0, vc0 = _r((rows, H, maxp, 64), 1), _r((rows, H, maxp, 64), 2)
    cos, sin = O.rope_cos_sin(torch.arange(maxp), O.llama3_inv_freq())
    cd, sd_ = (cos.to(dev), sin.to(dev)) if rope else (None, None)
    for n in list(range(0, 40)) + [127, 128, 129, 511, 700, 1023]:
        pos = torch.tensor([(n + 37 * r) % maxp for r in range(rows)], dtype=torch.int32)
        qkv = _r((rows, 3 * H * 64), 100 + n)
        kc, vc, out = kc0.clone().to(dev), vc0.clone().to(dev), torch.empty(rows, H * 64, device=dev)
        ops.decode_attn_rope(qkv.to(dev), pos.to(dev), cd, sd_, kc, vc, out, 0.125)
        q, k, v = (qkv.view(rows, 3, H, 64)[:, i] for i in range(3))
        if rope:
            c, s = cos[pos.long()][:, None], sin[pos.long()][:, None]
            q, k = q * c + O._rot_half(q) * s, k * c + O._rot_half(k) * s
        for r in range(rows):
            m = int(pos[r])
            kk = torch.cat([kc0[r, :, :m], k[r][:, None]], 1)
            vv = torch.cat([vc0[r, :, :m], v[r][:, None]], 1)
            ref = F.scaled_dot_product_attention(q[r].reshape(H, 1, 64), kk, vv)
            _close(out[r].view(H, 64), ref[:, 0], 2e-5, f"split decode attention ctx {m + 1} row {r}")
            _close(kc[r, :, m], k[r], 1e-6, "k appended")
            _close(vc[r, :, m], v[r], 0.0, "v appended")
    ops.lib.cbx_set_decode_attn_split_min(512)
    assert int(ops._DA_WS[torch.device(dev).index or 0][1].abs().sum()) == 0, "arrival counters are back at zero"


def _unpack_operand(img, rows, K):
    """Inverse of the packed GEMV operand layout (include/cbx.h): image (ceil(rows/16)*16, K) -> row-major (rows, K)."""
    T = img.shape[0] // 16
    v = img.view(T, K // 32, 2, 4, 16, 4)            # [tile][kb][h][q][c][s]
    return v.permute(0, 4, 1, 3, 2, 5).reshape(T * 16, K)[:rows]  # [tile][c][kb][q][h][s]


@pytest.mark.parametrize("M,N,K,swiglu,nw", [(16, 3072, 1024, False, 8), (16, 4096, 1024, True, 8), (16, 8194, 1024, False, 4),
                                             (6, 96, 768, False, 4), (32, 1024, 1024, True, 4), (50, 64, 256, False, 4)])
def test_gemv_packed_rms_fused(dev, M, N, K, swiglu, nw):
    """Packed-operand decode GEMV with LlamaRMSNorm folded in: out = RMSNorm(x) W^T (+ SwiGLU) against torch fp32; the packed
    weight path must be bit-identical to the row-major one (same arithmetic order)."""
    from chatterbox_amd import ops, weights
    from oracle import ref_torch as O
    x, nwt = _r((M, K), 1), 1 + 0.1 * _r((K,), 2)
    hn = O.rms_norm(x, nwt)
    xp = ops.pack_gemv_weight(x.to(dev))
    assert torch.equal(_unpack_operand(xp, M, K).cpu(), x), "operand packer"
    out = torch.empty(M, N, device=dev)
    if swiglu:
        g, u = _r((N, K), 4, 1 / math.sqrt(K)), _r((N, K), 5, 1 / math.sqrt(K))
        wp = ops.pack_gemv_weight(torch.cat([g, u]).to(dev), swiglu=True)
        ops.gemv(xp, wp, out, N=N, M=M, K=K, swiglu=True, nw=nw, w_packed=True, x_packed=True, norm_w=nwt.to(dev))
        ref = F.silu(F.linear(hn, g)) * F.linear(hn, u)
        if nw >= 8:  # CBX_GEMV_SHALLOW (ABI v13): 2-deep load batches (<= 128 VGPRs: the co-resident form of the gate | up launch) -- same products, same order
            sh = torch.empty(M, N, device=dev)
            ops.gemv(xp, wp, sh, N=N, M=M, K=K, swiglu=True, nw=nw, w_packed=True, x_packed=True, norm_w=nwt.to(dev), flags=ops.GEMV_SHALLOW)
            assert torch.equal(sh, out), "the shallow form must equal the default one bit for bit"
        # no-norm variants: packed W (+ packed x) == row-major image, bit for bit
        o1, o2, o3 = (torch.empty(M, N, device=dev) for _ in range(3))
        ops.gemv(x.to(dev), weights.pack_swiglu(g, u).to(dev), o1, swiglu=True, nw=nw)
        ops.gemv(x.to(dev), wp, o2, N=N, swiglu=True, nw=nw, w_packed=True)
        ops.gemv(xp, wp, o3, N=N, M=M, K=K, swiglu=True, nw=nw, w_packed=True, x_packed=True)
    else:
        w = _r((N, K), 4, 1 / math.sqrt(K))
        wp = ops.pack_gemv_weight(w.to(dev))
        ops.gemv(xp, wp, out, N=N, M=M, K=K, nw=nw, w_packed=True, x_packed=True, norm_w=nwt.to(dev))
        ref = F.linear(hn, w)
        o1, o2, o3 = (torch.empty(M, N, device=dev) for _ in range(3))
        ops.gemv(x.to(dev), w.to(dev), o1, nw=nw)
        ops.gemv(x.to(dev), wp, o2, N=N, nw=nw, w_packed=True)
        ops.gemv(xp, wp, o3, N=N, M=M, K=K, nw=nw, w_packed=True, x_packed=True)
    _close(out, ref, 3e-5 * max(1.0, math.sqrt(K / 256)), "gemv rms-fused")
    assert torch.equal(o1, o2) and torch.equal(o1, o3), "packed operand paths must equal the row-major path bit for bit"


@pytest.mark.parametrize("M,N,K,nw", [(16, 1024, 1024, 16), (16, 1024, 4096, 16), (7, 1024, 1024, 8), (16, 64, 256, 4)])
def test_gemv_packed_residual_epilogue(dev, M, N, K, nw):
    """o / down projection form: out_packed = res_packed + x W^T, in place on the packed residual stream."""
    from chatterbox_amd import ops
    x, w, r = _r((M, K), 1), _r((N, K), 2, 1 / math.sqrt(K)), _r((M, N), 3)
    xp, wp = ops.pack_gemv_weight(x.to(dev)), ops.pack_gemv_weight(w.to(dev))
    rp = ops.pack_gemv_weight(r.to(dev))
    ops.gemv(xp, wp, rp, N=N, M=M, K=K, nw=nw, w_packed=True, x_packed=True, res=rp, out_packed=True)
    got = _unpack_operand(rp, M, N)
    _close(got, r + F.linear(x, w), 3e-5 * max(1.0, math.sqrt(K / 256)), "gemv residual epilogue")
    if M < 16:
        assert float(rp.view(-1, 16, 4)[:, M:].abs().max()) == 0.0, "pad rows of the packed image must stay zero"


@pytest.mark.parametrize("M,N,npart,swiglu", [(16, 3072, 4, False), (16, 8194, 4, False), (9, 1024, 2, False), (16, 4096, 2, True)])
def test_gemv_partial_sum_operand(dev, M, N, npart, swiglu):
    """x operand = x + sum of the producer's split-K partial images (fixed order), RMSNorm folded in, the sum written to x_out; and
    the producer side: a split-K GEMV writing its partial images in the packed layout."""
    from chatterbox_amd import ops
    from oracle import ref_torch as O
    K = 1024
    x, parts, nwt = _r((M, K), 1), _r((npart, M, K), 2, 0.3), 1 + 0.1 * _r((K,), 3)
    h = x.clone()
    for j in range(npart):
        h = h + parts[j]
    hn = O.rms_norm(h, nwt)
    xp = ops.pack_gemv_weight(x.to(dev))
    pp = torch.stack([ops.pack_gemv_weight(parts[j].to(dev)) for j in range(npart)])
    x_out = torch.zeros_like(xp)
    out = torch.empty(M, N, device=dev)
    if swiglu:
        g, u = _r((N, K), 4, 1 / math.sqrt(K)), _r((N, K), 5, 1 / math.sqrt(K))
        wp = ops.pack_gemv_weight(torch.cat([g, u]).to(dev), swiglu=True)
        ref = F.silu(F.linear(hn, g)) * F.linear(hn, u)
    else:
        w = _r((N, K), 4, 1 / math.sqrt(K))
        wp = ops.pack_gemv_weight(w.to(dev))
        ref = F.linear(hn, w)
    ops.gemv(xp, wp, out, N=N, M=M, K=K, swiglu=swiglu, nw=8, w_packed=True, x_packed=True, norm_w=nwt.to(dev), xpart=pp, x_out=x_out)
    _close(out, ref, 4e-5, "gemv with partial-sum operand")
    assert torch.equal(_unpack_operand(x_out, M, K).cpu(), h), "x_out must be x + sum(partials) in fixed order, bit for bit"
    # producer: split-K partial images in the packed layout
    w2 = _r((64, K), 6, 1 / math.sqrt(K))
    pimg = torch.zeros(4, 16, 64, device=dev)
    ops.gemv(xp, ops.pack_gemv_weight(w2.to(dev)), pimg, N=64, M=M, K=K, ksplit=4, nw=4, w_packed=True, x_packed=True, out_packed=True)
    got = sum(_unpack_operand(pimg[j], M, 64) for j in range(4))
    _close(got, F.linear(x, w2), 3e-5, "packed split-K partial images")


def test_decode_attn_packed_output_and_embed_packed(dev):
    from chatterbox_amd import ops
    from oracle import ref_torch as O
    rows, H, maxp = 6, 16, 128
    kc, vc, qkv = _r((rows, H, maxp, 64), 1).to(dev), _r((rows, H, maxp, 64), 2).to(dev), _r((rows, 3 * H * 64), 3).to(dev)
    pos = torch.tensor([0, 4, 63, 100, 17, 127], dtype=torch.int32, device=dev)
    cos, sin = O.rope_cos_sin(torch.arange(maxp), O.llama3_inv_freq())
    a, b = torch.empty(rows, H * 64, device=dev), torch.zeros(16, H * 64, device=dev)
    ops.decode_attn_rope(qkv, pos, cos.to(dev), sin.to(dev), kc.clone(), vc.clone(), a, 0.125)
    ops.decode_attn_rope(qkv, pos, cos.to(dev), sin.to(dev), kc.clone(), vc.clone(), b, 0.125, out_packed=True)
    assert torch.equal(_unpack_operand(b, rows, H * 64), a)
    ids = torch.tensor([5, 0, 99, 3, 3, 7, 1], dtype=torch.int64, device=dev)
    ids2 = torch.arange(7, dtype=torch.int32, device=dev)
    t1, t2 = _r((100, 256), 5).to(dev), _r((16, 256), 6).to(dev)
    e1, e2 = torch.empty(7, 256, device=dev), torch.zeros(16, 256, device=dev)
    ops.embed(ids, t1, e1, table2=t2, ids2=ids2)
    ops.embed(ids, t1, e2, table2=t2, ids2=ids2, out_packed=True)
    assert torch.equal(_unpack_operand(e2, 7, 256), e1)


@pytest.mark.parametrize("M,N,K,npart,act", [(16, 3072, 1024, 0, False), (5, 2304, 768, 4, False), (16, 4096, 1024, 0, True), (1, 6563, 1024, 4, False)])
def test_gemv_layernorm_fused(dev, M, N, K, npart, act):
    """GPT-2 form of the fused decode GEMV: out = act(LayerNorm(x + sum partials) W^T + bias) with the LayerNorm folded in as
    rstd (sum_k x w W - mean cw) + cb (cbx_gemv_t.ln_cw / ln_cb), against torch fp32."""
    from chatterbox_amd import ops
    x, parts = _r((M, K), 1) + 0.3, _r((max(npart, 1), M, K), 2, 0.3)
    lw, lb = 1 + 0.1 * _r((K,), 3), 0.1 * _r((K,), 4)
    w, b = _r((N, K), 5, 1 / math.sqrt(K)), _r((N,), 6)
    h = x.clone()
    for j in range(npart):
        h = h + parts[j]
    ref = F.linear(F.layer_norm(h, (K,), lw, lb, 1e-5), w, b)
    if act:
        ref = F.gelu(ref, approximate="tanh")
    wd = w.to(dev)
    cw, cb = torch.empty(1, N, device=dev), torch.empty(1, N, device=dev)
    ops.gemv(lw.view(1, -1).to(dev), wd, cw, nw=4)
    ops.gemv(lb.view(1, -1).to(dev), wd, cb, bias=b.to(dev), nw=4)
    out = torch.empty(M, N, device=dev)
    kw = {}
    if npart:
        kw = dict(xpart=torch.stack([ops.pack_gemv_weight(parts[j].to(dev)) for j in range(npart)]), x_out=torch.zeros(16, K, device=dev))
    ops.gemv(ops.pack_gemv_weight(x.to(dev)), ops.pack_gemv_weight(wd), out, N=N, M=M, K=K, nw=8, w_packed=True, x_packed=True,
             norm_w=lw.to(dev), ln_cw=cw.view(-1), ln_cb=cb.view(-1), act=ops.GELU_TANH if act else ops.NONE, **kw)
    _close(out, ref, 5e-5, "gemv layernorm-fused")


@pytest.mark.parametrize("M,N,K,ks,nw,res", [(16, 1024, 4096, 2, 8, False), (16, 1024, 1024, 1, 8, True), (9, 40, 256, 1, 4, False), (16, 1024, 4096, 2, 16, False)])
def test_gemv_half_tile(dev, M, N, K, ks, nw, res):
    """8-column output tiles (cbx_gemv_t.half_tile + the swiglu = 8 packed image): same results as the 16-column form, bit for bit."""
    from chatterbox_amd import ops
    x, w, r = _r((M, K), 1), _r((N, K), 2, 1 / math.sqrt(K)), _r((M, (N + 31) // 32 * 32), 3)
    xp = ops.pack_gemv_weight(x.to(dev))
    w16, w8 = ops.pack_gemv_weight(w.to(dev)), ops.pack_gemv_weight(w.to(dev), half_tile=True)
    shape = (ks, M, N) if ks > 1 else (M, N)
    a, b = torch.zeros(shape, device=dev), torch.zeros(shape, device=dev)
    kw = dict(N=N, M=M, K=K, ksplit=ks, nw=nw, w_packed=True, x_packed=True)
    if res and N % 32 == 0:
        ra, rb = ops.pack_gemv_weight(r.to(dev)), ops.pack_gemv_weight(r.to(dev))
        ops.gemv(xp, w16, ra, res=ra, out_packed=True, **kw)
        ops.gemv(xp, w8, rb, res=rb, out_packed=True, half_tile=True, **kw)
        assert torch.equal(ra, rb)
        _close(_unpack_operand(ra, M, N), r[:, :N] + F.linear(x, w), 3e-5 * max(1.0, math.sqrt(K / 256)), "half-tile gemv + residual")
    else:
        ops.gemv(xp, w16, a, **kw)
        ops.gemv(xp, w8, b, half_tile=True, **kw)
        assert torch.equal(a, b)
        got = b.sum(0) if ks > 1 else b
        _close(got, F.linear(x, w), 3e-5 * max(1.0, math.sqrt(K / 256)), "half-tile gemv")


@pytest.mark.parametrize("case", ["plain", "rms_swiglu", "half_ks2", "rms_np2"])
def test_gemv_bf16_weights_equal_rounded_fp32(dev, case):
    """Opt-in bf16 decode weights (cbx_pack_gemv_weight_bf16 / cbx_gemv_t.w_bf16): bit-identical to the fp32 kernel run on the
    bf16-ROUNDED weights (the widening is exact and the MFMA order is the same), i.e. the only deviation is the rounding itself."""
    from chatterbox_amd import ops
    M, K = 16, 1024
    x = _r((M, K), 1)
    xp = ops.pack_gemv_weight(x.to(dev))
    kw = dict(M=M, K=K, w_packed=True, x_packed=True, nw=8)
    if case == "rms_swiglu":
        N = 4096
        w = _r((2 * N, K), 2, 1 / math.sqrt(K))
        kw.update(N=N, swiglu=True, norm_w=(1 + 0.1 * _r((K,), 3)).to(dev))
        pk = dict(swiglu=True)
    elif case == "half_ks2":
        N, K = 1024, 4096
        x = _r((M, K), 1)
        xp = ops.pack_gemv_weight(x.to(dev))
        w = _r((N, K), 2, 1 / math.sqrt(K))
        kw.update(N=N, K=K, ksplit=2, half_tile=True)
        pk = dict(half_tile=True)
    else:
        N = 3072
        w = _r((N, K), 2, 1 / math.sqrt(K))
        kw.update(N=N)
        pk = {}
        if case == "rms_np2":
            parts = torch.stack([ops.pack_gemv_weight(_r((M, K), 5 + j, 0.3).to(dev)) for j in range(2)])
            kw.update(norm_w=(1 + 0.1 * _r((K,), 3)).to(dev), xpart=parts, x_out=torch.zeros(16, K, device=dev))
    wr = w.bfloat16().float()
    shape = (2, M, N) if kw.get("ksplit", 1) > 1 else (M, N)
    a, b = torch.zeros(shape, device=dev), torch.zeros(shape, device=dev)
    ops.gemv(xp, ops.pack_gemv_weight(w.to(dev), bf16=True, **pk), a, **kw)
    ops.gemv(xp, ops.pack_gemv_weight(wr.to(dev), **pk), b, **kw)
    assert torch.equal(a, b), f"{case}: bf16-weight kernel != fp32 kernel on rounded weights (max {float((a - b).abs().max()):.3e})"


@pytest.mark.parametrize("N,K,pro,R", [(40, 256, "plain", 0), (1030, 256, "ln", 3), (3072, 1024, "ln", 0), (6563, 1024, "ln", 8), (2304, 768, "ln", 0), (1024, 4096, "plain", 0), (768, 3072, "plain", 2),
                                       (1024, 1024, "attn", 1), (768, 768, "attn", 2)])
def test_gemv_row(dev, N, K, pro, R):
    """cbx_gemv_row_f32 (batch-1 decode GEMV, ABI v14): every prologue (plain / LayerNorm / attention-record merge), bias + gelu_new + in-place
    residual, ragged N (rows past N inside the last wave and the last workgroup) against fp64."""
    from chatterbox_amd import ops
    g = torch.Generator().manual_seed(N * 7 + K)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    bias, res = torch.randn(N, generator=g), torch.randn(N, generator=g)
    kw, x = {}, None
    if pro == "attn":
        H, S = K // 64, 5
        parts = torch.zeros(H, S, ops.ATTN_PART_REC)
        parts[:, :, 0] = torch.randn(H, S, generator=g) * 3
        parts[:, :, 1] = torch.rand(H, S, generator=g) + 0.5
        parts[:, :, 4:] = torch.randn(H, S, 64, generator=g)
        parts[1, 2, 0], parts[1, 2, 1], parts[1, 2, 4:] = float("-inf"), 0.0, 0.0  # a slice that saw no position
        m = parts[:, :, 0].double()
        f = torch.exp(m - m.max(1, keepdim=True).values)
        want_x = ((f[:, :, None] * parts[:, :, 4:].double()).sum(1) / (f * parts[:, :, 1].double()).sum(1, keepdim=True)).reshape(-1)
        kw = dict(parts=parts.to(dev))
    else:
        xc = torch.randn(K, generator=g) * 2 + 0.3
        x, want_x = xc.to(dev), xc.double()
        if pro == "ln":
            lw, lb = torch.randn(K, generator=g), torch.randn(K, generator=g)
            want_x = F.layer_norm(xc.double(), (K,), lw.double(), lb.double(), 1e-5)
            kw = dict(ln=(lw.to(dev), lb.to(dev)))
    out = res.clone().to(dev)  # in place: res aliases out
    ops.gemv_row(x, w.to(dev), out, bias=bias.to(dev), res=out, act=ops.GELU_TANH, rows_per_wave=R, **kw)
    want = F.gelu(w.double() @ want_x + bias.double(), approximate="tanh") + res.double()
    err = (out.cpu().double() - want).abs().max()
    assert err < 3e-5 * max(1.0, math.sqrt(K / 256)), float(err)


@pytest.mark.parametrize("S,chunks,rope", [(8, 4, False), (3, 2, True), (16, 8, False)])
def test_decode_attn_parts(dev, S, chunks, rope):
    """cbx_decode_attn_parts + the merge prologue of cbx_gemv_row_f32 == softmax attention of the new token over the cache (fp64), for ragged
    contexts incl. a first token (context 1), slices that stay empty, and contexts that need a second batch of chunks; the new k / v land in the cache."""
    from chatterbox_amd import ops
    rows, H, max_ctx = 3, 2, 320
    g = torch.Generator().manual_seed(S)
    k0, v0 = torch.randn(rows, H, max_ctx, 64, generator=g), torch.randn(rows, H, max_ctx, 64, generator=g)
    qkv = torch.randn(rows, 3 * H * 64, generator=g)
    pos = torch.tensor([0, 37, min(16 * S * chunks + 5, max_ctx - 1)], dtype=torch.int32)
    cos = sin = None
    if rope:
        ang = torch.rand(max_ctx, 32, generator=g) * 6.28
        cos, sin = torch.cat([ang.cos(), ang.cos()], 1).contiguous(), torch.cat([ang.sin(), ang.sin()], 1).contiguous()
    parts = torch.full((rows, H, S, ops.ATTN_PART_REC), float("nan")).to(dev)
    kc, vc = k0.clone().to(dev), v0.clone().to(dev)
    ops.decode_attn_parts(qkv.to(dev), pos.to(dev), kc, vc, parts, 0.125, cos_t=None if cos is None else cos.to(dev), sin_t=None if sin is None else sin.to(dev),
                          chunks=chunks)
    pc = parts.cpu()
    kc, vc = kc.cpu(), vc.cpu()
    for r in range(rows):
        p = int(pos[r])
        m = pc[r, :, :, 0].double()
        f = torch.where(torch.isinf(m), torch.zeros_like(m), torch.exp(m - m.max(1, keepdim=True).values))
        got = ((f[:, :, None] * pc[r, :, :, 4:].double()).sum(1) / (f * pc[r, :, :, 1].double()).sum(1, keepdim=True)).reshape(-1)
        q, k, v = (qkv[r].view(3, H, 64)[i].double() for i in range(3))
        if rope:
            rot = lambda t: torch.cat([-t[:, 32:], t[:, :32]], 1)
            q, k = q * cos[p].double() + rot(q) * sin[p].double(), k * cos[p].double() + rot(k) * sin[p].double()
        K_ = torch.cat([k0[r, :, :p].double(), k[:, None]], 1)
        V_ = torch.cat([v0[r, :, :p].double(), v[:, None]], 1)
        a = torch.softmax((K_ @ q[:, :, None]).squeeze(-1) * 0.125, -1)
        want = (a[:, :, None] * V_).sum(1).reshape(-1)
        assert (got - want).abs().max() < 1e-5, (r, float((got - want).abs().max()))
        assert torch.allclose(kc[r, :, p].double(), k, atol=1e-6) and torch.equal(vc[r, :, p], qkv[r].view(3, H, 64)[2])
        assert torch.equal(kc[r, :, :p], k0[r, :, :p]) and torch.equal(kc[r, :, p + 1:], k0[r, :, p + 1:])
        assert torch.equal(vc[r, :, :p], v0[r, :, :p]) and torch.equal(vc[r, :, p + 1:], v0[r, :, p + 1:])


@pytest.mark.parametrize("M,N,K,pro,R", [(2, 3072, 1024, "ln", 0), (4, 1030, 256, "ln", 3), (3, 1024, 4096, "plain", 0), (4, 768, 3072, "plain", 2), (2, 1024, 1024, "attn", 1),
                                         (4, 768, 768, "attn", 2), (2, 6563, 1024, "ln", 8)])
def test_gemv_row_few_rows(dev, M, N, K, pro, R):
    """cbx_gemv_row_f32 with 2 .. 4 activation rows (strided x / out / res, rows past M and columns past N never touched; K >= 3072: two passes over the rows)
    and the attention-record merge per row, against fp64."""
    from chatterbox_amd import ops
    g = torch.Generator().manual_seed(M * 1000 + N + K)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    res = torch.randn(M, N + 8, generator=g)  # row stride > N: the strides are honoured
    xc = torch.randn(M, K + 4, generator=g) * 1.5 + 0.2
    kw, x = {}, xc[:, :K]
    want_x = x.double()
    if pro == "ln":
        lw, lb = torch.randn(K, generator=g), torch.randn(K, generator=g)
        want_x = F.layer_norm(x.double(), (K,), lw.double(), lb.double(), 1e-5)
        kw = dict(ln=(lw.to(dev), lb.to(dev)))
    elif pro == "attn":
        H, S = K // 64, 6
        parts = torch.zeros(M, H, S, ops.ATTN_PART_REC)
        parts[..., 0] = torch.randn(M, H, S, generator=g) * 3
        parts[..., 1] = torch.rand(M, H, S, generator=g) + 0.5
        parts[..., 4:] = torch.randn(M, H, S, 64, generator=g)
        parts[M - 1, 1, 2, 0], parts[M - 1, 1, 2, 1] = float("-inf"), 0.0
        m = parts[..., 0].double()
        f = torch.exp(m - m.max(-1, keepdim=True).values)
        want_x = ((f[..., None] * parts[..., 4:].double()).sum(2) / (f * parts[..., 1].double()).sum(2, keepdim=True)).reshape(M, -1)
        kw, x = dict(parts=parts.to(dev)), None
    bias = torch.randn(N, generator=g)
    want = F.gelu(want_x @ w.double().t() + bias.double(), approximate="tanh") + res[:, :N].double()
    buf = torch.full((M + 1, N + 8), 7.0)
    buf[:M] = res
    buf = buf.to(dev)
    out = buf[:M, :N]  # in place: res aliases out; the row after M and the columns past N must stay untouched
    xd = None if x is None else xc.to(dev)[:, :K]
    ops.gemv_row(xd, w.to(dev), out, bias=bias.to(dev), res=out, act=ops.GELU_TANH, rows_per_wave=R, **kw)
    err = (out.cpu().double() - want).abs().max()
    assert err < 4e-5 * max(1.0, math.sqrt(K / 256)), float(err)
    assert bool((buf[M] == 7.0).all()) and torch.equal(buf[:M, N:].cpu(), res[:, N:])
