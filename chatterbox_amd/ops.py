"""Thin Python wrappers that hand raw device pointers of torch tensors + the current HIP stream to the C ABI.

torch is used only as the allocator / stream provider; every function below launches hand-written gfx950 kernels
from libcbx_hip.so and raises if the library is unavailable (no eager fallback).
"""
import ctypes
import os

import torch

from ._lib import (ATTN_PART_REC, ATTN_PL_CORESIDENT, GEMV_DEEP, GEMV_PRE_EPI, GEMV_SHALLOW, PL_TILE_CORESIDENT, AttnPartsParams, DecodeAttnParams, GemmParams,
                   GemmPlParams, GemvParams, GemvRowParams, SamplerParams, check, lib)

NONE, SILU, GELU_ERF, GELU_TANH, MISH, LRELU, ELU, TANH, SNAKE, ABS = range(10)


class KernelTimer:
    """HIP-event timing of individual launches of selected kernel classes, on the stream they are launched on.
    Used by bench.py for the roofline line (average launch duration + algorithmic FLOPs/bytes per launch)."""

    def __init__(self, kinds):
        self.kinds = set(kinds)
        self.rec = []  # (kind, start_event, end_event, flops, bytes)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for kind, e0, e1, fl, by in self.rec:
            d = out.setdefault(kind, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += by
        return out


TIMER = None  # set to a KernelTimer to time launches (never during hipGraph capture)


def _timed(kind, flops, nbytes, fn):
    if TIMER is None or kind not in TIMER.kinds:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    TIMER.rec.append((kind, e0, e1, float(flops), float(nbytes)))
    return r


def on_device(fn):
    """Method decorator for the engines: run the body with `self.dev` as the current HIP device, so that every launch (which goes
    to torch's CURRENT stream of the CURRENT device, see _stream()) lands on the GPU the engine's tensors live on -- the drop-in
    API accepts any device string (e.g. 'cuda:1' in a process whose current device is 0)."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        dev = getattr(self, "dev", None)
        if dev is None:  # decorating __init__(self, state_dict, device="cuda", ...): the device is the 2nd argument
            dev = k.get("device", a[1] if len(a) > 1 else "cuda")
        dev = dev if isinstance(dev, torch.device) else torch.device(dev)
        if dev.type != "cuda" or dev.index is None or dev.index == torch.cuda.current_device():
            return fn(self, *a, **k)
        with torch.cuda.device(dev):
            return fn(self, *a, **k)
    return wrapper


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _f32(t, name):
    assert t.dtype == torch.float32 and t.is_cuda, f"{name}: expected a CUDA fp32 tensor, got {t.dtype} on {t.device}"
    return t


# cbx_gemm_t.precision of every gemm()/linear()/conv1d() issued inside `with gemm_precision(p)`: 0 default, 1 exact, 3 / 6 split-bf16, 16 split-fp16.
# PER HOST THREAD (round 5): synthesize_pipelined enqueues T3 (exact) on one thread while the flow + vocoder (16) are enqueued on another; a module
# global would let the flow's scope leak into T3's conditioning / head GEMMs.  `ops.GEMM_PRECISION` reads the calling thread's value (module __getattr__).
import threading as _threading

_TLS = _threading.local()


def _prec():
    return getattr(_TLS, "prec", 0)


def __getattr__(name):
    if name == "GEMM_PRECISION":
        return _prec()
    raise AttributeError(name)


GEMM_DIAG = 0  # cbx_gemm_t.reserved0: only read by a -DCBX_DIAG build of gemm_split.hip (scripts/diag_gemm.sh)


class gemm_precision:
    """`with ops.gemm_precision(6): ...` -- the engines scope their numerics policy this way (T3 stays exact: sampled
    tokens must match the reference bit for bit; the CFM / vocoder run the fp32-accurate split-bf16 kernels)."""

    def __init__(self, precision):
        self.precision = int(precision)

    def __enter__(self):
        self._prev, _TLS.prec = _prec(), self.precision

    def __exit__(self, *exc):
        _TLS.prec = self._prev


LN_FUSION = os.environ.get("CBX_LN_FUSION", "1") != "0"  # CFM transformer blocks: LayerNorm folded into the consuming Linear
_RANGE_FLAGS = {}  # device index -> int32 device word (one per GPU: an engine per GPU may live in one process, see on_device())


def _dev_index(device=None):
    d = torch.device("cuda" if device is None else device)
    return torch.cuda.current_device() if d.index is None else d.index


def enable_range_flag(device=None):
    """Allocate (once per GPU) the device words the f16x3 kernels (precision 16) raise when an operand exceeds the fp16 range.  The library
    keeps one pointer per device ordinal (cbx_set_range_flag registers it for the CURRENT device; a launch looks its own device's word up),
    so engines on different GPUs of one process never OR into a foreign-device pointer.  Two words per GPU: select_range_flag() switches the
    registered one between batches whose launches are in flight at once (the launch carries the pointer that was registered when it was ENQUEUED)."""
    idx = _dev_index(device)
    if idx not in _RANGE_FLAGS:
        with torch.cuda.device(idx):
            _RANGE_FLAGS[idx] = [torch.zeros(2, dtype=torch.int32, device=torch.device("cuda", idx)), 0]
            check(lib.cbx_set_range_flag(_p(_RANGE_FLAGS[idx][0])), "cbx_set_range_flag")
    words, which = _RANGE_FLAGS[idx]
    return words[which: which + 1]


def select_range_flag(device, which):
    """Every precision-16 launch ENQUEUED from now on reports into word `which` (0 / 1) of this GPU: engine.synthesize_pipelined gives consecutive batches
    alternating words, so a trip is attributed to the batch that raised it although the vocoder of one batch runs beside the flow of the next."""
    enable_range_flag(device)
    idx = _dev_index(device)
    words = _RANGE_FLAGS[idx][0]
    _RANGE_FLAGS[idx][1] = int(which)
    if words.is_cuda:
        with torch.cuda.device(idx):  # (the registration is for the CURRENT device)
            check(lib.cbx_set_range_flag(words.data_ptr() + 4 * int(which)), "cbx_set_range_flag")
    else:  # the SIMT emulator's host words (tests/simt/harness.py)
        check(lib.cbx_set_range_flag(words.data_ptr() + 4 * int(which)), "cbx_set_range_flag")


def range_flag_tripped(device=None, which=None):
    """True when a precision-16 launch on this GPU since the last call saw an operand outside the fp16 range (synchronises; clears the flag).
    `which`: the word to look at (default: the one currently registered)."""
    ent = _RANGE_FLAGS.get(_dev_index(device))
    if ent is None:
        return False
    words, cur = ent
    w = cur if which is None else int(which)
    flag = words[w: w + 1]
    hit = bool(flag.item())
    if hit:
        flag.zero_()
    return hit


def gemm(A, W, C, *, M, N, K, lda, ldw, ldc, bias=None, R=None, ldr=0, C2=None, ldc2=0, act1=NONE, act2=NONE,
         act1_param=None, act2_param=None, act1_slope=0.0, act2_slope=0.0, alpha=1.0, beta=0.0, lens=None, Cin=0, taps=1,
         dil=1, stride=1, pad_left=0, up=1, Tin=0, nz1=1, nz2=1, a_s=(0, 0), w_s=(0, 0), c_s=(0, 0), r_s=(0, 0),
         c2_s=(0, 0), w_kn=False, swiglu=False, ln=None):
    """Raw access to cbx_gemm_f32 (see include/cbx.h).  A/W/C... are tensors (their data_ptr() is the base)."""
    p = GemmParams()
    p.A, p.W, p.C = _p(_f32(A, "A")), _p(_f32(W, "W")), _p(_f32(C, "C"))
    p.bias, p.R, p.C2 = _p(bias), _p(R), _p(C2)
    p.act1_param, p.act2_param, p.lens = _p(act1_param), _p(act2_param), _p(lens)
    if lens is not None:
        assert lens.dtype == torch.int32
    p.M, p.N, p.K = M, N, K
    p.Cin, p.taps, p.dil, p.stride, p.pad_left, p.up, p.Tin = Cin or (K // taps), taps, dil, stride, pad_left, up, Tin
    p.nz1, p.nz2, p.w_kn, p.swiglu = nz1, nz2, int(w_kn), int(swiglu)
    p.act1, p.act2, p.act1_slope, p.act2_slope, p.alpha, p.beta = act1, act2, act1_slope, act2_slope, alpha, beta
    p.lda, p.a_s1, p.a_s2 = lda, a_s[0], a_s[1]
    p.ldw, p.w_s1, p.w_s2 = ldw, w_s[0], w_s[1]
    p.ldc, p.c_s1, p.c_s2 = ldc, c_s[0], c_s[1]
    p.ldr, p.r_s1, p.r_s2 = ldr, r_s[0], r_s[1]
    p.ldc2, p.c2_s1, p.c2_s2 = ldc2, c2_s[0], c2_s[1]
    p.precision = _prec()
    p.reserved0 = GEMM_DIAG
    if ln is not None:  # (stats (M, 2), ln_w (K,), ln_b (K,)): LayerNorm folded into the A operand (ln_fusable())
        p.ln_stats, p.ln_w, p.ln_b = _p(_f32(ln[0], "ln stats")), _p(_f32(ln[1], "ln_w")), _p(_f32(ln[2], "ln_b"))
    nz = nz1 * nz2
    split = _prec() in (3, 6, 16) and M > 32 and not w_kn and not swiglu and (taps == 1 or (Cin or K // taps) % 32 == 0)
    kind = "gemm_f32_skinny" if M <= 32 else ("gemm_split" if split else "gemm_f32")  # mirrors the dispatch in gemm_f32.hip
    _timed(kind, 2.0 * M * N * K * nz, 4.0 * nz * (M * K / max(1, taps) + N * K + M * N),
           lambda: check(lib.cbx_gemm_f32(ctypes.byref(p), _stream()), "cbx_gemm_f32"))
    return C


def ln_fusable(M, K, lda=None):
    """Can `linear(..., ln=...)` fold a LayerNorm over K into its A operand here?  (the f16x3 split kernel on its Linear loader, C = 256
    statistics kernel: the CFM transformer blocks).  The library is asked (cbx_gemm_ln_fusable mirrors the dispatcher's own conditions:
    no forced tile / generic-loader knob, 31-bit byte offsets), so a tuning knob or a > 2 GiB activation falls back to layernorm + linear
    instead of raising."""
    if not (_prec() == 16 and K == 256 and M > 32 and LN_FUSION):
        return False
    return bool(lib.cbx_gemm_ln_fusable(int(M), int(K), int(K if lda is None else lda)))


def row_stats(x, stats, eps=1e-5):
    """stats[r] = (mean, rstd) of row r of x (rows, 256), as layernorm() computes them."""
    rows, C = x.shape
    assert x.stride(1) == 1 and stats.is_contiguous() and stats.shape == (rows, 2)
    check(lib.cbx_row_stats_f32(_p(_f32(x, "x")), _p(_f32(stats, "stats")), rows, C, x.stride(0), eps, _stream()), "cbx_row_stats_f32")
    return stats


def linear(x, w, out, bias=None, act=NONE, residual=None, out2=None, act2=NONE, act_param=None, act2_param=None,
           act_slope=0.0, act2_slope=0.0, alpha=1.0, beta=0.0, swiglu=False, ln=None):
    """out[M,N] = epilogue(x[M,K] @ w[N,K]^T).  x/out/residual are 2-D views with unit inner stride.
    ln = (stats, ln_w, ln_b): x is LayerNorm'ed on the way in (row_stats() + ln_fusable())."""
    M, K = x.shape
    N = w.shape[0]
    assert x.stride(1) == 1 and out.stride(1) == 1 and w.stride(1) == 1 and w.shape[1] == K
    return gemm(x, w, out, M=M, N=N, K=K, lda=x.stride(0), ldw=w.stride(0), ldc=out.stride(0), bias=bias, R=residual,
                ldr=0 if residual is None else residual.stride(0), C2=out2, ldc2=0 if out2 is None else out2.stride(0),
                act1=act, act2=act2, act1_param=act_param, act2_param=act2_param, act1_slope=act_slope,
                act2_slope=act2_slope, alpha=alpha, beta=beta, swiglu=swiglu, ln=ln)


def conv1d(x, w, out, *, taps, cin, bias=None, dil=1, stride=1, pad_left=0, up=1, lens=None, act=NONE, residual=None,
           out2=None, act2=NONE, act_param=None, act2_param=None, act_slope=0.0, act2_slope=0.0, alpha=1.0, beta=0.0,
           t_in=None):
    """Channel-last conv1d as implicit GEMM.  x (B,Tin,>=cin), w packed (N, taps*cin), out (B,Tout,N)."""
    B, Tin = x.shape[0], (t_in if t_in is not None else x.shape[1])
    Tout, N = out.shape[1], w.shape[0]
    assert x.stride(2) == 1 and out.stride(2) == 1 and w.shape[1] == taps * cin
    return gemm(x, w, out, M=Tout, N=N, K=taps * cin, Cin=cin, taps=taps, dil=dil, stride=stride, pad_left=pad_left, up=up,
                Tin=Tin, lens=lens, lda=x.stride(1), ldw=w.stride(0), ldc=out.stride(1), nz1=B, a_s=(x.stride(0), 0),
                c_s=(out.stride(0), 0), bias=bias, R=residual, ldr=0 if residual is None else residual.stride(1),
                r_s=(0 if residual is None else residual.stride(0), 0), C2=out2,
                ldc2=0 if out2 is None else out2.stride(1), c2_s=(0 if out2 is None else out2.stride(0), 0), act1=act,
                act2=act2, act1_param=act_param, act2_param=act2_param, act1_slope=act_slope, act2_slope=act2_slope,
                alpha=alpha, beta=beta)


# ----------------------------------------------------------------------------- plane-format operands (ABI v7, gemm_planes.hip)

class Planes:
    """An fp32 tensor (rows, C) stored as two fp16 planes x = h + l / 2048, row by row: `t` (rows, 2 * C) fp16 = [h | l].  `cols(c0, n)`
    is the operand made of n columns from c0 on (a pointer offset: row stride and plane offset stay those of the whole tensor)."""

    def __init__(self, rows, C, device, zero=False, t=None, c0=0, width=None):
        self.t = t if t is not None else (torch.zeros if zero else torch.empty)(rows, 2 * C, dtype=torch.float16, device=device)
        assert self.t.dtype == torch.float16 and self.t.is_cuda and self.t.stride(1) == 1
        self.rows, self.Call, self.c0 = self.t.shape[0], self.t.shape[1] // 2, c0
        self.C = self.Call - c0 if width is None else width

    def cols(self, c0, n):
        assert 0 <= c0 and c0 + n <= self.C
        return Planes(0, 0, None, t=self.t, c0=self.c0 + c0, width=n)

    def rows_view(self, r0, n):
        return Planes(0, 0, None, t=self.t[r0:r0 + n], c0=self.c0, width=self.C)

    @property
    def ptr(self):
        return self.t.data_ptr() + 2 * self.c0

    @property
    def ld(self):
        return self.t.stride(0)

    @property
    def lo(self):
        return self.Call

    def float(self):
        """Back to fp32 (tests): h + l / 2048."""
        h = self.t[:, self.c0:self.c0 + self.C].float()
        return h + self.t[:, self.Call + self.c0:self.Call + self.c0 + self.C].float() / 2048.0


def split_planes(x, out=None):
    """x (rows, C) fp32 (unit inner stride, C % 4 == 0) -> Planes; `out` may be a column range of a wider Planes."""
    rows, C = x.shape
    assert x.stride(1) == 1
    out = out if out is not None else Planes(rows, C, x.device)
    assert out.C == C
    check(lib.cbx_split_planes_f32(_p(_f32(x, "x")), out.ptr, rows, C, x.stride(0), out.ld, out.lo, _stream()), "cbx_split_planes_f32")
    return out


def layernorm_planes(x, w, b, out, eps=1e-5, act=NONE, post_add=None, scale=1.0):
    """LayerNorm (C = 256) (+ activation + per-channel post_add) of fp32 rows, written in plane format."""
    rows, C = x.shape
    assert x.stride(1) == 1 and out.C == C
    check(lib.cbx_layernorm_planes_f32(_p(_f32(x, "x")), out.ptr, _p(w), _p(b), _p(post_add), rows, C, x.stride(0), out.ld, out.lo, eps, act,
                                       scale, _stream()), "cbx_layernorm_planes_f32")
    return out


# launch geometry of the plane-format kernels issued inside `with planes_geometry(tile, attn_version)` (cbx_gemm_pl_t.tile / the version argument of
# cbx_flash_attn_planes_v: per-call descriptor fields since ABI v13, nothing process-wide); (0, 0) = the library's measured defaults
def _planes_geom():
    """(tile form, attention version) of the calling HOST THREAD's planes_geometry scope -- thread-local like the GEMM precision (a plane
    GEMM issued by another thread while the flow thread sits inside a scope must not pick the scope's forms up)."""
    return getattr(_TLS, "planes_geom", (0, 0))



class planes_geometry:
    def __init__(self, tile=0, attn_version=0):
        self.new = [int(tile), int(attn_version)]

    def __enter__(self):
        self.old = _planes_geom()
        _TLS.planes_geom = tuple(self.new)

    def __exit__(self, *a):
        _TLS.planes_geom = self.old


def gemm_planes(A, W, *, M, N, K, C=None, P=None, bias=None, R=None, act=NONE, act_slope=0.0, alpha=1.0, lens=None, Cin=0, taps=1, dil=1,
                stride=1, pad_left=0, Tin=0, nz1=1, a_s1=0, w_s1=0, ldc=0, c_s1=0, ldr=0, r_s1=0, p_s1=0, PT=None, pt_n0=0, pt_T=0, pt_zs=0, tile=None,
                ln=None, lnp=None, lnp_s1=0, ln_eps=1e-5):
    """Raw access to cbx_gemm_planes (include/cbx.h): A, W, P are Planes operands, C / R fp32 tensors (their data_ptr() is the base).
    PT (Planes over (groups * (N - pt_n0), >= pt_T) rows): output columns n >= pt_n0 are written TRANSPOSED per group of pt_T rows.
    ln = (w, b) + lnp (Planes, N == 256): the epilogue also writes LayerNorm(C row) * w + b to lnp (ABI v13: replaces a layernorm_planes launch)."""
    p = GemmPlParams()
    p.A, p.W, p.C, p.P = A.ptr, W.ptr, _p(C), None if P is None else P.ptr
    p.bias, p.R, p.lens = _p(bias), _p(R), _p(lens)
    if lens is not None:
        assert lens.dtype == torch.int32
    p.M, p.N, p.K = M, N, K
    p.Cin, p.taps, p.dil, p.stride, p.pad_left, p.Tin, p.nz1 = Cin or (K // taps), taps, dil, stride, pad_left, Tin, nz1
    p.act, p.act_slope, p.alpha = act, act_slope, alpha
    p.lda, p.a_lo, p.a_s1 = A.ld, A.lo, a_s1
    p.ldw, p.w_lo, p.w_s1 = W.ld, W.lo, w_s1
    p.ldc, p.c_s1, p.ldr, p.r_s1 = ldc, c_s1, ldr, r_s1
    if P is not None:
        p.ldp, p.p_lo, p.p_s1 = P.ld, P.lo, p_s1
    p.reserved0 = GEMM_DIAG
    p.tile = _planes_geom()[0] if tile is None else int(tile)
    if ln is not None:
        assert lnp is not None and N == 256
        p.ln_w, p.ln_b, p.LNP, p.ld_lnp, p.lnp_lo, p.lnp_s1, p.ln_eps = _p(_f32(ln[0], "ln_w")), _p(ln[1]), lnp.ptr, lnp.ld, lnp.lo, lnp_s1, ln_eps
    if PT is not None:
        p.PT, p.pt_n0, p.pt_T, p.pt_ld, p.pt_lo, p.pt_zs = PT.ptr, pt_n0, pt_T, PT.ld, PT.lo, pt_zs
    _timed("gemm_planes", 2.0 * M * N * K * nz1, 4.0 * nz1 * (M * K / max(1, taps) + N * K + M * N),
           lambda: check(lib.cbx_gemm_planes(ctypes.byref(p), _stream()), "cbx_gemm_planes"))


def linear_planes(x, w, *, out=None, outp=None, bias=None, act=NONE, residual=None, act_slope=0.0, ln=None, lnp=None):
    """epilogue(x @ w^T) for Planes x (M, K), w (N, K): fp32 `out` (M, N) and / or Planes `outp`; residual fp32 (may alias out).
    ln = (w, b), lnp (Planes (M, 256)): LayerNorm of the finished row to lnp as well (N == 256)."""
    M, K, N = x.rows, x.C, w.rows
    assert w.C == K and (out is not None or outp is not None)
    gemm_planes(x, w, M=M, N=N, K=K, C=out, P=outp, bias=bias, R=residual, act=act, act_slope=act_slope,
                ldc=0 if out is None else out.stride(0), ldr=0 if residual is None else residual.stride(0), ln=ln, lnp=lnp)


def conv1d_planes(x, w, *, B, T, taps, cin, out=None, outp=None, bias=None, pad_left=0, lens=None, act=NONE, residual=None):
    """Channel-last causal Conv1d as implicit GEMM on plane operands: x Planes over (B * T, cin) rows, w Planes (N, taps * cin);
    fp32 out (B, T, N) and / or Planes outp over (B * T, N)."""
    N = w.rows
    assert w.C == taps * cin and x.C >= cin
    gemm_planes(x, w, M=T, N=N, K=taps * cin, Cin=cin, taps=taps, pad_left=pad_left, Tin=T, lens=lens, nz1=B, a_s1=T * x.ld,
                C=out, P=outp, bias=bias, R=residual, act=act, ldc=0 if out is None else out.stride(1), c_s1=0 if out is None else out.stride(0),
                ldr=0 if residual is None else residual.stride(1), r_s1=0 if residual is None else residual.stride(0),
                p_s1=0 if outp is None else T * outp.ld)


def flash_attn_planes(q, k, vt, out, *, Z, H, T, vt_sb, scale, key_lens=None, causal=False, version=None):
    """q, k: Planes column ranges (Z * T rows, H * 64 columns); vt: Planes over (Z * H * 64 rows, >= T rounded up to 8 columns) = V^T
    (batch stride vt_sb halves); out: Planes (Z * T rows, H * 64)."""
    args = (q.ptr, k.ptr, vt.ptr, out.ptr, _p(key_lens), Z, H, T, T, T * q.ld, q.ld, q.lo, T * k.ld, k.ld, k.lo, vt_sb, vt.ld, vt.lo,
            T * out.ld, out.ld, out.lo, scale, int(causal))
    _timed("flash_attn_planes", 4.0 * Z * H * T * T * 64 * (0.5 if causal else 1.0), 4.0 * Z * H * 64 * 4 * T,
           lambda: check(lib.cbx_flash_attn_planes_v(*args, _planes_geom()[1] if version is None else int(version), _stream()), "cbx_flash_attn_planes_v"))
    return out


def bmm(a, b, out, *, nn=False, alpha=1.0):
    """Two-level batched matmul on 4-D strided views (Z1, Z2, M, K):  out = a @ b^T (b: Z1,Z2,N,K) or, with nn=True,
    out = a @ b (b: Z1,Z2,K,N)."""
    Z1, Z2, M, K = a.shape
    N = b.shape[3] if nn else b.shape[2]
    assert a.stride(3) == 1 and b.stride(3) == 1 and out.stride(3) == 1
    return gemm(a, b, out, M=M, N=N, K=K, lda=a.stride(2), ldw=b.stride(2), ldc=out.stride(2), nz1=Z1, nz2=Z2,
                a_s=(a.stride(0), a.stride(1)), w_s=(b.stride(0), b.stride(1)), c_s=(out.stride(0), out.stride(1)),
                w_kn=nn, alpha=alpha)


def _tile_rows(half_tile):
    """gemv / pack_gemv_weight `half_tile`: False -> 16-column tiles, True -> 8, or the tile width itself (8, 12, 4)."""
    tr = 0 if not half_tile else (8 if half_tile is True or half_tile == 1 else int(half_tile))
    assert tr in (0, 4, 8, 12), f"narrow gemv tiles are 4, 8 or 12 columns wide, got {half_tile!r}"
    return tr


def pack_gemv_weight(w, swiglu=False, half_tile=False, bf16=False):
    """(N, K) fp32 weight [swiglu: (2F, K) = gate rows then up rows] -> lane-ordered packed image of cbx_gemv_f32 (w_packed = 1):
    (ceil(N/16)*16, K) floats [swiglu: (2*ceil(F/16)*16, K)].  half_tile (True = 8, or 12 / 4): the narrow-tile image of the same weight.
    Done once at load (weights are constants)."""
    w = _f32(w, "w").contiguous()
    R, K = w.shape
    N = R // 2 if swiglu else R
    tr = _tile_rows(half_tile)
    assert not (tr and swiglu)
    rows = (N + tr - 1) // tr * tr if tr else (N + 15) // 16 * 16 * (2 if swiglu else 1)
    if bf16:  # opt-in decode numerics: weights rounded to bf16 (cbx_gemv_t.w_bf16), half the streamed bytes
        out = torch.empty(rows, K, dtype=torch.bfloat16, device=w.device)
        check(lib.cbx_pack_gemv_weight_bf16(_p(w), _p(out), N, K, w.stride(0), tr or int(swiglu), _stream()), "cbx_pack_gemv_weight_bf16")
        return out
    out = torch.empty(rows, K, device=w.device)
    check(lib.cbx_pack_gemv_weight_f32(_p(w), _p(out), N, K, w.stride(0), tr or int(swiglu), _stream()), "cbx_pack_gemv_weight_f32")
    return out


def gemv(x, w, out, *, N=None, bias=None, ksplit=1, nw=4, swiglu=False, act=NONE, w_packed=False, x_packed=False, M=None, K=None,
         norm_w=None, eps=1e-5, res=None, out_packed=False, xpart=None, x_out=None, ln_cw=None, ln_cb=None, half_tile=False, flags=0,
         col_tiles=0, ssq_out=None):
    """Decode GEMM: x (M<=64, K), w (N, K) [swiglu: packed (2N, K)], out (M, N) or, for ksplit > 1, (ksplit, M, N) partials.
    w_packed: w is a pack_gemv_weight() image (pass N); x_packed: x is in the same lane-ordered layout (pass M, K);
    norm_w: RMSNorm(x) folded in (packed operands only); res: residual added in the epilogue (same layout as out, may alias it);
    out_packed: out is written in the packed operand layout of the next gemv (ksplit > 1: out (ksplit, rows16, N) partial images);
    half_tile (True = 8, or 12 / 4): output columns per workgroup, w being the pack_gemv_weight image of that tile width;
    xpart (2 or 4, rows16, K): split-K partial images summed into the x operand on the fly, x_out receives x + sum(xpart);
    flags: cbx_gemv_t.flags (GEMV_PRE_EPI | GEMV_DEEP: per-launch geometry bits, results unchanged);
    col_tiles (1 .. 4, RMSNorm-folded packed fp32 form): a workgroup owns that many 16-column tiles sharing every x register and 1 / ksplit of K;
    ksplit > 1 then leaves UN-normalised partial sums out (ksplit, M, N) + ssq_out (ksplit, 16) for the consumer (decode_attn_rope(qkv_parts=...))."""
    p, M, N, K = _gemv_params(x, w, out, N=N, bias=bias, ksplit=ksplit, nw=nw, swiglu=swiglu, act=act, w_packed=w_packed, x_packed=x_packed, M=M, K=K,
                              norm_w=norm_w, eps=eps, res=res, out_packed=out_packed, xpart=xpart, x_out=x_out, ln_cw=ln_cw, ln_cb=ln_cb,
                              half_tile=half_tile, flags=flags, col_tiles=col_tiles, ssq_out=ssq_out)
    _timed("gemv_f32", 2.0 * M * N * K * (2 if swiglu else 1), 4.0 * N * K * (2 if swiglu else 1),
           lambda: check(lib.cbx_gemv_f32(ctypes.byref(p), _stream()), "cbx_gemv_f32"))
    return out


def _gemv_params(x, w, out, *, N=None, bias=None, ksplit=1, nw=4, swiglu=False, act=NONE, w_packed=False, x_packed=False, M=None, K=None,
                 norm_w=None, eps=1e-5, res=None, out_packed=False, xpart=None, x_out=None, ln_cw=None, ln_cb=None, half_tile=False, flags=0,
                 col_tiles=0, ssq_out=None):
    """The cbx_gemv_t descriptor of a gemv() call: (descriptor, M, N, K)."""
    if x_packed:
        assert w_packed and M is not None and K is not None
    else:
        M, K = x.shape
    N = N or (w.shape[0] // 2 if swiglu else w.shape[0])
    p = GemvParams()
    w_bf16 = w.dtype == torch.bfloat16
    assert not w_bf16 or (w_packed and x_packed), "bf16 weights are a packed-operand feature"
    p.x, p.W, p.bias, p.out = _p(_f32(x, "x")), _p(w if w_bf16 else _f32(w, "w")), _p(bias), _p(_f32(out, "out"))
    p.M, p.N, p.K, p.ksplit, p.nw, p.swiglu, p.act = M, N, K, ksplit, nw, int(swiglu), act
    p.ldx, p.ldw = x.stride(0), w.stride(0)
    p.w_packed, p.x_packed, p.half_tile, p.w_bf16 = int(w_packed), int(x_packed), _tile_rows(half_tile), int(w_bf16)
    p.out_packed, p.norm_w, p.res, p.eps = int(out_packed), _p(norm_w), _p(res), eps
    p.ln_cw, p.ln_cb = _p(ln_cw), _p(ln_cb)  # LayerNorm form (GPT-2): see cbx_gemv_t
    p.flags = int(flags)
    p.col_tiles, p.ssq_out = int(col_tiles), _p(ssq_out)
    if xpart is not None:
        p.n_xpart, p.xpart, p.xpart_stride, p.x_out = xpart.shape[0], _p(_f32(xpart, "xpart")), xpart.stride(0), _p(x_out)
    if ksplit > 1:
        assert out.dim() == 3 and out.shape[0] == ksplit
        p.ldo, p.part_stride = out.stride(1), out.stride(0)
    else:
        p.ldo, p.part_stride = out.stride(0), 0
    return p, M, N, K


def add_rmsnorm(x, part, w, h, eps=1e-5, bias=None, rms=True):
    """x += sum_k part[k]; h = rmsnorm(x) * w  (rms=False: LayerNorm with bias).   part (ksplit, rows, C) or None."""
    rows, C = x.shape
    ks = 0 if part is None else part.shape[0]
    check(lib.cbx_add_norm_f32(_p(x), _p(part), ks, 0 if part is None else part.stride(0), 0 if part is None else part.stride(1),
                               _p(w), _p(bias), _p(h), rows, C, x.stride(0), h.stride(0), eps, int(rms), _stream()), "cbx_add_norm_f32")
    return h


def layernorm(x, w, b, out, eps=1e-5, rms=False, act=NONE, post_add=None, scale=1.0):
    rows, C = x.shape
    assert x.stride(1) == 1 and out.stride(1) == 1
    check(lib.cbx_layernorm_f32(_p(x), _p(out), _p(w), _p(b), _p(post_add), rows, C, x.stride(0), out.stride(0), eps,
                                int(rms), act, scale, _stream()), "cbx_layernorm_f32")
    return out


def flash_attn(q, k, v, out, scale, key_lens=None, causal=False):
    """q (Z,Tq,H,64), k/v (Z,Tk,H,64), out (Z,Tq,H,64): strided views with head stride 64, unit inner stride.  k / v may carry another head stride (ABI v15:
    views of the KV cache, exact fp32 only)."""
    Z, Tq, H, D = q.shape
    Tk = k.shape[1]
    assert D == 64
    for t in (q, out):
        assert t.stride(3) == 1 and t.stride(2) == 64
    assert k.stride(3) == 1 and v.stride(3) == 1
    if k.stride(2) != 64 or v.stride(2) != 64:
        assert _prec() not in (3, 6, 16), "K / V head strides: the exact fp32 attention only"
        kv_args = (_p(q), _p(k), _p(v), _p(out), _p(key_lens), Z, H, Tq, Tk, q.stride(0), q.stride(1), k.stride(0), k.stride(1), k.stride(2),
                   v.stride(0), v.stride(1), v.stride(2), out.stride(0), out.stride(1), scale, int(causal))
        _timed("flash_attn_f32", 4.0 * Z * H * Tq * Tk * 64 * (0.5 if causal else 1.0), 4.0 * Z * H * 64 * (2 * Tq + 2 * Tk),
               lambda: check(lib.cbx_flash_attn_kv_f32(*kv_args, _stream()), "cbx_flash_attn_kv_f32"))
        return out
    args = (_p(q), _p(k), _p(v), _p(out), _p(key_lens), Z, H, Tq, Tk, q.stride(0), q.stride(1), k.stride(0), k.stride(1),
            v.stride(0), v.stride(1), out.stride(0), out.stride(1), scale, int(causal))
    if _prec() in (3, 6, 16):  # same numerics policy as the GEMMs issued in this scope
        prec = _prec()
        fn = lambda: check(lib.cbx_flash_attn_split_f32(*args, prec, _stream()), "cbx_flash_attn_split_f32")
    else:
        fn = lambda: check(lib.cbx_flash_attn_f32(*args, _stream()), "cbx_flash_attn_f32")
    _timed("flash_attn_f32", 4.0 * Z * H * Tq * Tk * 64 * (0.5 if causal else 1.0), 4.0 * Z * H * 64 * (2 * Tq + 2 * Tk), fn)
    return out


def flash_relpos(q4, pp, out, scale, key_lens=None):
    """Conformer rel-pos attention without materialised scores.  q4 (Z,T,4,H,64): the fused [q + u | q + v | k | v] projection (any batch /
    token strides, unit inner strides); pp (2T-1, H*64) = linear_pos(pos_emb); out (Z,T,H,64) view."""
    Z, T, four, H, D = q4.shape
    assert four == 4 and D == 64 and q4.stride(4) == 1 and q4.stride(3) == 64 and q4.stride(2) == H * 64
    assert pp.shape[0] == 2 * T - 1 and pp.stride(1) == 1 and out.stride(3) == 1 and out.stride(2) == 64
    args = (_p(q4[:, :, 0]), _p(q4[:, :, 1]), _p(q4[:, :, 2]), _p(q4[:, :, 3]), _p(pp), _p(out), _p(key_lens), Z, H, T,
            q4.stride(0), q4.stride(1), pp.stride(0), out.stride(0), out.stride(1), scale)
    _timed("flash_relpos_f32", 6.0 * Z * H * T * T * 64, 4.0 * Z * H * 64 * 5 * T + 4.0 * H * 64 * (2 * T - 1),
           lambda: check(lib.cbx_flash_relpos_f32(*args, _stream()), "cbx_flash_relpos_f32"))
    return out


def decode_attn(q, kc, vc, out, ctx_lens, scale):
    """q/out (rows, H*64) views; kc/vc (rows, H, max_pos, 64) contiguous caches; ctx_lens int32 (rows,)."""
    rows, H = kc.shape[0], kc.shape[1]
    check(lib.cbx_decode_attn_f32(_p(q), _p(kc), _p(vc), _p(out), _p(ctx_lens), rows, H, q.stride(0), out.stride(0),
                                  kc.stride(0), kc.stride(1), scale, _stream()), "cbx_decode_attn_f32")
    return out


_DA_WS = {}  # device index -> (partials, arrival counters) of the split-context decode attention: the TEST-HOOK workspace of the positional entry


def ensure_decode_attn_workspace(device):
    """TEST HOOK: register, once per device, the process-wide workspace the POSITIONAL cbx_decode_attn_rope_f32 splits a context through when
    rows * heads < 128 (op-level tests; single-stream use).  The engines own a DecodeAttnGeom each instead."""
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    if idx in _DA_WS:
        return
    with torch.cuda.device(idx):
        ws = torch.empty(128 * 8 * 66, dtype=torch.float32, device=f"cuda:{idx}")
        cnt = torch.zeros(128, dtype=torch.int32, device=f"cuda:{idx}")
        torch.cuda.current_stream().synchronize()
        check(lib.cbx_set_decode_attn_workspace(_p(ws), _p(cnt), 128), "cbx_set_decode_attn_workspace")
    _DA_WS[idx] = (ws, cnt)


class DecodeAttnGeom:
    """Geometry + CALLER-OWNED split-context workspace of cbx_decode_attn_rope (cbx_decode_attn_t, ABI v10).  One per engine state (i.e. per
    stream of launches that may be in flight together): nothing of it is process-wide, so two engines in one process -- or a hipGraph captured
    before another engine changed its geometry -- keep their own.  Allocate outside stream captures (the engines do it at construction)."""

    def __init__(self, device, unroll=0, pipeline=0, split_min=None, max_pairs=128, split=True):
        dev = torch.device(device)
        if split_min is None:  # 0 = the library's 512
            split_min = 0
        self.unroll, self.pipeline, self.split_min, self.max_pairs = int(unroll), int(pipeline), int(split_min), int(max_pairs)
        self.ws = torch.empty(max_pairs * 8 * 66, dtype=torch.float32, device=dev) if split else None
        self.cnt = torch.zeros(max_pairs, dtype=torch.int32, device=dev) if split else None

    def fill(self, p):
        p.unroll, p.pipeline, p.split_min = self.unroll, self.pipeline, self.split_min
        p.split_ws, p.split_cnt, p.split_pairs = _p(self.ws), _p(self.cnt), (self.max_pairs if self.ws is not None else 0)


def decode_attn_rope(qkv, positions, cos_t, sin_t, kc, vc, out, scale, out_packed=False, geom=None, qkv_ssq=None, rms_dim=0, rms_eps=1e-5):
    """Fused RoPE + KV append + decode attention: qkv (rows, 3*H*64), caches (rows,H,max,64), out (rows, H*64)
    [out_packed: the packed operand image of the o-projection gemv, (ceil(rows/16)*16, H*64)].
    geom: a DecodeAttnGeom (the engines: geometry and workspace per call); None = the positional entry point on the process-wide test hooks.
    qkv (nparts, rows, 3*H*64) + qkv_ssq (nparts, 16) + rms_dim: split-K partial sums of the RMSNorm-folded projection (gemv(col_tiles=..., ksplit > 1))."""
    rows, H = kc.shape[0], kc.shape[1]
    nparts = 0
    if qkv.dim() == 3:
        assert geom is not None and qkv_ssq is not None and rms_dim > 0 and qkv.shape[0] <= 4 and qkv_ssq.shape == (qkv.shape[0], 16)
        nparts = qkv.shape[0]
    if geom is None:
        check(lib.cbx_decode_attn_rope_f32(_p(qkv), _p(positions), _p(cos_t), _p(sin_t), _p(kc), _p(vc), _p(out), rows, H,
                                           qkv.stride(0), out.stride(0), int(out_packed), kc.stride(0), kc.stride(1), scale, _stream()),
              "cbx_decode_attn_rope_f32")
        return out
    p = DecodeAttnParams()
    p.qkv, p.positions, p.cos_t, p.sin_t, p.kc, p.vc, p.o = _p(qkv), _p(positions), _p(cos_t), _p(sin_t), _p(kc), _p(vc), _p(out)
    p.rows, p.n_heads, p.ld_qkv, p.o_ld, p.o_packed = rows, H, qkv.stride(-2), out.stride(0), int(out_packed)
    p.cache_row_stride, p.cache_head_stride, p.scale = kc.stride(0), kc.stride(1), scale
    if nparts:
        p.qkv_nparts, p.qkv_part_stride, p.qkv_ssq, p.rms_dim, p.rms_eps = nparts, qkv.stride(0), _p(qkv_ssq), int(rms_dim), float(rms_eps)
    geom.fill(p)
    check(lib.cbx_decode_attn_rope(ctypes.byref(p), _stream()), "cbx_decode_attn_rope")
    return out


def gemv_row(x, w, out, *, bias=None, res=None, ln=None, eps=1e-5, parts=None, act=NONE, rows_per_wave=0):
    """Few-row decode GEMV (cbx_gemv_row_f32): out (M, N) [or (N,) for one row] = act(x' . w[n] + bias) + res, M <= 4, with x' = x (M, K) / (K,),
    LayerNorm(x) (ln = (weight, bias)) or the merge of the split-context attention records `parts` (M, n_heads, n_splits, ATTN_PART_REC) [or 3-D for one
    row] that decode_attn_parts left (x is then None).  w (N, K) row-major fp32 -- the checkpoint layout, no packed image."""
    N, K = w.shape
    o2 = out if out.dim() == 2 else out.view(1, -1)
    M = o2.shape[0]
    assert o2.shape[1] == N and M <= 4
    p = GemvRowParams()
    p.W, p.bias, p.out = _p(_f32(w, "w")), _p(bias), _p(_f32(out, "out"))
    p.M, p.ldo = M, o2.stride(0)
    if x is not None:
        x2 = x if x.dim() == 2 else x.view(1, -1)
        assert x2.shape == (M, K) and x2.stride(1) == 1
        p.x, p.ldx = _p(_f32(x, "x")), x2.stride(0)
    if res is not None:
        r2 = res if res.dim() == 2 else res.view(1, -1)
        assert r2.shape == (M, N)
        p.res, p.ldr = _p(res), r2.stride(0)
    if ln is not None:
        p.ln_w, p.ln_b, p.eps = _p(ln[0]), _p(ln[1]), eps
    if parts is not None:
        p4 = parts if parts.dim() == 4 else parts.unsqueeze(0)
        assert p4.shape[0] == M and p4.shape[3] == ATTN_PART_REC and p4[0].is_contiguous()
        p.parts, p.n_parts, p.n_heads, p.parts_row_stride = _p(parts), p4.shape[2], p4.shape[1], p4.stride(0)
    p.N, p.K, p.ldw, p.act, p.rows_per_wave = N, K, w.stride(0), act, int(rows_per_wave)
    _timed("gemv_f32", 2.0 * M * N * K, 4.0 * N * K, lambda: check(lib.cbx_gemv_row_f32(ctypes.byref(p), _stream()), "cbx_gemv_row_f32"))
    return out


def decode_attn_parts(qkv, positions, kc, vc, parts, scale, cos_t=None, sin_t=None, chunks=0):
    """Split-context decode attention that leaves its partial results (cbx_decode_attn_parts): qkv (rows, 3*H*64), caches (rows, H, max_ctx, 64),
    parts (rows, H, n_splits, ATTN_PART_REC) -- merged by the consuming gemv_row(parts=parts[row])."""
    rows, H, max_ctx = kc.shape[0], kc.shape[1], kc.shape[2]
    assert parts.shape[:2] == (rows, H) and parts.shape[3] == ATTN_PART_REC and parts.is_contiguous()
    p = AttnPartsParams()
    p.qkv, p.positions, p.cos_t, p.sin_t, p.kc, p.vc, p.parts = _p(qkv), _p(positions), _p(cos_t), _p(sin_t), _p(kc), _p(vc), _p(parts)
    p.rows, p.n_heads, p.n_splits, p.chunks, p.max_ctx = rows, H, parts.shape[2], int(chunks), max_ctx
    p.ld_qkv, p.cache_row_stride, p.cache_head_stride, p.scale = qkv.stride(0), kc.stride(0), kc.stride(1), scale
    check(lib.cbx_decode_attn_parts(ctypes.byref(p), _stream()), "cbx_decode_attn_parts")
    return parts


def gemv_flags(pre_epi=0, deep=0, shallow=0):
    return (GEMV_PRE_EPI if pre_epi else 0) | (GEMV_DEEP if deep else 0) | (GEMV_SHALLOW if shallow else 0)


def softmax_relpos(ac, bd, p, scale, key_lens=None):
    """ac (Z1,Z2,Tq,>=Tk), bd (Z1,Z2,Tq,>=2Tk-1) or None, p (Z1,Z2,Tq,ld_p) (pad columns written as 0)."""
    Z1, Z2, Tq = ac.shape[:3]
    Tk = Tq
    assert ac.is_contiguous() or ac.stride(3) == 1
    check(lib.cbx_softmax_relpos_f32(_p(ac), _p(bd), _p(p), _p(key_lens), Z1, Z2, Tq, Tk, ac.stride(2),
                                     0 if bd is None else bd.stride(2), p.stride(2), ac.stride(1),
                                     0 if bd is None else bd.stride(1), p.stride(1), scale, _stream()),
          "cbx_softmax_relpos_f32")
    return p


def softmax_rows(s, p, scale, n_keys, key_lens=None):
    """Plain row softmax of s (Z1,Z2,Tq,>=n_keys) into p (pad columns zeroed): perceiver attention."""
    Z1, Z2, Tq = s.shape[:3]
    check(lib.cbx_softmax_relpos_f32(_p(s), None, _p(p), _p(key_lens), Z1, Z2, Tq, n_keys, s.stride(2), 0, p.stride(2),
                                     s.stride(1), 0, p.stride(1), scale, _stream()), "cbx_softmax_relpos_f32")
    return p


def act(x, out, kind, param=None, slope=0.0):
    rows, C = x.shape
    check(lib.cbx_act_f32(_p(x), _p(out), _p(param), rows, C, x.stride(0), out.stride(0), kind, slope, _stream()), "cbx_act_f32")
    return out


def axpby(x, y, a=1.0, b=0.0):
    """y = a*x + b*y on 2-D strided views."""
    rows, C = x.shape
    assert x.stride(1) == 1 and y.stride(1) == 1
    check(lib.cbx_axpby_f32(_p(x), _p(y), rows, C, x.stride(0), y.stride(0), a, b, _stream()), "cbx_axpby_f32")
    return y


def embed(ids, table, out, table2=None, ids2=None, scale=1.0, out_packed=False):
    """out_packed: out is the packed operand image (ceil(rows/16)*16, C) of a decode gemv; rows = len(ids)."""
    rows, C = (ids.numel(), out.shape[1]) if out_packed else out.shape
    assert ids.dtype == torch.int64 and (ids2 is None or ids2.dtype == torch.int32)
    check(lib.cbx_embed_f32(_p(ids), _p(table), _p(table2), _p(ids2), _p(out), rows, C, out.stride(0), scale, 3 if out_packed else 1, _stream()),
          "cbx_embed_f32")
    return out


def rope_kv(qkv, positions, cos_t, sin_t, kc, vc, n_heads, cache_rows=None):
    """In-place RoPE on the q,k parts of qkv (rows, 3*H*64) + append k,v at `positions` into caches (rows,H,max,64)."""
    n_rows = qkv.shape[0]
    check(lib.cbx_rope_kv_f32(_p(qkv), _p(positions), _p(cos_t), _p(sin_t), _p(kc), _p(vc), _p(cache_rows), n_rows, n_heads,
                              qkv.stride(0), 0 if kc is None else kc.stride(0), 0 if kc is None else kc.stride(1),
                              _stream()), "cbx_rope_kv_f32")


def cfm_euler(xin, v, B, T, C, dt, w, cfg=True):
    """xin (rows,T,ld) packed estimator input whose first C columns are x; v (rows,T,C)."""
    check(lib.cbx_cfm_euler_f32(_p(xin), _p(v), B, T, C, xin.stride(1), v.stride(1), xin.stride(0), v.stride(0), dt, w,
                                int(cfg), _stream()), "cbx_cfm_euler_f32")


def t3_sample(**kw):
    p = SamplerParams()
    for k, val in kw.items():
        setattr(p, k, _p(val) if torch.is_tensor(val) else val)
    check(lib.cbx_t3_sample(ctypes.byref(p), _stream()), "cbx_t3_sample")


def hift_source(f0, phase, noise, lin_w, lin_b, s, frame_cum, up=480, sr=24000.0):
    B, T = f0.shape
    check(lib.cbx_hift_source_f32(_p(f0), _p(phase), _p(noise), _p(lin_w), float(lin_b), _p(s), _p(frame_cum), B, T, up, sr,
                                  _stream()), "cbx_hift_source_f32")
    return s


def hift_stft(s, spec, sample_lens=None):
    B, L = s.shape
    check(lib.cbx_hift_stft_f32(_p(s), _p(spec), _p(sample_lens), B, L, spec.stride(1), _stream()), "cbx_hift_stft_f32")
    return spec


def hift_istft(x, wav, clamp=0.99, fade_n=0):
    B, frames = x.shape[0], x.shape[1]
    check(lib.cbx_hift_istft_f32(_p(x), _p(wav), B, frames, x.stride(1), clamp, fade_n, _stream()), "cbx_hift_istft_f32")
    return wav


# ----------------------------------------------------------------------------- front-end (prompt analysis / voice conversion) kernels

UN_LOG_CLAMP, UN_LOG10_CLAMP, UN_FLOOR_AFFINE, UN_AFFINE = 1, 2, 3, 4


def dwconv1d(x, w, y, taps, pad_left, lens=None, add_input=False):
    """Depthwise conv, channel-last: x, y (B, T, C); w (C, taps)."""
    B, T, C = x.shape
    check(lib.cbx_dwconv1d_f32(_p(_f32(x, "x")), _p(_f32(w, "w")), _p(y), _p(lens), B, T, C, taps, pad_left, x.stride(1), y.stride(1),
                               x.stride(0), y.stride(0), int(add_input), _stream()), "cbx_dwconv1d_f32")
    return y


def lstm_cell(pre, hh, c, h):
    B, H = c.shape
    check(lib.cbx_lstm_cell_f32(_p(pre), _p(hh), _p(c), _p(h), B, H, pre.stride(0), hh.stride(0), c.stride(0), h.stride(0), _stream()),
          "cbx_lstm_cell_f32")


def affine_act(x, y, scale, shift, act=NONE):
    rows, C = x.shape
    check(lib.cbx_affine_act_f32(_p(_f32(x, "x")), _p(y), _p(scale), _p(shift), rows, C, x.stride(0), y.stride(0), act, _stream()),
          "cbx_affine_act_f32")
    return y


def cplx_power(spec, out, mode=0, eps=0.0):
    rows, F = out.shape
    check(lib.cbx_cplx_power_f32(_p(_f32(spec, "spec")), _p(out), rows, F, spec.stride(0), out.stride(0), mode, eps, _stream()),
          "cbx_cplx_power_f32")
    return out


def unary(x, y, op, a=0.0, b=0.0, dev_scalar=None):
    rows, C = x.shape
    check(lib.cbx_unary_f32(_p(_f32(x, "x")), _p(y), rows, C, x.stride(0), y.stride(0), op, a, b, _p(dev_scalar), _stream()), "cbx_unary_f32")
    return y


def reduce_max(x, out):
    rows, C = x.shape
    check(lib.cbx_reduce_max_f32(_p(_f32(x, "x")), _p(out), rows, C, x.stride(0), _stream()), "cbx_reduce_max_f32")
    return out


def seg_context(x, ctx, seg_len=100):
    T, C = x.shape
    check(lib.cbx_seg_context_f32(_p(_f32(x, "x")), _p(ctx), T, C, seg_len, x.stride(0), ctx.stride(0), _stream()), "cbx_seg_context_f32")
    return ctx


def seg_gate_mul(y, m, seg_len=100):
    T, C = y.shape
    check(lib.cbx_seg_gate_mul_f32(_p(y), _p(m), T, C, seg_len, y.stride(0), m.stride(0), _stream()), "cbx_seg_gate_mul_f32")
    return y


def stats_pool(x, out):
    T, C = x.shape
    check(lib.cbx_stats_pool_f32(_p(_f32(x, "x")), _p(out), T, C, x.stride(0), _stream()), "cbx_stats_pool_f32")
    return out


def fsq_index(h, idx):
    check(lib.cbx_fsq_index(_p(_f32(h, "h")), _p(idx), h.shape[0], h.stride(0), _stream()), "cbx_fsq_index")
    return idx
