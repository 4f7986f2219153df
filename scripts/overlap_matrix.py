"""What does the T3 decode step cost beside ONE kind of flow kernel?  (run on the GPU box, unprofiled)
Stream A (high priority): T3 generate() of CBX_OV_TOKENS tokens at the bench shape (hipGraph replays), timed with its own events.
Stream B: a loop of one flow kernel at the CFM shape (rows 16 x T 1000) -- plane attention version 4 / 5, the q | k | V^T projection, ff1 + GELU, ff2, the
out projection, each on the default tile and on co-residency tiles, the planes LayerNorm -- long enough to outlast the T3 run.
Per kernel: its launches / s alone and beside T3, T3's ms per token alone and beside it."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chatterbox_amd import _lib, ops, synth
from chatterbox_amd.engine import ChatterboxEngine

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
L = int(os.environ.get("CBX_AB_LAYERS", "30"))
NT = int(os.environ.get("CBX_OV_TOKENS", "60"))
eng = ChatterboxEngine(synth.t3_state_dict(L, 0), synth.s3gen_state_dict(0), dev, n_t3_layers=L)
B = 8
t3c = synth.t3_cond(prompt_len=150)
texts = [synth.text_tokens(64, seed=b) for b in range(B)]
g = torch.Generator(device=dev).manual_seed(1)
u = torch.rand(B, NT, generator=g, device=dev)
kw = dict(max_new_tokens=NT, uniforms=u, ban_eos=True, ban_from=6561)
polite = os.environ.get("CBX_OV_T3_POLITE", "1") == "1"
if polite:
    eng.t3.apply_variant(dict(eng.t3.tune, half_tiles=0, d_ks2=4, d_nw2=8), dict(eng.t3.knobs, shallow=1))
eng.t3.generate(t3c, texts, **kw)
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(device=dev, priority=-1), torch.cuda.Stream(device=dev)

ROWS, T = 16, 1000
M = ROWS * T
h = torch.randn(M, 256, device=dev)
hP = ops.split_planes(h)
wqkv, w1, w2, wo = (ops.split_planes(torch.randn(n, k, device=dev) * 0.05) for n, k in ((1536, 256), (1024, 256), (256, 1024), (256, 512)))
qkP, vtP, attP, ffP = ops.Planes(M, 1024, dev), ops.Planes(ROWS * 512, T, dev, zero=True), ops.Planes(M, 512, dev), ops.Planes(M, 1024, dev)
x = torch.randn(M, 256, device=dev)
b1, b2, lw, lb = torch.randn(1024, device=dev), torch.randn(256, device=dev), torch.ones(256, device=dev), torch.zeros(256, device=dev)
ops.split_planes(torch.randn(M, 1024, device=dev), qkP)
ops.split_planes(torch.randn(ROWS * 512, T, device=dev), vtP)
ops.split_planes(torch.randn(M, 512, device=dev), attP)
ops.split_planes(torch.randn(M, 1024, device=dev), ffP)
lens = torch.full((ROWS,), T, dtype=torch.int32, device=dev)


def attn():
    ops.flash_attn_planes(qkP.cols(0, 512), qkP.cols(512, 512), vtP, attP, Z=ROWS, H=8, T=T, vt_sb=512 * vtP.ld, scale=0.125, key_lens=lens)


def qkv():
    ops.gemm_planes(hP, wqkv, M=M, N=1536, K=256, P=qkP, PT=vtP, pt_n0=1024, pt_T=T, pt_zs=512 * vtP.ld)


def ff1():
    ops.linear_planes(hP, w1, outp=ffP, bias=b1, act=ops.GELU_ERF)


def ff2():
    ops.linear_planes(ffP, w2, out=x, bias=b2, residual=x)


def outp():
    ops.linear_planes(attP, wo, out=x, bias=b2, residual=x)


def ln():
    ops.layernorm_planes(x, lw, lb, hP, 1e-5)


def t3_run():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sa):
        e0.record()
        eng.t3.generate(t3c, texts, async_mode=True, **kw)
        e1.record()
    return e0, e1


def bg_run(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sb):
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
    return e0, e1


torch.cuda.synchronize()
a0, a1 = t3_run()
torch.cuda.synchronize()
t3_alone = a0.elapsed_time(a1)  # includes the prefill
print(json.dumps(dict(t3_alone_ms=round(t3_alone, 1), tokens=NT, t3_polite_geometry=polite)), flush=True)
cases = [("attention v4 (8 waves x 194 VGPRs)", attn, dict(attn=4)), ("attention v5 (4 waves, 1 workgroup per CU)", attn, dict(attn=5)),
         ("q|k|V^T, default tile", qkv, dict(tile=0)), ("q|k|V^T, tile 17", qkv, dict(tile=17)), ("q|k|V^T, tile 32 (4 loader waves)", qkv, dict(tile=32)),
         ("ff1 + GELU, default tile", ff1, dict(tile=0)), ("ff1 + GELU, tile 17", ff1, dict(tile=17)),
         ("ff2, default tile", ff2, dict(tile=0)), ("ff2, tile 17", ff2, dict(tile=17)),
         ("out projection, default tile", outp, dict(tile=0)), ("out projection, tile 17", outp, dict(tile=17)),
         ("planes LayerNorm", ln, dict())]
for name, fn, knobs in cases:
    _lib.lib.cbx_set_planes_tile(knobs.get("tile", 0))
    _lib.lib.cbx_set_attn_planes_version(knobs.get("attn", 4))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g0, g1 = bg_run(fn, 200)
    torch.cuda.synchronize()
    us = g0.elapsed_time(g1) / 200 * 1e3
    n = int(4.0 * t3_alone * 1e3 / us) + 50  # outlasts a 3x slower T3 (the host needs ~9 us per launch to enqueue them)
    t0 = time.perf_counter()
    g0, g1 = bg_run(fn, n)
    host_ms = 1e3 * (time.perf_counter() - t0)
    a0, a1 = t3_run()
    torch.cuda.synchronize()
    t3_both, bg_both = a0.elapsed_time(a1), g0.elapsed_time(g1)
    print(json.dumps(dict(kernel=name, us_alone=round(us, 1), t3_ms_beside=round(t3_both, 1), t3_slowdown=round(t3_both / t3_alone, 2),
                          bg_ms_total=round(bg_both, 1), bg_ms_if_alone=round(n * us * 1e-3, 1), host_enqueue_ms=round(host_ms, 1),
                          note="bg_ms_total - bg_ms_if_alone = what the flow kernel lost to the co-running T3 (T3 ends first)")), flush=True)
_lib.lib.cbx_set_planes_tile(0)
_lib.lib.cbx_set_attn_planes_version(0)
