"""Accuracy + speed of the split-bf16 GEMM modes (cbx_gemm_t.precision 3 / 6) against the exact fp32 MFMA path.

(1) op level: linear / conv shapes incl. the edge cases (ragged K tail, odd tile counts, taps, stride, dilation, upsample,
    lens masking, residual + beta + second output) against an fp64 CPU reference;
(2) the CFM golden fixture (mel from the UNMODIFIED reference) and the HiFT golden waveform, per precision;
(3) wall time of one flow + vocoder pass at the bench shape, per precision.
"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from chatterbox_amd import ops, synth, weights
from chatterbox_amd.hift import HiFTEngine
from chatterbox_amd.s3gen import FlowEngine

dev = torch.device("cuda:0")
R = lambda shape, seed, scale=1.0: torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def relerr(got, ref):
    got, ref = got.double().cpu(), ref.double()
    return float((got - ref).abs().max() / (ref.abs().mean() + 1e-30)), float((got - ref).abs().mean() / (ref.abs().mean() + 1e-30))


def op_checks(prec):
    worst = 0.0
    with ops.gemm_precision(prec):
        for (M, N, K) in [(1000, 256, 256), (129, 257, 320), (300, 1536, 256), (4000, 80, 1000), (513, 1024, 80), (2048, 64, 1024), (33, 100, 36)]:
            x, w, b, r = R((M, K), 1), R((N, K), 2, 1 / math.sqrt(K)), R((N,), 3), R((M, N), 4)
            out = torch.empty(M, N, device=dev)
            ops.linear(x.to(dev), w.to(dev), out, bias=b.to(dev), act=ops.GELU_ERF, residual=r.to(dev))
            ref = F.gelu(F.linear(x.double(), w.double(), b.double())) + r.double()
            e = relerr(out, ref)
            worst = max(worst, e[0])
            print(f"  prec {prec} linear {M}x{N}x{K}: max/mean rel err {e[0]:.2e} {e[1]:.2e}")
        # accumulate + second output on strided views
        M, N, K = 200, 96, 64
        xb, w, ap, cb = R((M, K + 8), 1), R((N, K), 2, 0.1), 1.0 + 0.2 * R((N,), 5), R((M, N + 4), 6)
        out, out2 = cb.clone().to(dev), torch.empty(M, N, device=dev)
        ops.linear(xb.to(dev)[:, :K], w.to(dev), out[:, :N], alpha=1.0 / 3, beta=1.0, out2=out2, act2=ops.SNAKE, act2_param=ap.to(dev))
        ref = cb[:, :N].double() + F.linear(xb[:, :K].double(), w.double()) / 3
        e = relerr(out[:, :N], ref)
        e2 = relerr(out2, ref + (1.0 / (ap[None].double() + 1e-9)) * torch.sin(ref * ap[None].double()) ** 2)
        assert torch.equal(out[:, N:].cpu(), cb[:, N:]), "pad columns touched"
        worst = max(worst, e[0], e2[0])
        print(f"  prec {prec} accumulate/second output: {e[0]:.2e} {e2[0]:.2e}")
        for (cin, cout, k, dil, stride, pad, T) in [(32, 48, 3, 1, 1, 1, 77), (64, 64, 11, 5, 1, 25, 300), (32, 256, 30, 1, 15, 7, 1201),
                                                    (320, 256, 3, 1, 1, 2, 100), (256, 256, 3, 1, 1, 2, 1000), (512, 512, 3, 1, 1, 1, 64)]:
            B = 3
            x, w, b = R((B, cin, T), 1), R((cout, cin, k), 2, 1 / math.sqrt(cin * k)), R((cout,), 3)
            causal = (pad == k - 1 and dil == 1 and cin >= 256)
            ref = F.conv1d(F.pad(x.double(), (pad, 0)) if causal else x.double(), w.double(), b.double(), stride=stride, dilation=dil,
                           padding=0 if causal else pad)
            out = torch.empty(B, ref.shape[2], cout, device=dev)
            ops.conv1d(x.transpose(1, 2).contiguous().to(dev), weights.pack_conv(w).to(dev), out, taps=k, cin=cin, bias=b.to(dev),
                       dil=dil, stride=stride, pad_left=pad)
            e = relerr(out.transpose(1, 2), ref)
            worst = max(worst, e[0])
            print(f"  prec {prec} conv cin{cin} cout{cout} k{k} d{dil} s{stride}: {e[0]:.2e} {e[1]:.2e}")
        B, C, T = 3, 32, 50
        lens = torch.tensor([50, 31, 7], dtype=torch.int32)
        x, w, b = R((B, C, T), 1), R((C, C, 4), 2, 0.1), R((C,), 3)
        out = torch.empty(B, T, C, device=dev)
        ops.conv1d(x.transpose(1, 2).contiguous().to(dev), weights.pack_conv(w).to(dev), out, taps=4, cin=C, bias=b.to(dev), pad_left=0,
                   lens=lens.to(dev))
        for i in range(B):
            n = int(lens[i])
            e = relerr(out[i, :n].t(), F.conv1d(F.pad(x[i:i + 1, :, :n].double(), (0, 3)), w.double(), b.double())[0])
            worst = max(worst, e[0])
        w5 = R((C, C, 5), 4, 0.1)
        ref = F.conv1d(F.pad(x.double().repeat_interleave(2, dim=2), (4, 0)), w5.double(), b.double())
        out = torch.empty(B, 2 * T, C, device=dev)
        ops.conv1d(x.transpose(1, 2).contiguous().to(dev), weights.pack_conv(w5).to(dev), out, taps=5, cin=C, bias=b.to(dev), pad_left=4, up=2)
        e = relerr(out.transpose(1, 2), ref)
        worst = max(worst, e[0])
        print(f"  prec {prec} ragged + upsample conv: {e[0]:.2e}")
    print(f"prec {prec}: worst max-rel err {worst:.3e}")
    return worst


def golden(prec):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from test_models_gpu import _s3_inputs, _hift_noise, GOLD
    g = np.load(os.path.join(GOLD, "s3gen_small.npz"))
    P, N = int(g["P"]), int(g["N"])
    sd = synth.s3gen_state_dict(0)
    eng = FlowEngine(sd, dev, precision=prec)
    ref, toks, lens = _s3_inputs(P, [N])
    z = synth.randn((1, 80, 2 * (P + N)), seed=5)
    mel = eng.inference(toks, lens, ref, z=z.transpose(1, 2).contiguous(), n_steps=int(g["n_steps"]))
    err = (mel[0].cpu() - torch.from_numpy(g["mel"]).t()).abs()
    h = HiFTEngine(sd, dev, precision=prec)
    gm = torch.from_numpy(g["mel"])[None]
    phase, noise = _hift_noise(1, gm.shape[2])
    wav, _ = h.inference(gm.transpose(1, 2).contiguous().to(dev), phase, noise)
    rmse = (wav[0].cpu() - torch.from_numpy(g["wav"])).pow(2).mean().sqrt().item()
    print(f"prec {prec}: golden mel L1 {err.mean():.3e} max {err.max():.3e} | HiFT wav RMSE vs reference {rmse:.3e}", flush=True)


def speed(prec):
    sd = synth.s3gen_state_dict(0)
    flow, hift = FlowEngine(sd, dev, precision=prec), HiFTEngine(sd, dev, precision=prec)
    B, N = 8, 250
    toks = torch.stack([synth.speech_tokens(N, seed=b) for b in range(B)]).to(dev)
    lens = torch.full((B,), N, dtype=torch.int32, device=dev)
    ref = synth.s3gen_ref()
    mels = None
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mel = flow.inference(toks, lens, ref, z=synth.randn((B, 2 * (250 + N), 80), seed=9).to(dev))
        torch.cuda.synchronize(); t1 = time.perf_counter()
        wav, _ = hift.inference(mel)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"prec {prec} tile {os.environ.get('CBX_SPLIT_TILE', 'auto')}: flow {1e3 * (t1 - t0):.1f} ms  hift {1e3 * (t2 - t1):.1f} ms", flush=True)
    return mel


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "ops"):
        for prec in (1, 6, 3):
            op_checks(prec)
    if what in ("all", "golden"):
        for prec in (1, 6, 3):
            golden(prec)
    if what in ("all", "speed"):
        base = None
        for prec in (1, 6, 3):
            mel = speed(prec)
            if base is None:
                base = mel
            else:
                d = (mel - base).abs()
                print(f"prec {prec}: bench-shape mel vs exact: L1 {d.mean():.3e} max {d.max():.3e}")
