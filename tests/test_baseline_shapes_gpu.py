"""Parity AT THE SHAPES BASELINE.json NAMES (-m gpu), against golden vectors produced by the UNMODIFIED reference
(tests/golden/make_golden_big.py -> tests/golden/{t3_l30_b8,s3gen_t1000,vc_t3500,turbo_l24}.npz).

  configs[2]  30-layer T3, B = 8 in ONE device batch, 64 text tokens, 250 sampled tokens per utterance: all 2000 sampled ids equal
              the reference's (which ran the 8 utterances one by one), teacher-forced raw logits <= 1e-3;
              S3Gen at P = 250 / N = 250 (T = 1000 mel frames), 10 Euler steps with CFG, B = 2, in every numerics mode; HiFT at that
              length.
  configs[4]  60 s voice conversion from the token boundary: T = 3500 through the CFG estimator (2 Euler steps) + HiFT.
  configs[1]  Turbo: 24-layer GPT-2-medium T3, B = 1, 64 tokens.

Tolerances are the ones stated once in DESIGN.md section 1 (derived from the measured oracle-vs-reference noise floors printed by
make_golden_big.py); nothing here widens them.
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
SAMP = dict(temperature=0.8, cfg_weight=0.5, repetition_penalty=1.2, min_p=0.05, top_p=1.0)

# DESIGN.md section 1 -- stated tolerances
TOL_LOGITS = 1e-3                                            # T3 raw logits, max-abs
TOL_MEL = {1: (5e-6, 5e-5), 6: (5e-6, 5e-5), 16: (5e-6, 5e-5), 3: (1e-4, 1e-3)}  # CFM mel (L1, max-abs) per numerics mode (16 = default)
TOL_WAV_SAME_SOURCE = 1e-4                                   # HiFT decode, same mel + same source: RMSE of full scale
# HiFT full inference (own F0 -> own source): the waveform is phase-sensitive to F0 (2*pi*h*t*df0), so its tolerance is stated as
# 5x the measured noise floor = the RMSE of the CPU ORACLE against the reference on the same inputs (make_golden_big.py prints it):
#   10 s, reference mel in: floor 2.0e-4 -> 1e-3;   60 s, reference mel in: floor 2.7e-4 -> 1.35e-3;
#   60 s end to end (own CFM mel in): floor 7.4e-4 -> 3.7e-3.
TOL_WAV_FULL = {"s3gen_t1000": 1e-3, "vc_t3500": 1.35e-3}
TOL_WAV_E2E_60S = 3.7e-3
TOL_WAV_BF16_MODE = 2e-2                                     # SURVEY.md 8d "bf16-MFMA mode" waveform tolerance (mode 3, opt-in)
# End to end at the bench shape (T3 tokens -> own CFM mel -> own F0 -> waveform, 8 s): floor 1.31e-4 (oracle vs reference on the same inputs,
# measured when tests/golden/make_golden_e2e.py was run) -> 5x = 6.5e-4; mel of that run: the fp32 tolerance of TOL_MEL
TOL_WAV_E2E_8S = 6.5e-4
# F0 predictor (Hz, max-abs) on the reference's mels: floors 2.3e-4 (500 frames) / 2.7e-4 (3000 frames) -> 5x.  This is the quantity the
# full-inference waveform tolerances above rest on: the excitation phase is 2 pi h * cumsum(f0) / sr
TOL_F0 = {"f0_t1000": 1.2e-3, "f0_t3500": 1.4e-3}


# the floors the waveform / F0 tolerances are multiples of (oracle vs reference on the same inputs, printed by make_golden_big.py / make_golden_e2e.py)
FLOOR = {"s3gen_t1000": 2.0e-4, "vc_t3500": 2.7e-4, "e2e_60s": 7.4e-4, "e2e_8s": 1.31e-4, "f0_t1000": 2.3e-4, "f0_t3500": 2.7e-4}
MEASURED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "measured_vs_floor.jsonl")


def _measured(name, value, tol):
    """Record how far the GPU is from the reference in units of the oracle-vs-reference floor ("nothing tests that the GPU is closer to the
    reference than the oracle's floor x 2") -- appended to gpurun_out/measured_vs_floor.jsonl when that directory exists -- and hold it to 3 x the floor:
    the stated tolerance (5 x) is what DESIGN.md promises, this is the regression guard on what the hardware actually delivers."""
    import json
    floor = FLOOR[name]
    rec = dict(name=name, value=float(value), tolerance=float(tol), floor=floor, ratio_to_floor=round(float(value) / floor, 3))
    if os.path.isdir(os.path.dirname(MEASURED)):
        with open(MEASURED, "a") as f:
            f.write(json.dumps(rec) + "\n")
    assert value <= 3.0 * floor, f"{name}: {value:.3e} is more than 3 x the oracle-vs-reference floor {floor:.1e} (stated tolerance {tol:.1e})"


def _fp(sd):
    keys = sorted(sd)[:: max(1, len(sd) // 16)]
    return np.array([float(sd[k].double().sum()) for k in keys])


def _need(name):
    p = os.path.join(GOLD, name + ".npz")
    if not os.path.exists(p):
        pytest.fail(f"{p} missing: run tests/golden/make_golden_big.py in the authoring container")
    return np.load(p)


# ----------------------------------------------------------------------------- configs[2]: T3 30 L x 250 tokens x B = 8


@pytest.fixture(scope="module")
def t3_30(dev):
    from chatterbox_amd import synth
    from chatterbox_amd.t3 import T3Engine
    sd = synth.t3_state_dict(30, 0)
    return sd, T3Engine(sd, dev)


def test_t3_b8_250_tokens_identical_to_reference(dev, t3_30):
    """The bench batch: 8 utterances x 250 steps through the hipGraph decode loop; every sampled id equals the reference's."""
    from chatterbox_amd import synth
    g = _need("t3_l30_b8")
    sd, eng = t3_30
    np.testing.assert_allclose(_fp(sd), g["fp"], rtol=1e-9, err_msg="synthetic checkpoint drifted from the golden run")
    B, steps, n_text = int(g["B"]), int(g["steps"]), int(g["n_text"])
    texts = [synth.text_tokens(n_text, seed=1 + b) for b in range(B)]
    u = torch.from_numpy(g["uniforms"])
    toks = eng.generate(synth.t3_cond(), texts, max_new_tokens=steps, uniforms=u, ban_eos=True, **SAMP)
    gold = g["tokens"]
    for b in range(B):
        got = np.array(toks[b].tolist())
        nd = int((got != gold[b]).sum())
        first = int(np.argmax(got != gold[b])) if nd else -1
        assert nd == 0, (f"utt {b}: {nd}/{steps} sampled ids differ from the reference, first at step {first} "
                         f"(reference CDF gap there {g['cdf_gap'][b][first]:.2e})")


def test_t3_b2_ragged_250_steps_logits_and_tokens(dev, t3_30):
    """B = 2 (utterances 0 and 3 of the golden set, the second one with a 37-token text: ragged prefill), eager path with the raw
    logits of every step: teacher-forced logits <= 1e-3 at the stored steps and identical ids for the golden utterance; the ragged
    partner must not perturb it (rows never interact)."""
    from chatterbox_amd import synth
    g = _need("t3_l30_b8")
    sd, eng = t3_30
    steps, n_text = int(g["steps"]), int(g["n_text"])
    texts = [synth.text_tokens(n_text, seed=1), synth.text_tokens(37, seed=4)]
    u = torch.stack([torch.from_numpy(g["uniforms"][0]), synth.rand((steps,), seed=99)])
    toks, logits = eng.generate(synth.t3_cond(), texts, max_new_tokens=steps, uniforms=u, ban_eos=True, debug_logits=True, **SAMP)
    assert toks[0].tolist() == g["tokens"][0].tolist()
    sidx = torch.from_numpy(g["step_idx"]).long()
    lidx = torch.from_numpy(g["logit_idx"]).long()
    got = torch.stack([logits[:, 0], logits[:, 2]], 1).cpu()[sidx][:, :, lidx]  # rows 0 (cond) and B+0 (uncond) of utterance 0
    err = (got - torch.from_numpy(g["logits_sub"][0])).abs().max().item()
    assert err <= TOL_LOGITS, f"teacher-forced logits max-abs {err:.3e} over 250 steps"


def test_t3_config3_per_gpu_shape_64_rows(dev, t3_30):
    """configs[3] shards 256 utterances 8-way: 32 per GPU = 64 decode rows (the M = 64 decode path: 4 MFMA row tiles, split-K
    partials + add_rmsnorm).  The 8 golden utterances ride in a 32-utterance device batch; their first 64 sampled ids must equal the
    reference's (sampling is causal: the 250-step golden run's prefix), whatever their neighbours do."""
    from chatterbox_amd import synth
    g = _need("t3_l30_b8")
    sd, eng = t3_30
    steps, n_text = 64, int(g["n_text"])
    where = {0: 0, 5: 1, 9: 2, 14: 3, 17: 4, 22: 5, 27: 6, 31: 7}  # slot in the batch -> golden utterance
    texts, us = [], []
    for slot in range(32):
        if slot in where:
            texts.append(synth.text_tokens(n_text, seed=1 + where[slot]))
            us.append(torch.from_numpy(g["uniforms"][where[slot]][:steps]))
        else:
            texts.append(synth.text_tokens(20 + (slot * 7) % 45, seed=50 + slot))
            us.append(synth.rand((steps,), seed=200 + slot))
    toks = eng.generate(synth.t3_cond(), texts, max_new_tokens=steps, uniforms=torch.stack(us), ban_eos=True, **SAMP)
    for slot, b in where.items():
        assert toks[slot].tolist() == g["tokens"][b][:steps].tolist(), f"golden utterance {b} in slot {slot} of the 64-row batch"


# ----------------------------------------------------------------------------- configs[2]: S3Gen at T = 1000, HiFT at that length


def _s3_case(g, b):
    from chatterbox_amd import synth
    P, N = int(g["P"]), int(g["N"])
    T = 2 * (P + N)
    toks = synth.speech_tokens(N, seed=1 + b)
    z = synth.randn((1, 80, T), seed=5 + 10 * b)
    phase = (synth.rand((1, 9, 1), seed=6 + 10 * b) * 2 - 1) * math.pi
    phase[:, 0] = 0
    noise = synth.randn((1, 9, 960 * N), seed=6 + 10 * b)
    return toks, z, phase, noise


@pytest.fixture(scope="module")
def s3_sd():
    from chatterbox_amd import synth
    return synth.s3gen_state_dict(0)


@pytest.mark.parametrize("prec", [1, 6, 16, 3])
def test_flow_t1000_b2_vs_reference(dev, s3_sd, prec):
    from chatterbox_amd import synth
    from chatterbox_amd.s3gen import FlowEngine
    g = _need("s3gen_t1000")
    np.testing.assert_allclose(_fp(s3_sd), g["fp"], rtol=1e-9)
    P, N, B = int(g["P"]), int(g["N"]), int(g["B"])
    eng = FlowEngine(s3_sd, dev, precision=prec)
    cases = [_s3_case(g, b) for b in range(B)]
    toks = torch.stack([c[0] for c in cases])
    z = torch.cat([c[1] for c in cases]).transpose(1, 2).contiguous()
    mel = eng.inference(toks, torch.tensor([N] * B), synth.s3gen_ref(n_prompt_tokens=P), z=z, n_steps=int(g["n_steps"])).cpu()
    for b in range(B):
        err = (mel[b] - torch.from_numpy(g["mel"][b]).t()).abs()
        tol = TOL_MEL[prec]
        assert err.mean() <= tol[0] and err.max() <= tol[1], f"mode {prec} utt {b}: mel L1 {err.mean():.3e} max {err.max():.3e} (T = {2 * (P + N)})"
    if prec == 16:  # the CO-RESIDENT forms of the throughput schedule (ABI v13: plane GEMMs on tile 17, the 4-wave plane attention): same arithmetic, same mel
        eng.co_resident(True)
        mel2 = eng.inference(toks, torch.tensor([N] * B), synth.s3gen_ref(n_prompt_tokens=P), z=z, n_steps=int(g["n_steps"])).cpu()
        assert torch.equal(mel2, mel), f"co-resident kernel forms changed the mel: max |d| {(mel2 - mel).abs().max():.3e}"
        # LayerNorm as launches of its own (fused_ln = 0: the form of rounds 3-4; or = 2 if the engine runs another mode) must meet the same tolerance, and the forms each other
        eng.co_resident(False)
        eng.fused_ln = 0 if eng.fused_ln else 2
        mel3 = eng.inference(toks, torch.tensor([N] * B), synth.s3gen_ref(n_prompt_tokens=P), z=z, n_steps=int(g["n_steps"])).cpu()
        for b in range(B):
            err = (mel3[b] - torch.from_numpy(g["mel"][b]).t()).abs()
            assert err.mean() <= TOL_MEL[prec][0] and err.max() <= TOL_MEL[prec][1], f"fused_ln {eng.fused_ln}: utt {b}: mel L1 {err.mean():.3e} max {err.max():.3e}"
        assert (mel3 - mel).abs().max() <= 3e-5, f"LayerNorm in the epilogue vs as a launch: max |d| {(mel3 - mel).abs().max():.3e}"


def _window_rmse(wav, g, b):
    starts, win = g["win_start"], g["wav_win"][b]
    d = torch.cat([wav[int(s): int(s) + win.shape[1]] - torch.from_numpy(win[i]) for i, s in enumerate(starts)])
    return d.pow(2).mean().sqrt().item()


@pytest.mark.parametrize("name", ["s3gen_t1000", "vc_t3500"])
def test_hift_full_length_vs_reference(dev, s3_sd, name):
    """HiFT on the REFERENCE's own mel at 10 s (500 frames) and 60 s (3000 frames): full inference (F0 -> source -> decode) against
    windows of the reference waveform, in the shipped numerics mode."""
    from chatterbox_amd.hift import HiFTEngine
    g = _need(name)
    eng = HiFTEngine(s3_sd, dev)
    for b in range(int(g["B"])):
        _, _, phase, noise = _s3_case(g, b)
        mel = torch.from_numpy(g["mel"][b]).t().contiguous()[None].to(dev)
        wav, _ = eng.inference(mel, phase, noise)
        rmse = _window_rmse(wav[0].cpu(), g, b)
        assert rmse <= TOL_WAV_FULL[name], f"{name} utt {b}: waveform RMSE {rmse:.3e} (signal rms {float(g['wav_rms'][b]):.3e})"
        _measured(name, rmse, TOL_WAV_FULL[name])


@pytest.mark.parametrize("name,key", [("s3gen_t1000", "f0_t1000"), ("vc_t3500", "f0_t3500")])
def test_f0_max_abs_at_500_and_3000_frames(dev, s3_sd, name, key):
    """The F0 predictor on the reference's own mel at 10 s and 60 s against the reference's F0 (tests/golden/make_golden_e2e.py), max-abs in
    Hz: bounds the input of the phase integration that the waveform tolerances of the full-inference tests are derived from."""
    from chatterbox_amd.hift import HiFTEngine
    g, e = _need(name), _need("e2e_b8")
    mel = torch.from_numpy(g["mel"][0]).t().contiguous()[None].to(dev)
    f0 = HiFTEngine(s3_sd, dev).f0_predict(mel)[0].cpu()
    ref = torch.from_numpy(e[key])
    assert f0.shape == ref.shape
    err = (f0 - ref).abs()
    assert err.max() <= TOL_F0[key], f"{name}: F0 max-abs {err.max():.3e} Hz at {ref.numel()} frames (f0 up to {ref.max():.0f} Hz), mean {err.mean():.3e}"
    _measured(key, float(err.max()), TOL_F0[key])
    # the voiced / unvoiced decision (f0 > 0 after the abs()) must agree wherever the reference is clearly voiced
    assert bool(((f0 > 1.0) == (ref > 1.0))[ref > 5.0].all())


def test_e2e_synthesize_b8_250_tokens_vs_reference(dev, s3_sd):
    """`ChatterboxEngine.synthesize` AT THE BENCH SHAPE in the default numerics: 8 utterances in one device batch (T3 30 layers, 64 text tokens,
    250 sampled tokens each -> drop_invalid_tokens -> 10-step CFG CFM -> HiFT).  Utterance 0 carries the injected randomness of the
    reference run of tests/golden/make_golden_e2e.py: its tokens, its waveform windows (and, through a second call on those tokens, its
    mel) must match the reference's end-to-end output."""
    from chatterbox_amd import synth
    from chatterbox_amd.engine import ChatterboxEngine, drop_invalid_tokens
    e, t3g = _need("e2e_b8"), _need("t3_l30_b8")
    B, steps, P = int(t3g["B"]), int(t3g["steps"]), int(e["P"])
    eng = ChatterboxEngine(synth.t3_state_dict(30, 0), s3_sd, dev, n_t3_layers=30)
    assert eng.flow.precision == 16 and eng.flow.use_planes
    texts = [synth.text_tokens(int(t3g["n_text"]), seed=1 + b) for b in range(B)]
    u = torch.from_numpy(t3g["uniforms"]).to(dev)
    ns = [int(drop_invalid_tokens(torch.from_numpy(t3g["tokens"][b]).long()).numel()) for b in range(B)]
    N0, Nmax = int(e["N"]), max(ns)
    assert ns[0] == N0
    T0, Tmax = 2 * (P + N0), 2 * (P + Nmax)
    z = synth.randn((B, Tmax, 80), seed=77)
    z[0, :T0] = synth.randn((1, 80, T0), seed=105)[0].t()
    phase = (synth.rand((B, 9, 1), seed=78) * 2 - 1) * math.pi
    phase[0] = (synth.rand((1, 9, 1), seed=106) * 2 - 1)[0] * math.pi
    phase[:, 0] = 0
    noise = synth.randn((B, 9, 960 * Nmax), seed=79)
    noise[0, :, : 960 * N0] = synth.randn((1, 9, 960 * N0), seed=106)[0]
    wavs, st = eng.synthesize(texts, synth.t3_cond(), synth.s3gen_ref(n_prompt_tokens=P), max_new_tokens=steps, uniforms=u, z=z.to(dev),
                              phase=phase, noise=noise, drop_last_token=False, **SAMP)
    assert [int(t.numel()) for t in st] == ns
    assert torch.equal(st[0].cpu(), torch.from_numpy(e["tokens"]).long()), "utterance 0: speech tokens differ from the reference's"
    w0 = wavs[0].float().cpu()
    assert w0.numel() == 960 * N0
    rmse = _window_rmse(w0, dict(win_start=e["win_start"], wav_win=e["wav_win"][None]), 0)
    assert rmse <= TOL_WAV_E2E_8S, f"end-to-end waveform RMSE {rmse:.3e} (signal rms {float(e['wav_rms']):.3e}) in the batch of {B}"
    _measured("e2e_8s", rmse, TOL_WAV_E2E_8S)
    # the CFM mel of utterance 0 inside the ragged batch (vocode returns it)
    _, mel = eng.vocode(st, synth.s3gen_ref(n_prompt_tokens=P), z=z.to(dev), phase=phase, noise=noise)
    err = (mel[0, : 2 * N0].float().cpu() - torch.from_numpy(e["mel"]).t()).abs()
    assert err.mean() <= TOL_MEL[16][0] and err.max() <= TOL_MEL[16][1], f"mel L1 {err.mean():.3e} max {err.max():.3e}"


# ----------------------------------------------------------------------------- configs[4]: 60 s VC, T = 3500, CFG estimator


@pytest.mark.parametrize("prec", [16, 6, 3])
def test_vc_t3500_flow_and_wave_vs_reference(dev, s3_sd, prec):
    from chatterbox_amd import synth
    from chatterbox_amd.hift import HiFTEngine
    from chatterbox_amd.s3gen import FlowEngine
    g = _need("vc_t3500")
    P, N = int(g["P"]), int(g["N"])
    toks, z, phase, noise = _s3_case(g, 0)
    eng = FlowEngine(s3_sd, dev, precision=prec)
    mel = eng.inference(toks[None], torch.tensor([N]), synth.s3gen_ref(n_prompt_tokens=P), z=z.transpose(1, 2).contiguous(),
                        n_steps=int(g["n_steps"]))
    err = (mel[0].cpu() - torch.from_numpy(g["mel"][0]).t()).abs()
    tol = TOL_MEL[prec]
    assert err.mean() <= tol[0] and err.max() <= tol[1], f"mode {prec}: mel L1 {err.mean():.3e} max {err.max():.3e} at T = {2 * (P + N)}"
    del eng
    wav, _ = HiFTEngine(s3_sd, dev, precision=prec).inference(mel, phase, noise)
    assert wav.shape[1] == 960 * N
    rmse = _window_rmse(wav[0].cpu(), g, 0)
    tol_w = TOL_WAV_E2E_60S if prec != 3 else TOL_WAV_BF16_MODE
    assert rmse <= tol_w, f"mode {prec}: 60 s waveform RMSE {rmse:.3e} > {tol_w:.1e}"
    if prec != 3:
        _measured("e2e_60s", rmse, tol_w)


# ----------------------------------------------------------------------------- configs[1]: Turbo 24 layers


def test_turbo_l24_vs_reference(dev):
    from chatterbox_amd import synth
    from chatterbox_amd.t3_turbo import T3TurboEngine
    g = _need("turbo_l24")
    L, d, steps, n_text = int(g["n_layers"]), int(g["d"]), int(g["steps"]), int(g["n_text"])
    sd = synth.t3_turbo_state_dict(L, d, 0)
    np.testing.assert_allclose(_fp(sd), g["fp"], rtol=1e-9)
    eng = T3TurboEngine(sd, dev)
    tt = synth.turbo_text_tokens(n_text)
    cond = synth.t3_cond(prompt_len=375)
    u = torch.from_numpy(g["uniforms"])[None]
    kw = dict(temperature=0.8, top_k=1000, top_p=0.95, repetition_penalty=1.2)
    toks, logits = eng.generate(cond, [tt], max_gen_len=steps, uniforms=u, ban_eos=True, debug_logits=True, **kw)
    sidx, lidx = torch.from_numpy(g["step_idx"]).long(), torch.from_numpy(g["logit_idx"]).long()
    err = (logits.cpu()[:, 0][sidx][:, lidx] - torch.from_numpy(g["logits_sub"])).abs().max().item()
    assert err <= TOL_LOGITS, f"logits max-abs {err:.3e}"
    assert toks[0].tolist() == g["tokens"].tolist()
    toks_g = eng.generate(cond, [tt], max_gen_len=steps, uniforms=u, ban_eos=True, **kw)  # hipGraph path
    assert toks_g[0].tolist() == g["tokens"].tolist()
