"""Generate tests/golden/*.npz by running the UNMODIFIED reference (authoring container only).

    python tests/golden/make_golden.py            # writes fixtures, prints oracle-vs-reference errors

The reference modules are imported through oracle/ref_import.py, loaded with the
seeded synthetic checkpoints of chatterbox_amd/synth.py (same key layout as the
real checkpoints) and driven with injected RNG so that every stage is
deterministic:

  * torch.multinomial (t3.py:360)           -> inverse-CDF on injected uniforms
  * torch.randn_like  (flow_matching.py:216)-> injected z
  * torch.randn_like / Uniform.sample (hifigan.py:212-226,282) -> injected phase / noise

Fixtures are small slices (KBs); tests regenerate weights from the seeds.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from provenance import provenance  # noqa: E402
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)

from chatterbox_amd import synth  # noqa: E402
from oracle import ref_import, ref_torch as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
LOGIT_IDX = torch.arange(0, 8194, 16)


def fingerprint(sd):
    keys = sorted(sd)[:: max(1, len(sd) // 16)]
    return np.array([float(sd[k].double().sum()) for k in keys])


# ----------------------------------------------------------------------------- T3


def run_ref_t3(n_layers, steps, n_text, seed=0, **samp):
    T3, T3Config, T3Cond, lc = ref_import.load_T3()
    lc.LLAMA_CONFIGS["Llama_520M"]["num_hidden_layers"] = n_layers
    sd = synth.t3_state_dict(n_layers, seed)
    m = T3(T3Config.multilingual()).eval()
    m.load_state_dict(sd, strict=True)
    ci = synth.t3_cond()
    cond = T3Cond(speaker_emb=ci["speaker_emb"], cond_prompt_speech_tokens=ci["cond_prompt_speech_tokens"],
                  emotion_adv=ci["emotion_adv"])
    tt = synth.text_tokens(n_text)
    tt2 = torch.stack([tt, tt])
    u = synth.rand((steps,), seed=7)
    raw, probs_seen, step = [], [], [0]
    hook = m.speech_head.register_forward_hook(lambda mod, i, o: raw.append(o[:, -1].detach().clone()))
    orig = torch.multinomial

    def fake_multinomial(probs, num_samples=1, **kw):
        probs_seen.append(probs[0].clone())
        p = probs[0].clone()
        p[O.STOP_SPEECH] = 0.0  # EOS banned (fixed-length synthetic run)
        tok = O.sample_inverse_cdf(p, u[step[0]])
        step[0] += 1
        return torch.tensor([[tok]])

    torch.multinomial = fake_multinomial
    try:
        toks = m.inference(t3_cond=cond, text_tokens=tt2, max_new_tokens=steps, **samp)
    finally:
        torch.multinomial = orig
        hook.remove()
    return dict(sd=sd, ci=ci, tt2=tt2, u=u, tokens=toks[0], raw=torch.stack(raw[:steps]), probs=torch.stack(probs_seen))


def golden_t3(name, n_layers, steps, n_text):
    samp = dict(temperature=0.8, cfg_weight=0.5, repetition_penalty=1.2, min_p=0.05, top_p=1.0)
    r = run_ref_t3(n_layers, steps, n_text, **samp)
    # oracle restatement on the same inputs
    toks, logits = O.t3_inference(r["sd"], n_layers, r["ci"], r["tt2"], steps, r["u"], ban_eos=True,
                                  return_logits=True, **samp)
    err = (logits - r["raw"]).abs().max().item()
    print(f"[{name}] ref tokens {r['tokens'].tolist()}")
    print(f"[{name}] oracle-vs-reference raw logits max-abs {err:.3e}; tokens equal: {torch.equal(toks, r['tokens'])}")
    assert err < 2e-3 and torch.equal(toks, r["tokens"])
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), provenance=provenance(os.path.basename(__file__)), n_layers=n_layers, steps=steps, n_text=n_text,
                        tokens=r["tokens"].numpy(), logits_sub=r["raw"][:, :, LOGIT_IDX].numpy(),
                        logit_idx=LOGIT_IDX.numpy(), uniforms=r["u"].numpy(), fp=fingerprint(r["sd"]))


# ----------------------------------------------------------------------------- S3Gen


def golden_s3gen(name, P, N, n_steps=10):
    S3 = ref_import.load_S3Gen()
    sd = synth.s3gen_state_dict(0)
    m = S3().eval()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith(("tokenizer.", "speaker_encoder.")) for k in missing), [k for k in missing][:5]
    ref = synth.s3gen_ref(n_prompt_tokens=P)
    toks = synth.speech_tokens(N)[None]
    T = 2 * (P + N)
    z = synth.randn((1, 80, T), seed=5)
    phase = (synth.rand((1, 9, 1), seed=6) * 2 - 1) * np.pi
    phase[:, 0] = 0
    noise = synth.randn((1, 9, 480 * 2 * N), seed=6)

    import torch.distributions.uniform as U
    o_rl, o_us = torch.randn_like, U.Uniform.sample
    try:
        torch.randn_like = lambda t, **kw: z.clone()
        mel = m.flow_inference(toks, ref_dict=dict(ref), n_cfm_timesteps=n_steps, finalize=True)
        calls = [0]

        def fake_rl(t, **kw):
            calls[0] += 1
            return noise.clone() if t.shape == noise.shape else torch.zeros_like(t)

        torch.randn_like = fake_rl
        U.Uniform.sample = lambda self, sample_shape=torch.Size(): phase.clone()
        wav, src = m.hift_inference(mel)
    finally:
        torch.randn_like, U.Uniform.sample = o_rl, o_us
    wav = wav.clone()
    wav[:, : len(m.trim_fade)] *= m.trim_fade

    # oracle restatement
    o_wav, o_mel = O.s3gen_inference(sd, toks, torch.tensor([N]), ref, z, phase, noise, n_steps)
    o_wav2, o_src = O.hift_inference(sd, mel, phase, noise)
    e_mel = (o_mel - mel).abs()
    e_src = (o_src - src).abs().max().item()
    e_w2 = (O.trim_fade(o_wav2) - wav).pow(2).mean().sqrt().item()
    e_w = (o_wav - wav).pow(2).mean().sqrt().item()
    print(f"[{name}] mel std {mel.std():.3f}; oracle-vs-ref mel L1 {e_mel.mean():.3e} max {e_mel.max():.3e}")
    print(f"[{name}] wav rms {wav.pow(2).mean().sqrt():.4f} |max| {wav.abs().max():.3f}; source max err {e_src:.3e}; "
          f"wav RMSE (same mel) {e_w2:.3e}; wav RMSE (end-to-end) {e_w:.3e}")
    f0 = O.f0_predict(sd, mel)
    print(f"[{name}] f0 range {f0.min():.1f}..{f0.max():.1f}, voiced frac {(f0 > 10).float().mean():.2f}")
    assert e_mel.mean() < 1e-4 and e_src < 1e-3 and e_w2 < 1e-4
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), provenance=provenance(os.path.basename(__file__)), P=P, N=N, n_steps=n_steps, mel=mel[0].numpy(),
                        wav=wav[0].numpy(), src=src[0, 0, ::7].numpy(), fp=fingerprint(sd))


# ----------------------------------------------------------------------------- Turbo / Nano (GPT-2 T3, meanflow S3Gen)


def golden_t3_turbo(name, n_layers, d, steps, n_text, cfg_name):
    T3, T3Config, T3Cond, lc = ref_import.load_T3()
    lc.LLAMA_CONFIGS[cfg_name]["n_layer"] = n_layers
    hp = T3Config(text_tokens_dict_size=50276)
    hp.llama_config_name, hp.speech_tokens_dict_size, hp.input_pos_emb = cfg_name, 6563, None
    hp.speech_cond_prompt_len, hp.use_perceiver_resampler, hp.emotion_adv = 375, False, False
    sd = synth.t3_turbo_state_dict(n_layers, d, 0, include_wte=True)
    m = T3(hp).eval()
    m.load_state_dict(sd, strict=True)
    del m.tfmr.wte
    n_heads = d // 64
    ci = synth.t3_cond(prompt_len=375)
    cond = T3Cond(speaker_emb=ci["speaker_emb"], cond_prompt_speech_tokens=ci["cond_prompt_speech_tokens"], emotion_adv=None)
    tt = synth.turbo_text_tokens(n_text)
    u = synth.rand((steps + 1,), seed=7)
    raw, step = [], [0]
    hook = m.speech_head.register_forward_hook(lambda mod, i, o: raw.append(o[:, -1].detach().clone()))
    orig = torch.multinomial

    def fake_multinomial(probs, num_samples=1, **kw):
        p = probs[0].clone()
        p[6562] = 0.0
        tok = O.sample_inverse_cdf(p, u[step[0]])
        step[0] += 1
        return torch.tensor([[tok]])

    torch.multinomial = fake_multinomial
    try:
        toks = m.inference_turbo(cond, tt[None], temperature=0.8, top_k=1000, top_p=0.95, repetition_penalty=1.2, max_gen_len=steps)
    finally:
        torch.multinomial = orig
        hook.remove()
    raw = torch.stack(raw[: steps + 1])[:, 0]
    o_toks, o_logits = O.t3_inference_turbo(sd, n_layers, n_heads, ci, tt, steps, u, ban_eos=True, return_logits=True)
    err = (o_logits - raw).abs().max().item()
    print(f"[{name}] ref tokens {toks[0].tolist()}\n[{name}] oracle-vs-reference logits max-abs {err:.3e}; tokens equal: {torch.equal(o_toks, toks[0])}")
    assert err < 2e-3 and torch.equal(o_toks, toks[0])
    idx = torch.arange(0, 6563, 13)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), provenance=provenance(os.path.basename(__file__)), n_layers=n_layers, d=d, steps=steps, n_text=n_text, tokens=toks[0].numpy(),
                        logits_sub=raw[:, idx].numpy(), logit_idx=idx.numpy(), uniforms=u.numpy(), fp=fingerprint({k: v for k, v in sd.items() if k != "tfmr.wte.weight"}))


def golden_meanflow(name, P, N):
    S3 = ref_import.load_S3Gen()
    sd = synth.s3gen_state_dict(0, meanflow=True)
    m = S3(meanflow=True).eval()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith(("tokenizer.", "speaker_encoder.")) for k in missing)
    ref = synth.s3gen_ref(n_prompt_tokens=P)
    toks = synth.speech_tokens(N)[None]
    T = 2 * (P + N)
    z = synth.randn((1, 80, T), seed=5)
    o_rl, o_rn = torch.randn_like, torch.randn
    try:
        torch.randn_like = lambda t, **kw: z.clone()
        torch.randn = lambda *a, **kw: z[:, :, 2 * P:].clone()  # the generated-region noise drawn in flow_inference (s3gen.py:314-316)
        mel = m.flow_inference(toks, ref_dict=dict(ref), n_cfm_timesteps=2, finalize=True)
    finally:
        torch.randn_like, torch.randn = o_rl, o_rn
    o_mel = O.flow_inference(sd, toks, torch.tensor([N]), ref, z, 2, meanflow=True)
    e = (o_mel - mel).abs()
    print(f"[{name}] meanflow mel std {mel.std():.3f}; oracle-vs-ref L1 {e.mean():.3e} max {e.max():.3e}")
    assert e.mean() < 1e-4
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), provenance=provenance(os.path.basename(__file__)), P=P, N=N, mel=mel[0].numpy(), fp=fingerprint(sd))


if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1:] or ["t3_l2", "t3_l30", "s3gen_small", "turbo_l2", "nano_l12", "meanflow_small"]
    with torch.inference_mode():
        if "t3_l2" in which:
            golden_t3("t3_l2", 2, 12, 16)
        if "t3_l30" in which:
            golden_t3("t3_l30", 30, 6, 24)
        if "s3gen_small" in which:
            golden_s3gen("s3gen_small", P=12, N=20)
        if "turbo_l2" in which:
            golden_t3_turbo("turbo_l2", 2, 1024, 10, 20, "GPT2_medium")
        if "nano_l12" in which:
            golden_t3_turbo("nano_l12", 12, 768, 6, 16, "GPT2_small")
        if "meanflow_small" in which:
            golden_meanflow("meanflow_small", P=10, N=16)
