"""CPU (-m "not gpu"): the oracle restatement (oracle/ref_torch.py) against the golden vectors that
tests/golden/make_golden.py produced by running the UNMODIFIED reference.  This is what pins the oracle."""
import math
import os

import numpy as np
import pytest
import torch

from chatterbox_amd import synth
from oracle import ref_torch as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SAMP = dict(temperature=0.8, cfg_weight=0.5, repetition_penalty=1.2, min_p=0.05, top_p=1.0)


def _fp(sd):
    keys = sorted(sd)[:: max(1, len(sd) // 16)]
    return np.array([float(sd[k].double().sum()) for k in keys])


@pytest.mark.parametrize("name", ["t3_l2", "t3_l30"])
def test_t3_oracle_matches_reference(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    L, steps, n_text = int(g["n_layers"]), int(g["steps"]), int(g["n_text"])
    sd = synth.t3_state_dict(L, 0)
    np.testing.assert_allclose(_fp(sd), g["fp"], rtol=1e-9)
    tt = synth.text_tokens(n_text)
    with torch.inference_mode():
        toks, logits = O.t3_inference(sd, L, synth.t3_cond(), torch.stack([tt, tt]), steps, torch.from_numpy(g["uniforms"]),
                                      ban_eos=True, return_logits=True, **SAMP)
    idx = torch.from_numpy(g["logit_idx"]).long()
    err = (logits[:, :, idx] - torch.from_numpy(g["logits_sub"])).abs().max().item()
    assert err < 1e-4, err
    assert toks.tolist() == g["tokens"].tolist()


def test_s3gen_oracle_matches_reference():
    g = np.load(os.path.join(GOLD, "s3gen_small.npz"))
    P, N = int(g["P"]), int(g["N"])
    sd = synth.s3gen_state_dict(0)
    np.testing.assert_allclose(_fp(sd), g["fp"], rtol=1e-9)
    ref = synth.s3gen_ref(n_prompt_tokens=P)
    toks = synth.speech_tokens(N)[None]
    z = synth.randn((1, 80, 2 * (P + N)), seed=5)
    phase = (synth.rand((1, 9, 1), seed=6) * 2 - 1) * math.pi
    phase[:, 0] = 0
    noise = synth.randn((1, 9, 960 * N), seed=6)
    with torch.inference_mode():
        wav, mel = O.s3gen_inference(sd, toks, torch.tensor([N]), ref, z, phase, noise, int(g["n_steps"]))
        gm = torch.from_numpy(g["mel"])
        assert (mel[0] - gm).abs().mean() < 1e-5
        # vocoder on the reference's own mel (removes the F0 phase-drift amplification of mel rounding)
        w2, src = O.hift_inference(sd, gm[None], phase, noise)
    gw = torch.from_numpy(g["wav"])
    assert (O.trim_fade(w2)[0] - gw).pow(2).mean().sqrt() < 1e-4
    assert (src[0, 0, ::7] - torch.from_numpy(g["src"])).abs().max() < 1e-3
    assert (wav[0] - gw).pow(2).mean().sqrt() < 1e-3


def test_sampler_semantics_match_hf_processors():
    """process_logits restates the HF logits processors the reference calls (t3.py:320-356)."""
    from transformers.generation.logits_process import (MinPLogitsWarper, RepetitionPenaltyLogitsProcessor,
                                                        TopPLogitsWarper)
    g = torch.Generator().manual_seed(0)
    for top_p, min_p in ((1.0, 0.05), (0.9, 0.02), (0.7, 0.0)):
        c, u = torch.randn(8194, generator=g) * 2, torch.randn(8194, generator=g) * 2
        ids = torch.randint(0, 8194, (1, 40), generator=g)
        l = (c + 0.5 * (c - u))[None]
        l = RepetitionPenaltyLogitsProcessor(1.2)(ids, l) / 0.8
        if min_p > 0:
            l = MinPLogitsWarper(min_p)(ids, l)
        l = TopPLogitsWarper(top_p)(ids, l)
        mine = O.process_logits(c, u, ids[0], 0.5, 0.8, min_p, top_p, 1.2)
        assert torch.equal(torch.isinf(mine), torch.isinf(l[0]))
        keep = ~torch.isinf(mine)
        assert torch.allclose(mine[keep], l[0][keep], atol=1e-6)


TURBO = dict(temperature=0.8, top_k=1000, top_p=0.95, repetition_penalty=1.2)


@pytest.mark.parametrize("name", ["turbo_l2", "nano_l12"])
def test_turbo_oracle_matches_reference(name):
    """GPT-2 backbone T3 (Turbo d = 1024 / Nano d = 768): the oracle against the reference's own `inference_turbo` run."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    L, d, steps, n_text = int(g["n_layers"]), int(g["d"]), int(g["steps"]), int(g["n_text"])
    sd = synth.t3_turbo_state_dict(L, d, 0)
    np.testing.assert_allclose(_fp(sd), g["fp"], rtol=1e-9)
    with torch.inference_mode():
        toks, logits = O.t3_inference_turbo(sd, L, 16 if d == 1024 else 12, synth.t3_cond(prompt_len=375), synth.turbo_text_tokens(n_text),
                                            steps, torch.from_numpy(g["uniforms"]), ban_eos=True, return_logits=True, **TURBO)
    idx = torch.from_numpy(g["logit_idx"]).long()
    err = (logits[:, idx] - torch.from_numpy(g["logits_sub"])).abs().max().item()
    assert err < 1e-4, err
    assert toks.tolist() == g["tokens"].tolist()


def test_meanflow_oracle_matches_reference():
    """2-step meanflow CFM (no CFG) of Turbo / Nano against the reference's golden mel."""
    g = np.load(os.path.join(GOLD, "meanflow_small.npz"))
    P, N = int(g["P"]), int(g["N"])
    sd = synth.s3gen_state_dict(0, meanflow=True)
    np.testing.assert_allclose(_fp(sd), g["fp"], rtol=1e-9)
    z = synth.randn((1, 80, 2 * (P + N)), seed=5)
    with torch.inference_mode():
        mel = O.flow_inference(sd, synth.speech_tokens(N, seed=1)[None], torch.tensor([N]), synth.s3gen_ref(n_prompt_tokens=P), z, 2, meanflow=True)
    assert (mel[0] - torch.from_numpy(g["mel"])).abs().mean() < 1e-5


def test_frontend_oracle_vs_reference_golden():
    """oracle/ref_frontend.py against golden vectors of the UNMODIFIED reference (tests/golden/make_golden_frontend.py): S3 log-mel,
    24 kHz prompt mel, CAMPPlus body, voice-encoder LSTM body.  (S3TokenizerV2.quantize / Kaldi fbank are third-party: unpinned.)"""
    import numpy as np
    from chatterbox_amd import synth
    from oracle import ref_frontend as RF
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "frontend.npz"))
    w16, w24 = synth.prompt_wav(3.0, 16000), synth.prompt_wav(3.0, 24000)
    assert (RF.s3_log_mel(w16) - torch.from_numpy(g["s3_logmel"])).abs().max() <= 1e-5
    assert (RF.mel_spectrogram_24k(w24)[0] - torch.from_numpy(g["mel24k"])).abs().max() <= 1e-5
    fb = RF.kaldi_fbank(w16)
    fb = fb - fb.mean(0, keepdim=True)
    assert (fb - torch.from_numpy(g["fbank"])).abs().max() <= 1e-4  # the stored features ARE this restatement (regression guard)
    xv = RF.campplus_forward(synth.campplus_state_dict(0, prefix=""), torch.from_numpy(g["fbank"])[None])[0]
    assert (xv - torch.from_numpy(g["xvector"])).abs().max() <= 1e-4
    emb = RF.ve_inference(synth.voice_encoder_state_dict(0), torch.from_numpy(g["ve_mel"]))
    assert (emb - torch.from_numpy(g["ve_embed"])).abs().max() <= 1e-5


def test_frontend_host_signal_conditioning(tmp_path):
    """CPU-side pieces of the prompt path (the reference also runs them on the CPU): WAV decoding, resampling, silence trimming, the
    voice encoder's partial-utterance arithmetic."""
    import numpy as np
    from scipy.io import wavfile
    from chatterbox_amd import frontend as fe, synth
    from oracle import ref_frontend as RF
    w = synth.prompt_wav(1.0, 22050).numpy()
    wavfile.write(tmp_path / "a.wav", 22050, (w * 32767).astype(np.int16))
    y, sr = fe.load_wav(tmp_path / "a.wav", 16000)
    assert sr == 16000 and abs(len(y) - 16000) <= 1 and np.abs(y).max() <= 1.0
    # a pure tone keeps its frequency and amplitude through the resampler
    t = np.arange(24000) / 24000.0
    tone = fe.resample(0.5 * np.sin(2 * np.pi * 440 * t).astype(np.float32), 24000, 16000)
    spec = np.abs(np.fft.rfft(tone[2000:10000] * np.hanning(8000)))
    assert abs(np.argmax(spec) * 16000 / 8000 - 440) <= 2.0 and abs(np.abs(tone[2000:10000]).max() - 0.5) < 5e-3
    padded = np.concatenate([np.zeros(8000, np.float32), w[:8000], np.zeros(6000, np.float32)])
    tr = fe.trim_silence(padded, 20.0)
    assert 7000 <= len(tr) <= 10000 and np.array_equal(tr, RF.trim_silence(padded, 20.0))
    assert fe.VoiceEncoderEngine.frame_step(0.5, 1.3) == RF.ve_frame_step(0.5, 1.3) == 77
    for n in (1, 159, 160, 161, 301, 1001):
        assert fe.VoiceEncoderEngine.num_wins(n, 77) == RF.ve_num_wins(n, 77)


def test_goldens_record_their_provenance():
    """Every fixture names the third-party versions whose arithmetic it holds (the reference pins transformers 5.2.0; this image has 5.15.0:
    SURVEY.md 8c asks for the skew to be recorded with the results)."""
    import glob
    import json
    import numpy as np
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))
    assert len(files) >= 12
    for f in files:
        with np.load(f) as z:
            assert "provenance" in z.files, f
            p = json.loads(str(z["provenance"]))
        assert p["transformers"] and p["torch"] and p["reference_pins"]["transformers"] == "5.2.0", (f, p)
