"""Prompt-analysis / voice-conversion front-end on the GPU (-m gpu) against (a) golden vectors of the UNMODIFIED reference
(tests/golden/frontend.npz: S3 log-mel, 24 kHz prompt mel, CAMPPlus body, voice-encoder LSTM body) and (b) the CPU oracle for the
pieces whose arithmetic lives in third-party packages absent from the reference tree (S3TokenizerV2.quantize, Kaldi fbank: the oracle
itself is an unpinned restatement there, so these tests prove HIP == restatement, not HIP == upstream).

Stated tolerances (fp32 path, framed DFT as an fp32 GEMM instead of an FFT):
  log-mel features: max-abs <= 2e-3 (units of log / log10-scaled mel; quiet bins sit 4 decades of amplitude below the peak, where an
  fp32 DFT -- FFT or GEMM -- has ~1e-3 relative error), mean-abs <= 5e-5;   x-vector / utterance embedding: max-abs <= 2e-4;
  S3 tokens: identical ids wherever the FSQ pre-rounding value is >= 2e-3 away from a rounding boundary.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "frontend.npz"))


def test_s3_log_mel_vs_reference(dev, gold):
    from chatterbox_amd import synth
    from chatterbox_amd.frontend import S3TokenizerEngine
    eng = S3TokenizerEngine({}, dev)
    mel = eng.log_mel(synth.prompt_wav(3.0, 16000)).cpu()
    ref = torch.from_numpy(gold["s3_logmel"]).t()
    assert mel.shape == ref.shape
    e = (mel - ref).abs()
    assert e.max() <= 2e-3 and e.mean() <= 5e-5, f"S3 log-mel max {e.max():.3e} mean {e.mean():.3e}"


def test_mel24k_vs_reference(dev, gold):
    from chatterbox_amd import synth
    from chatterbox_amd.frontend import Mel24kExtractor
    mel = Mel24kExtractor(dev)(synth.prompt_wav(3.0, 24000)).cpu()
    ref = torch.from_numpy(gold["mel24k"]).t()
    assert mel.shape == ref.shape
    e = (mel - ref).abs()
    assert e.max() <= 2e-3 and e.mean() <= 5e-5, f"24 kHz mel max {e.max():.3e} mean {e.mean():.3e}"


def test_campplus_vs_reference(dev, gold):
    from chatterbox_amd import synth
    from chatterbox_amd.frontend import CAMPPlusEngine
    from oracle import ref_frontend as RF
    sd = synth.campplus_state_dict(0)
    eng = CAMPPlusEngine(sd, dev)
    xv = eng.forward(torch.from_numpy(gold["fbank"]).to(dev)).cpu()
    e = (xv - torch.from_numpy(gold["xvector"])).abs().max().item()
    assert e <= 2e-4, f"x-vector max-abs {e:.3e} vs the reference's CAMPPlus"
    # fbank front (unpinned restatement): HIP == oracle
    w16 = synth.prompt_wav(3.0, 16000)
    fb = eng.fbank(w16).cpu()
    ofb = RF.kaldi_fbank(w16)
    ofb = ofb - ofb.mean(0, keepdim=True)
    e2 = (fb - ofb).abs()
    assert fb.shape == ofb.shape and e2.max() <= 5e-3 and e2.mean() <= 1e-4, f"fbank max {e2.max():.3e} mean {e2.mean():.3e}"
    # a ragged length that is not a multiple of the 100-frame pooling segment
    feats = torch.from_numpy(gold["fbank"])[:137]
    e3 = (eng.forward(feats.to(dev)).cpu() - RF.campplus_forward(synth.campplus_state_dict(0, prefix=""), feats[None])[0]).abs().max().item()
    assert e3 <= 2e-4, f"137-frame x-vector max-abs {e3:.3e}"


def test_voice_encoder_vs_reference(dev, gold):
    from chatterbox_amd import synth
    from chatterbox_amd.frontend import VoiceEncoderEngine
    eng = VoiceEncoderEngine(synth.voice_encoder_state_dict(0), dev)
    emb = eng.inference(torch.from_numpy(gold["ve_mel"]).to(dev)).cpu()
    e = (emb - torch.from_numpy(gold["ve_embed"])).abs().max().item()
    assert e <= 2e-4 and abs(float(emb.norm()) - 1.0) <= 1e-5, f"utterance embedding max-abs {e:.3e}"
    mel = eng.melspectrogram(synth.prompt_wav(3.0, 16000)).cpu()
    ref = torch.from_numpy(gold["ve_mel"])
    rel = ((mel - ref).abs() / (ref.abs() + 1e-3 * ref.abs().max())).max().item()
    assert mel.shape == ref.shape and rel <= 1e-3, f"voice-encoder power mel rel. err {rel:.3e}"


def test_s3tokenizer_quantize_vs_oracle(dev):
    """S3TokenizerV2.quantize (third-party, PARITY UNPINNED upstream): the HIP path against the CPU restatement on the same weights."""
    from chatterbox_amd import synth
    from chatterbox_amd.frontend import S3TokenizerEngine
    from oracle import ref_frontend as RF
    sd = synth.s3tokenizer_state_dict(0, n_layer=2)
    digits = lambda t: torch.stack([(t // 3 ** d) % 3 for d in range(8)], -1)
    RF.S3TOK["n_layer"] = 2
    try:
        w16 = synth.prompt_wav(2.0, 16000)
        eng = S3TokenizerEngine(sd, dev)
        mel = eng.log_mel(w16)
        ids = eng.quantize(mel).cpu()
        oid, hq = RF.s3tokenizer_quantize(sd, RF.s3_log_mel(w16))
        tok, n = eng(w16, max_len=20)                    # S3Tokenizer.forward(max_len): the mel is cut to 4 * max_len frames first
        otok = RF.s3_tokenize(sd, w16, max_len=20)
    finally:
        RF.S3TOK["n_layer"] = 6
    assert ids.shape == oid.shape == (mel.shape[0] // 4,) and int(ids.min()) >= 0 and int(ids.max()) < 6561
    # FSQ digits may differ only where the pre-rounding value sits on a rounding boundary (fp32 accumulation order)
    diff = digits(ids) != digits(oid)
    assert diff.float().mean() <= 0.02, f"{int(diff.sum())} of {diff.numel()} FSQ digits differ from the restatement"
    assert tok.shape == (1, 20) and int(n[0]) == 20
    assert (digits(tok[0].cpu()) != digits(otok)).float().mean() <= 0.02


def test_embed_ref_shapes_and_contract(dev):
    """S3Gen.embed_ref contract (s3gen.py:118-171): 2 mel frames per prompt token, x-vector (1, 192), tokens in range."""
    from chatterbox_amd import synth
    from chatterbox_amd.frontend import PromptAnalyzer
    sd = dict(synth.s3tokenizer_state_dict(0, n_layer=1), **synth.campplus_state_dict(0))
    pa = PromptAnalyzer(sd, synth.voice_encoder_state_dict(0), dev)
    w24 = synth.prompt_wav(4.0, 24000).numpy()
    ref = pa.embed_ref(w24, 24000)
    n = ref["prompt_token"].shape[1]
    assert ref["prompt_feat"].shape == (1, 2 * n, 80) and ref["embedding"].shape == (1, 192) and int(ref["prompt_token_len"][0]) == n
    assert n == 100 and ref["prompt_feat_len"] is None
    spk, ptoks = pa.t3_prompt(synth.prompt_wav(4.0, 16000).numpy(), 150)
    assert spk.shape == (1, 256) and ptoks.shape == (1, 100) and abs(float(spk.norm()) - 1.0) < 1e-3


def test_prepare_conditionals_and_vc_end_to_end(dev, tmp_path):
    """The reference's example flows on synthetic weights: `generate(audio_prompt_path=...)` (tts.py:208-272 -> prepare_conditionals)
    and `ChatterboxVC.generate(audio, target_voice_path=...)` (vc.py:83-104), from WAV files on disk to a 24 kHz waveform."""
    from scipy.io import wavfile
    from chatterbox_amd import synth
    from chatterbox_amd.api import ChatterboxMultilingualTTS, ChatterboxVC
    prompt, source = tmp_path / "prompt.wav", tmp_path / "source.wav"
    wavfile.write(prompt, 22050, (synth.prompt_wav(6.5, 22050).numpy() * 32767).astype(np.int16))  # an odd rate: exercises resampling
    wavfile.write(source, 16000, synth.prompt_wav(2.0, 16000, seed=3).numpy())
    m = ChatterboxMultilingualTTS.from_synthetic(dev, t3_layers=2, with_prompt_nets=True, tokenizer_layers=1)
    m.prepare_conditionals(str(prompt), exaggeration=0.7)
    c = m.conds
    assert c.t3.speaker_emb.shape == (1, 256) and c.t3.cond_prompt_speech_tokens.shape == (1, 150) and float(c.t3.emotion_adv) == pytest.approx(0.7)
    n = c.gen["prompt_token"].shape[1]
    # 6.5 s is not a whole number of 40 ms tokens: 325 mel frames, tokens trimmed to 325 // 2 = 162 (s3gen.py:152-158), the mel keeps
    # its odd frame (flow.py:170-195 then returns 2N - 1 frames)
    assert c.gen["prompt_feat"].shape == (1, 325, 80) and c.gen["embedding"].shape == (1, 192) and n == 162
    wav = m._generate(synth.text_tokens(10)[1:-1], drop_last_token=True, temperature=0.8, cfg_weight=0.5, repetition_penalty=1.2, min_p=0.05, top_p=1.0)
    assert wav.dim() == 2 and wav.shape[0] == 1 and torch.isfinite(wav).all()
    vc = ChatterboxVC.from_synthetic(dev, tokenizer_layers=1)
    out = vc.generate(str(source), target_voice_path=str(prompt))
    assert out.shape == (1, 50 * 960 - 480) and torch.isfinite(out).all() and out.abs().max() <= 0.99  # 2 s -> 50 tokens, 99 mel frames
    vc.ref_dict = synth.s3gen_ref()
    out2 = vc.generate(s3_tokens=synth.speech_tokens(30))
    assert out2.shape == (1, 30 * 960)
