"""Drop-in boundary: the reference's Python class API on top of the MI355X engines.

`ChatterboxTTS` (reference tts.py:106), `ChatterboxMultilingualTTS` (mtl_tts.py:155), `ChatterboxVC` (vc.py:16) keep
the reference names, signatures, defaults, return convention (CPU float tensor (1, n_samples) at `.sr` = 24 kHz) and
error behaviour (ValueError for an unknown language / model alias, assert when no voice is prepared).  Checkpoints
are read in the reference's own file and state-dict layout.  `ChatterboxTurboTTS` (tts_turbo.py:111) covers the Turbo / Nano GPT-2
backbone.  Voice-prompt analysis (`prepare_conditionals`, `ChatterboxVC.set_target_voice`, `ChatterboxVC.generate(audio)`) runs on
the device through chatterbox_amd/frontend.py (S3 tokenizer, CAMPPlus, voice encoder, 24 kHz mel); it needs the `tokenizer.*` /
`speaker_encoder.*` tensors of the S3Gen checkpoint and `ve.safetensors`, and fails with a clear message when a checkpoint lacks
them.  The watermarker (third-party `perth`) is applied only if importable -- parity is defined on the pre-watermark waveform;
Turbo's `norm_loudness` needs `pyloudnorm` and is skipped with a warning when that is missing (as the reference does on error).
"""
import os
from dataclasses import dataclass
from pathlib import Path
from typing import Optional

import torch

from . import synth
from .engine import ChatterboxEngine, TurboEngine
from .text import EnTokenizer, MTLTokenizer, punc_norm, punc_norm_en, punc_norm_turbo

S3GEN_SR, S3_SR = 24000, 16000
REPO_ID = "ResembleAI/chatterbox"
DEFAULT_MULTILINGUAL_T3_MODEL = "t3_mtl23ls_v2.safetensors"
MULTILINGUAL_T3_MODELS = {"v2": "t3_mtl23ls_v2.safetensors", "t3_mtl23ls_v2": "t3_mtl23ls_v2.safetensors",
                          "v3": "t3_mtl23ls_v3.safetensors", "t3_mtl23ls_v3": "t3_mtl23ls_v3.safetensors"}
SUPPORTED_LANGUAGES = {
    "ar": "Arabic", "da": "Danish", "de": "German", "el": "Greek", "en": "English", "es": "Spanish", "fi": "Finnish",
    "fr": "French", "he": "Hebrew", "hi": "Hindi", "it": "Italian", "ja": "Japanese", "ko": "Korean", "ms": "Malay",
    "nl": "Dutch", "no": "Norwegian", "pl": "Polish", "pt": "Portuguese", "ru": "Russian", "sv": "Swedish",
    "sw": "Swahili", "tr": "Turkish", "zh": "Chinese"}


def _resolve_multilingual_t3_model(t3_model):
    if t3_model is None:
        return DEFAULT_MULTILINGUAL_T3_MODEL
    if t3_model in MULTILINGUAL_T3_MODELS:
        return MULTILINGUAL_T3_MODELS[t3_model]
    if t3_model.endswith(".safetensors"):
        return t3_model
    raise ValueError(f"Unknown multilingual T3 model '{t3_model}'. Expected one of {sorted(MULTILINGUAL_T3_MODELS)} "
                     f"or a .safetensors filename.")


@dataclass
class T3Cond:
    """Same fields as the reference dataclass (models/t3/modules/cond_enc.py:12-22)."""
    speaker_emb: torch.Tensor
    clap_emb: Optional[torch.Tensor] = None
    cond_prompt_speech_tokens: Optional[torch.Tensor] = None
    cond_prompt_speech_emb: Optional[torch.Tensor] = None
    emotion_adv: Optional[torch.Tensor] = 0.5

    def to(self, *, device=None, dtype=None):
        for k, v in self.__dict__.items():
            if torch.is_tensor(v):
                setattr(self, k, v.to(device=device, dtype=dtype if v.is_floating_point() else None))
        return self

    def as_dict(self):
        return dict(speaker_emb=self.speaker_emb, cond_prompt_speech_tokens=self.cond_prompt_speech_tokens,
                    emotion_adv=self.emotion_adv)


@dataclass
class Conditionals:
    """`conds.pt` = torch.save(dict(t3=T3Cond.__dict__, gen=dict)) (reference tts.py:64-103); this is also exactly the
    payload that dist.broadcast_conditionals ships to the other ranks."""
    t3: T3Cond
    gen: dict

    def to(self, device):
        self.t3 = self.t3.to(device=device)
        self.gen = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self.gen.items()}
        return self

    def save(self, fpath):
        if str(fpath).endswith(".safetensors"):  # pickle-free container (formats.py); conds.pt stays the reference's torch pickle
            from . import formats
            return formats.save_conds(self, fpath)
        torch.save(dict(t3=self.t3.__dict__, gen=self.gen), fpath)

    @classmethod
    def load(cls, fpath, map_location="cpu"):
        if str(fpath).endswith(".safetensors"):
            from . import formats
            return formats.load_conds(fpath, map_location)
        kw = torch.load(fpath, map_location=torch.device(map_location) if isinstance(map_location, str) else map_location,
                        weights_only=True)
        return cls(T3Cond(**kw["t3"]), kw["gen"])


def _load_state(path):
    path = str(path)
    if path.endswith(".safetensors"):
        from .formats import read_safetensors
        sd = read_safetensors(path)  # zero-copy views of one memory map (no host copy before the H2D transfer)
    else:
        sd = torch.load(path, map_location="cpu", weights_only=True)
    if "model" in sd and not torch.is_tensor(sd["model"]):  # `{"model": [state]}` wrapper (reference tts.py:146-147)
        sd = sd["model"][0]
    return sd


def _make_analyzer(s3gen_sd, ve_sd, device):
    from .frontend import PromptAnalyzer
    return PromptAnalyzer(s3gen_sd, ve_sd, device)


def _norm_loudness(wav, sr, target_lufs=-27.0):
    """Gain a waveform to `target_lufs` integrated loudness (reference ChatterboxTurboTTS.norm_loudness, tts_turbo.py:223-239): needs the
    third-party `pyloudnorm`; on any error the reference prints a warning and carries on with the input, and so does this."""
    try:
        import numpy as np
        import pyloudnorm as ln
        gain = 10.0 ** ((target_lufs - ln.Meter(sr).integrated_loudness(wav)) / 20.0)
        if np.isfinite(gain) and gain > 0.0:
            wav = wav * gain
    except Exception as e:
        print(f"Warning: Error in norm_loudness, skipping: {e}")
    return wav


def _prepare_conditionals(analyzer, wav, exaggeration, prompt_len, device, min_seconds=None, norm_loudness=False, enc_cond_len=None):
    """The common body of prepare_conditionals (tts.py:182-206, mtl_tts.py:253-277, tts_turbo.py:241-270).  `wav`: a file path or a
    (waveform, sample_rate) pair."""
    from . import frontend as fe
    if analyzer is None or not analyzer.tokenizer.available or not analyzer.speaker_encoder.available or analyzer.ve is None:
        raise RuntimeError("voice-prompt analysis needs the `tokenizer.*` and `speaker_encoder.*` tensors of the S3Gen checkpoint and "
                           "ve.safetensors; this model was built without them -- load a prepared voice with Conditionals.load('conds.pt')")
    if isinstance(wav, (tuple, list)):
        w24 = fe.resample(wav[0], wav[1], S3GEN_SR)
    else:
        w24, _ = fe.load_wav(wav, S3GEN_SR)
    if min_seconds is not None:
        assert len(w24) / S3GEN_SR > min_seconds, "Audio prompt must be longer than 5 seconds!"
    if norm_loudness:
        w24 = _norm_loudness(w24, S3GEN_SR)
    w16 = fe.resample(w24, S3GEN_SR, S3_SR)
    gen = analyzer.embed_ref(w24[: analyzer.DEC_COND_LEN], S3GEN_SR)
    spk, ptoks = analyzer.t3_prompt(w16, prompt_len, enc_cond_len=enc_cond_len)
    t3 = T3Cond(speaker_emb=spk, cond_prompt_speech_tokens=ptoks, emotion_adv=exaggeration * torch.ones(1, 1, 1)).to(device=device)
    return Conditionals(t3, {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in gen.items()})


def _watermarker():
    try:
        import perth
        return perth.PerthImplicitWatermarker()
    except Exception:
        return None


class _Base:
    sr = S3GEN_SR

    ENC_COND_LEN, DEC_COND_LEN = 6 * S3_SR, 10 * S3GEN_SR
    PROMPT_LEN = 150  # T3Config.speech_cond_prompt_len

    def __init__(self, engine: ChatterboxEngine, tokenizer, device, conds: Optional[Conditionals] = None, analyzer=None):
        self.engine, self.tokenizer, self.device, self.conds = engine, tokenizer, device, conds
        self.analyzer = analyzer  # frontend.PromptAnalyzer (S3 tokenizer + CAMPPlus + voice encoder + 24 kHz mel) or None
        self.t3, self.s3gen, self.ve = engine.t3, engine, (analyzer.ve if analyzer is not None else None)
        self.watermarker = _watermarker()

    def prepare_conditionals(self, wav_fpath, exaggeration=0.5):
        """reference tts.py:182-206 / mtl_tts.py:253-277: waveform file -> self.conds."""
        self.conds = _prepare_conditionals(self.analyzer, wav_fpath, exaggeration, self.PROMPT_LEN, self.device)

    def _finish(self, wav):
        wav = wav.detach().float().cpu()
        if self.watermarker is not None:
            wav = torch.from_numpy(self.watermarker.apply_watermark(wav.numpy(), sample_rate=self.sr))
        return wav.unsqueeze(0)

    def _set_exaggeration(self, exaggeration):
        cur = float(torch.as_tensor(self.conds.t3.emotion_adv).reshape(-1)[0])
        if float(exaggeration) != cur:
            c = self.conds.t3
            self.conds.t3 = T3Cond(speaker_emb=c.speaker_emb, cond_prompt_speech_tokens=c.cond_prompt_speech_tokens,
                                   emotion_adv=exaggeration * torch.ones(1, 1, 1))

    def _generate(self, text_tokens, drop_last_token, **samp):
        sot, eot = 255, 0
        tt = torch.cat([torch.tensor([sot]), text_tokens.view(-1).long().cpu(), torch.tensor([eot])])
        wavs, _ = self.engine.synthesize([tt], self.conds.t3.as_dict(), self.conds.gen, max_new_tokens=1000,
                                         drop_last_token=drop_last_token, **samp)
        return self._finish(wavs[0])

    @classmethod
    def from_synthetic(cls, device="cuda", seed=0, t3_layers=30, **kw):
        """Seeded random-init model in the reference checkpoint layout + a synthetic voice (no network / no checkpoints)."""
        eng = ChatterboxEngine(synth.t3_state_dict(t3_layers, seed, text_vocab=cls._TEXT_VOCAB), synth.s3gen_state_dict(seed),
                               device, n_t3_layers=t3_layers)
        c = synth.t3_cond()
        conds = Conditionals(T3Cond(**c), synth.s3gen_ref())
        analyzer = None
        if kw.get("with_prompt_nets"):  # synthetic S3 tokenizer / CAMPPlus / voice encoder so that prepare_conditionals runs end to end
            nl = kw.get("tokenizer_layers", 6)
            analyzer = _make_analyzer(dict(synth.s3tokenizer_state_dict(seed, n_layer=nl), **synth.campplus_state_dict(seed)),
                                      synth.voice_encoder_state_dict(seed), device)
        return cls(eng, None, device, conds, analyzer)


class ChatterboxTTS(_Base):
    _TEXT_VOCAB = 704

    @classmethod
    def from_local(cls, ckpt_dir, device):
        d = Path(ckpt_dir)
        s3 = _load_state(d / "s3gen.safetensors")
        eng = ChatterboxEngine(_load_state(d / "t3_cfg.safetensors"), s3, device)
        conds = Conditionals.load(d / "conds.pt") if (d / "conds.pt").exists() else None
        ve = _load_state(d / "ve.safetensors") if (d / "ve.safetensors").exists() else None
        return cls(eng, EnTokenizer(d / "tokenizer.json"), device, conds, _make_analyzer(s3, ve, device))

    @classmethod
    def from_pretrained(cls, device):
        from huggingface_hub import hf_hub_download
        for f in ("ve.safetensors", "t3_cfg.safetensors", "s3gen.safetensors", "tokenizer.json", "conds.pt"):
            local = hf_hub_download(repo_id=REPO_ID, filename=f)
        return cls.from_local(Path(local).parent, device)

    def generate(self, text, repetition_penalty=1.2, min_p=0.05, top_p=1.0, audio_prompt_path=None, exaggeration=0.5,
                 cfg_weight=0.5, temperature=0.8):
        if audio_prompt_path:
            self.prepare_conditionals(audio_prompt_path, exaggeration=exaggeration)
        else:
            assert self.conds is not None, "Please `prepare_conditionals` first or specify `audio_prompt_path`"
        self._set_exaggeration(exaggeration)
        toks = self.tokenizer.text_to_tokens(punc_norm_en(text))
        return self._generate(toks, drop_last_token=False, temperature=temperature, cfg_weight=cfg_weight,
                              repetition_penalty=repetition_penalty, min_p=min_p, top_p=top_p)


class ChatterboxMultilingualTTS(_Base):
    _TEXT_VOCAB = 2454

    @classmethod
    def get_supported_languages(cls):
        return SUPPORTED_LANGUAGES.copy()

    @classmethod
    def from_local(cls, ckpt_dir, device, t3_model=None):
        d = Path(ckpt_dir)
        t3_file = _resolve_multilingual_t3_model(t3_model)
        s3 = _load_state(d / "s3gen.pt")
        eng = ChatterboxEngine(_load_state(d / t3_file), s3, device)
        conds = Conditionals.load(d / "conds.pt") if (d / "conds.pt").exists() else None
        ve = _load_state(d / "ve.pt") if (d / "ve.pt").exists() else None
        return cls(eng, MTLTokenizer(d / "grapheme_mtl_merged_expanded_v1.json"), device, conds, _make_analyzer(s3, ve, device))

    @classmethod
    def from_pretrained(cls, device, t3_model=None):
        from huggingface_hub import snapshot_download
        t3_file = _resolve_multilingual_t3_model(t3_model)
        d = snapshot_download(repo_id=REPO_ID, repo_type="model", revision="main", token=os.getenv("HF_TOKEN"),
                              allow_patterns=["ve.pt", t3_file, "s3gen.pt", "grapheme_mtl_merged_expanded_v1.json", "conds.pt",
                                              "Cangjie5_TC.json"])
        return cls.from_local(d, device, t3_model=t3_model)

    def generate(self, text, language_id, audio_prompt_path=None, exaggeration=0.5, cfg_weight=0.5, temperature=0.8,
                 repetition_penalty=1.2, min_p=0.05, top_p=1.0):
        if language_id and language_id.lower() not in SUPPORTED_LANGUAGES:
            raise ValueError(f"Unsupported language_id '{language_id}'. Supported languages: {', '.join(SUPPORTED_LANGUAGES)}")
        if audio_prompt_path:
            self.prepare_conditionals(audio_prompt_path, exaggeration=exaggeration)
        else:
            assert self.conds is not None, "Please `prepare_conditionals` first or specify `audio_prompt_path`"
        self._set_exaggeration(exaggeration)
        toks = self.tokenizer.text_to_tokens(punc_norm(text), language_id=language_id.lower() if language_id else None)
        return self._generate(toks, drop_last_token=True, temperature=temperature, cfg_weight=cfg_weight,
                              repetition_penalty=repetition_penalty, min_p=min_p, top_p=top_p)


class ChatterboxTurboTTS:
    """Reference tts_turbo.py:111-320: GPT2-medium (Turbo) or GPT2-small (Nano) T3, meanflow S3Gen, GPT-2 BPE tokenizer."""
    sr = S3GEN_SR
    ENC_COND_LEN, DEC_COND_LEN = 15 * S3_SR, 10 * S3GEN_SR  # tts_turbo.py:112-113: the T3 prompt covers up to 15 s (375 tokens)

    def __init__(self, engine, tokenizer, device, conds=None, model_label="Turbo", analyzer=None):
        self.engine, self.tokenizer, self.device, self.conds, self.model_label = engine, tokenizer, device, conds, model_label
        self.analyzer = analyzer
        self.t3, self.s3gen, self.ve = engine.t3, engine, (analyzer.ve if analyzer is not None else None)
        self.watermarker = _watermarker()

    @classmethod
    def from_local(cls, ckpt_dir, device, nano=False):
        d = Path(ckpt_dir)
        t3_sd = _load_state(d / ("t3_nano_v1.safetensors" if nano else "t3_turbo_v1.safetensors"))
        t3_sd.pop("tfmr.wte.weight", None)  # present in the file, unused (reference deletes it after loading, tts_turbo.py:167)
        s3 = _load_state(d / "s3gen_meanflow.safetensors")
        eng = TurboEngine(t3_sd, s3, device)
        ve = _load_state(d / "ve.safetensors") if (d / "ve.safetensors").exists() else None
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(str(d))
        if tok.pad_token is None:
            tok.pad_token = tok.eos_token
        conds = Conditionals.load(d / "conds.pt") if (d / "conds.pt").exists() else None
        return cls(eng, tok, device, conds, "Nano" if nano else "Turbo", _make_analyzer(s3, ve, device))

    @classmethod
    def from_pretrained(cls, device, nano=False):
        from huggingface_hub import snapshot_download
        d = snapshot_download(repo_id="ResembleAI/chatterbox-nano" if nano else "ResembleAI/chatterbox-turbo", token=os.getenv("HF_TOKEN"),
                              allow_patterns=["*.safetensors", "*.json", "*.txt", "*.pt", "*.model"])
        return cls.from_local(d, device, nano=nano)

    @classmethod
    def from_synthetic(cls, device="cuda", seed=0, nano=False, t3_layers=None):
        dmodel, layers = (768, 12) if nano else (1024, 24)
        layers = t3_layers or layers
        eng = TurboEngine(synth.t3_turbo_state_dict(layers, dmodel, seed), synth.s3gen_state_dict(seed, meanflow=True), device,
                          n_t3_layers=layers)
        c = synth.t3_cond(prompt_len=375)
        return cls(eng, None, device, Conditionals(T3Cond(speaker_emb=c["speaker_emb"], cond_prompt_speech_tokens=c["cond_prompt_speech_tokens"],
                                                          emotion_adv=None), synth.s3gen_ref()), "Nano" if nano else "Turbo")

    def norm_loudness(self, wav, sr, target_lufs=-27):
        return _norm_loudness(wav, sr, target_lufs)

    def prepare_conditionals(self, wav_fpath, exaggeration=0.0, norm_loudness=True):
        """reference tts_turbo.py:241-270 (prompt > 5 s, optional loudness normalisation to -27 LUFS, 375 prompt tokens)."""
        self.conds = _prepare_conditionals(self.analyzer, wav_fpath, exaggeration, 375, self.device, min_seconds=5.0,
                                           norm_loudness=norm_loudness, enc_cond_len=self.ENC_COND_LEN)

    def _generate(self, text_tokens, **samp):
        wavs, _ = self.engine.synthesize([text_tokens.view(-1).long().cpu()], self.conds.t3.as_dict(), self.conds.gen, **samp)
        wav = wavs[0].detach().float().cpu()
        if self.watermarker is not None:
            wav = torch.from_numpy(self.watermarker.apply_watermark(wav.numpy(), sample_rate=self.sr))
        return wav.unsqueeze(0)

    def generate(self, text, repetition_penalty=1.2, min_p=0.00, top_p=0.95, audio_prompt_path=None, exaggeration=0.0, cfg_weight=0.0,
                 temperature=0.8, top_k=1000, norm_loudness=True):
        if audio_prompt_path:
            self.prepare_conditionals(audio_prompt_path, exaggeration=exaggeration, norm_loudness=norm_loudness)
        else:
            assert self.conds is not None, "Please `prepare_conditionals` first or specify `audio_prompt_path`"
        if cfg_weight > 0.0 or exaggeration > 0.0 or min_p > 0.0:
            import logging
            logging.getLogger(__name__).warning(f"CFG, min_p and exaggeration are not supported by the {self.model_label} version and will be ignored.")
        ids = self.tokenizer(punc_norm_turbo(text), return_tensors="pt", padding=True, truncation=True).input_ids
        return self._generate(ids[0], temperature=temperature, top_k=top_k, top_p=top_p, repetition_penalty=repetition_penalty)


class ChatterboxVC:
    """Voice conversion (reference vc.py:16-104): S3 tokens of the source audio (S3 tokenizer on the device) -> S3Gen with the target
    voice -> HiFT.  `generate` also accepts the source as S3 tokens (`s3_tokens=`): the parity contract of config 5 starts at the token
    boundary because the tokenizer's arithmetic is third-party."""
    sr = S3GEN_SR
    ENC_COND_LEN, DEC_COND_LEN = 6 * S3_SR, 10 * S3GEN_SR

    def __init__(self, engine, device, ref_dict=None, analyzer=None):
        self.engine, self.device, self.ref_dict, self.analyzer = engine, device, ref_dict, analyzer
        self.s3gen = engine
        self.watermarker = _watermarker()

    @staticmethod
    def _engine(s3, device):
        from .hift import HiFTEngine
        from .s3gen import FlowEngine
        eng = ChatterboxEngine.__new__(ChatterboxEngine)
        eng.dev, eng.t3, eng.flow, eng.hift, eng.last_timing = torch.device(device), None, FlowEngine(s3, device), HiFTEngine(s3, device), {}
        return eng

    @classmethod
    def from_local(cls, ckpt_dir, device):
        d = Path(ckpt_dir)
        s3 = _load_state(d / "s3gen.safetensors")
        ref = Conditionals.load(d / "conds.pt").gen if (d / "conds.pt").exists() else None
        return cls(cls._engine(s3, device), device, ref, _make_analyzer(s3, None, device))

    @classmethod
    def from_pretrained(cls, device):
        """reference vc.py:61-74"""
        from huggingface_hub import hf_hub_download
        for f in ("s3gen.safetensors", "conds.pt"):
            local = hf_hub_download(repo_id=REPO_ID, filename=f)
        return cls.from_local(Path(local).parent, device)

    @classmethod
    def from_synthetic(cls, device="cuda", seed=0, tokenizer_layers=6):
        s3 = dict(synth.s3gen_state_dict(seed), **synth.s3tokenizer_state_dict(seed, n_layer=tokenizer_layers), **synth.campplus_state_dict(seed))
        return cls(cls._engine(s3, device), device, synth.s3gen_ref(), _make_analyzer(s3, None, device))

    def _need_analyzer(self):
        if self.analyzer is None or not self.analyzer.tokenizer.available or not self.analyzer.speaker_encoder.available:
            raise RuntimeError("this S3Gen checkpoint carries no `tokenizer.*` / `speaker_encoder.*` tensors: pass s3_tokens= and a prepared "
                               "ref_dict instead of waveforms")

    def set_target_voice(self, wav_fpath):
        """reference vc.py:76-81"""
        from . import frontend as fe
        self._need_analyzer()
        w24 = fe.resample(wav_fpath[0], wav_fpath[1], S3GEN_SR) if isinstance(wav_fpath, (tuple, list)) else fe.load_wav(wav_fpath, S3GEN_SR)[0]
        self.ref_dict = self.analyzer.embed_ref(w24[: self.DEC_COND_LEN], S3GEN_SR)

    def generate(self, audio=None, target_voice_path=None, s3_tokens=None):
        """reference vc.py:83-104.  audio: a WAV path or a (waveform, sample_rate) pair."""
        if target_voice_path:
            self.set_target_voice(target_voice_path)
        else:
            assert self.ref_dict is not None, "Please `prepare_conditionals` first or specify `target_voice_path`"
        if s3_tokens is None:
            from . import frontend as fe
            self._need_analyzer()
            w16 = fe.resample(audio[0], audio[1], S3_SR) if isinstance(audio, (tuple, list)) else fe.load_wav(audio, S3_SR)[0]
            s3_tokens, _ = self.analyzer.tokenizer(torch.from_numpy(w16))
        wavs, _ = self.engine.vocode([torch.as_tensor(s3_tokens).view(-1).long().cpu()], self.ref_dict)
        wav = wavs[0].detach().float().cpu()
        if self.watermarker is not None:
            wav = torch.from_numpy(self.watermarker.apply_watermark(wav.numpy(), sample_rate=self.sr))
        return wav.unsqueeze(0)
