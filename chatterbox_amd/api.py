"""Drop-in boundary: the reference's Python class API on top of the MI355X engines.

`ChatterboxTTS` (reference tts.py:106), `ChatterboxMultilingualTTS` (mtl_tts.py:155), `ChatterboxVC` (vc.py:16) keep
the reference names, signatures, defaults, return convention (CPU float tensor (1, n_samples) at `.sr` = 24 kHz) and
error behaviour (ValueError for an unknown language / model alias, assert when no voice is prepared).  Checkpoints
are read in the reference's own file and state-dict layout.  Out of scope this round (SURVEY.md 8f "next" rows):
voice-prompt analysis (`prepare_conditionals` needs the S3 tokenizer, CAMPPlus and the voice encoder), hence voices
come from `conds.pt` / `Conditionals`; the Turbo/Nano GPT-2 T3 backbone; the watermarker (third-party `perth`,
applied only if importable -- parity is defined on the pre-watermark waveform).
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
        torch.save(dict(t3=self.t3.__dict__, gen=self.gen), fpath)

    @classmethod
    def load(cls, fpath, map_location="cpu"):
        kw = torch.load(fpath, map_location=torch.device(map_location) if isinstance(map_location, str) else map_location,
                        weights_only=True)
        return cls(T3Cond(**kw["t3"]), kw["gen"])


def _load_state(path):
    path = str(path)
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path)
    else:
        sd = torch.load(path, map_location="cpu", weights_only=True)
    if "model" in sd and not torch.is_tensor(sd["model"]):  # `{"model": [state]}` wrapper (reference tts.py:146-147)
        sd = sd["model"][0]
    return sd


def _watermarker():
    try:
        import perth
        return perth.PerthImplicitWatermarker()
    except Exception:
        return None


class _Base:
    sr = S3GEN_SR

    def __init__(self, engine: ChatterboxEngine, tokenizer, device, conds: Optional[Conditionals] = None):
        self.engine, self.tokenizer, self.device, self.conds = engine, tokenizer, device, conds
        self.t3, self.s3gen, self.ve = engine.t3, engine, None
        self.watermarker = _watermarker()

    def prepare_conditionals(self, wav_fpath, exaggeration=0.5):
        raise NotImplementedError(
            "voice-prompt analysis (S3 tokenizer + CAMPPlus x-vector + voice encoder, reference tts.py:182-206) is a "
            "'next' row of this build (SURVEY.md 8f N1/N2): load a voice with Conditionals.load('conds.pt') instead")

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
        return cls(eng, None, device, conds)


class ChatterboxTTS(_Base):
    _TEXT_VOCAB = 704

    @classmethod
    def from_local(cls, ckpt_dir, device):
        d = Path(ckpt_dir)
        eng = ChatterboxEngine(_load_state(d / "t3_cfg.safetensors"), _load_state(d / "s3gen.safetensors"), device)
        conds = Conditionals.load(d / "conds.pt") if (d / "conds.pt").exists() else None
        return cls(eng, EnTokenizer(d / "tokenizer.json"), device, conds)

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
        eng = ChatterboxEngine(_load_state(d / t3_file), _load_state(d / "s3gen.pt"), device)
        conds = Conditionals.load(d / "conds.pt") if (d / "conds.pt").exists() else None
        return cls(eng, MTLTokenizer(d / "grapheme_mtl_merged_expanded_v1.json"), device, conds)

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

    def __init__(self, engine, tokenizer, device, conds=None, model_label="Turbo"):
        self.engine, self.tokenizer, self.device, self.conds, self.model_label = engine, tokenizer, device, conds, model_label
        self.t3, self.s3gen, self.ve = engine.t3, engine, None
        self.watermarker = _watermarker()

    @classmethod
    def from_local(cls, ckpt_dir, device, nano=False):
        d = Path(ckpt_dir)
        t3_sd = _load_state(d / ("t3_nano_v1.safetensors" if nano else "t3_turbo_v1.safetensors"))
        t3_sd.pop("tfmr.wte.weight", None)  # present in the file, unused (reference deletes it after loading, tts_turbo.py:167)
        eng = TurboEngine(t3_sd, _load_state(d / "s3gen_meanflow.safetensors"), device)
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(str(d))
        if tok.pad_token is None:
            tok.pad_token = tok.eos_token
        conds = Conditionals.load(d / "conds.pt") if (d / "conds.pt").exists() else None
        return cls(eng, tok, device, conds, "Nano" if nano else "Turbo")

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

    def prepare_conditionals(self, wav_fpath, exaggeration=0.0, norm_loudness=True):
        raise NotImplementedError("voice-prompt analysis is a 'next' row (SURVEY.md 8f N1/N2): load conds.pt instead")

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
    """Voice conversion = S3 tokens of the source audio -> S3Gen with the target voice (reference vc.py:83-104).
    The S3 tokenizer front-end is a 'next' row, so `generate` accepts the source as S3 tokens (config 5's parity
    contract starts at the token boundary)."""
    sr = S3GEN_SR

    def __init__(self, engine, device, ref_dict=None):
        self.engine, self.device, self.ref_dict = engine, device, ref_dict
        self.s3gen = engine
        self.watermarker = _watermarker()

    @classmethod
    def from_local(cls, ckpt_dir, device):
        d = Path(ckpt_dir)
        s3 = _load_state(d / "s3gen.safetensors")
        eng = ChatterboxEngine.__new__(ChatterboxEngine)
        from .hift import HiFTEngine
        from .s3gen import FlowEngine
        eng.dev, eng.t3, eng.flow, eng.hift, eng.last_timing = torch.device(device), None, FlowEngine(s3, device), HiFTEngine(s3, device), {}
        ref = Conditionals.load(d / "conds.pt").gen if (d / "conds.pt").exists() else None
        return cls(eng, device, ref)

    def set_target_voice(self, wav_fpath):
        raise NotImplementedError("reference-voice analysis is a 'next' row (SURVEY.md 8f N2): pass a prepared ref_dict")

    def generate(self, audio=None, target_voice_path=None, s3_tokens=None):
        if target_voice_path:
            self.set_target_voice(target_voice_path)
        assert self.ref_dict is not None, "Please `prepare_conditionals` first or specify `target_voice_path`"
        if s3_tokens is None:
            raise NotImplementedError("waveform -> S3 tokens (S3TokenizerV2) is a 'next' row (SURVEY.md 8f N1): pass s3_tokens=")
        wavs, _ = self.engine.vocode([torch.as_tensor(s3_tokens).view(-1).long()], self.ref_dict)
        wav = wavs[0].detach().float().cpu()
        if self.watermarker is not None:
            wav = torch.from_numpy(self.watermarker.apply_watermark(wav.numpy(), sample_rate=self.sr))
        return wav.unsqueeze(0)
