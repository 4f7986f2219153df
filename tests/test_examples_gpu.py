"""The reference's acceptance surface (-m gpu): the four usage scenarios of its example programs (tests/example_scenarios.py: this repo's own
scripts over the `chatterbox` import paths; the CPU test test_reference_examples_use_only_the_api_we_export ties them to the reference's
programs, whose text is NOT stored in this repo) run against the alias package ("a user of the reference switches the
import and runs").  Rounds 3-5 executed the reference's programs themselves, unmodified, from a stored copy of their text (GPUTEST_r05: green).

What is synthetic: the checkpoints (no network).  `huggingface_hub.hf_hub_download` / `snapshot_download` are pointed at a directory
that holds a seeded random-init checkpoint IN THE REFERENCE'S FILE LAYOUT -- t3_cfg.safetensors, s3gen.safetensors (with its tokenizer.* and
speaker_encoder.* tensors), ve.safetensors, tokenizer.json, conds.pt; the multilingual s3gen.pt / ve.pt / grapheme_mtl_merged_expanded_v1.json;
the Turbo / Nano GPT-2 tokenizer files -- written with formats.write_safetensors / torch.save, so the real `from_pretrained` ->
`from_local` code paths, the tokenizers, conds.pt and the voice-prompt analysis all run.  `torchaudio` (absent from the image) is a
five-line shim over scipy.io.wavfile.  Reduced depth (2 transformer layers) keeps the run short; the architecture per layer is the real one.
"""
import json
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def _char_tokenizer(path, extra_tokens=(), vocab_size=None):
    """A character-level `tokenizers` JSON with the special tokens the reference's tokenizers look for."""
    from tokenizers import Tokenizer, models, pre_tokenizers
    chars = list("abcdefghijklmnopqrstuvwxyz0123456789.,!?'-:;\"()çàéèêëîïôöùûüß") + ["[SPACE]"]
    vocab = {}
    for t in ["[STOP]", "[UNK]", "[START]", "[PAD]", "[SEP]", "[CLS]", "[MASK]"] + list(extra_tokens) + chars:
        vocab.setdefault(t, len(vocab))
    tok = Tokenizer(models.WordLevel(vocab, unk_token="[UNK]"))
    tok.pre_tokenizer = pre_tokenizers.Split(pattern=tokenizers_regex(), behavior="isolated")
    tok.save(str(path))
    return len(vocab)


def tokenizers_regex():
    from tokenizers import Regex
    return Regex(r"\[[A-Za-z]+\]|.")


def _gpt2_tokenizer(d):
    """Byte-level BPE without merges (256 byte tokens + <|endoftext|>): loads through AutoTokenizer like the GPT-2 files of the Turbo repo."""
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers
    from tokenizers.pre_tokenizers import ByteLevel
    vocab = {c: i for i, c in enumerate(sorted(ByteLevel.alphabet()))}
    vocab["<|endoftext|>"] = len(vocab)
    tok = Tokenizer(models.BPE(vocab=vocab, merges=[]))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    tok.save(str(d / "tokenizer.json"))
    json.dump(dict(tokenizer_class="PreTrainedTokenizerFast", eos_token="<|endoftext|>", bos_token="<|endoftext|>", unk_token="<|endoftext|>",
                   model_max_length=1024), open(d / "tokenizer_config.json", "w"))


@pytest.fixture(scope="module")
def hub(tmp_path_factory, dev):
    """Synthetic checkpoint directories in the reference's file layout + the patched hub functions."""
    from chatterbox_amd import formats, synth
    from chatterbox_amd.api import Conditionals, T3Cond
    root = tmp_path_factory.mktemp("hub")
    small = dict(n_mid=1, n_enc=1, n_up_enc=1)
    prompt_nets = dict(synth.s3tokenizer_state_dict(0, n_layer=2), **synth.campplus_state_dict(0))
    ve = synth.voice_encoder_state_dict(0)

    def conds(plen, emotion=True):
        c = synth.t3_cond(prompt_len=plen)
        t3 = T3Cond(speaker_emb=c["speaker_emb"], cond_prompt_speech_tokens=c["cond_prompt_speech_tokens"], emotion_adv=c["emotion_adv"] if emotion else None)
        return Conditionals(t3, synth.s3gen_ref(n_prompt_tokens=60))

    en = root / "en"
    en.mkdir()
    formats.write_safetensors(synth.t3_state_dict(2, 0, text_vocab=704), en / "t3_cfg.safetensors")
    formats.write_safetensors(dict(synth.s3gen_state_dict(0, **small), **prompt_nets), en / "s3gen.safetensors")
    formats.write_safetensors(ve, en / "ve.safetensors")
    assert _char_tokenizer(en / "tokenizer.json") <= 704
    conds(150).save(en / "conds.pt")

    mtl = root / "mtl"
    mtl.mkdir()
    formats.write_safetensors(synth.t3_state_dict(2, 0), mtl / "t3_mtl23ls_v2.safetensors")
    torch.save(dict(synth.s3gen_state_dict(0, **small), **prompt_nets), mtl / "s3gen.pt")
    torch.save(ve, mtl / "ve.pt")
    from chatterbox_amd.api import SUPPORTED_LANGUAGES
    assert _char_tokenizer(mtl / "grapheme_mtl_merged_expanded_v1.json", extra_tokens=[f"[{k}]" for k in SUPPORTED_LANGUAGES]) <= 2454
    conds(150).save(mtl / "conds.pt")

    dirs = {"en": en, "mtl": mtl}
    for name, (layers, dm, fn) in {"turbo": (2, 1024, "t3_turbo_v1.safetensors"), "nano": (2, 768, "t3_nano_v1.safetensors")}.items():
        d = root / name
        d.mkdir()
        formats.write_safetensors(synth.t3_turbo_state_dict(layers, dm, 0, include_wte=True), d / fn)
        formats.write_safetensors(dict(synth.s3gen_state_dict(0, meanflow=True, **small), **prompt_nets), d / "s3gen_meanflow.safetensors")
        formats.write_safetensors(ve, d / "ve.safetensors")
        _gpt2_tokenizer(d)
        conds(375, emotion=False).save(d / "conds.pt")
        dirs[name] = d

    def hf_hub_download(repo_id=None, filename=None, **kw):
        d = mtl if filename in ("s3gen.pt", "ve.pt", "grapheme_mtl_merged_expanded_v1.json") or str(filename).startswith("t3_mtl") else en
        return str(d / filename)

    def snapshot_download(repo_id=None, **kw):
        if "nano" in repo_id:
            return str(dirs["nano"])
        return str(dirs["turbo"] if "turbo" in repo_id else mtl)

    return dirs, hf_hub_download, snapshot_download


def _run_example(name, tmp_path, monkeypatch, hub, max_tokens=40):
    """run one scenario of tests/example_scenarios.py in tmp_path with the hub and torchaudio shims in place."""
    import huggingface_hub
    from scipy.io import wavfile
    from chatterbox_amd import synth
    from chatterbox_amd.t3 import T3Engine
    from chatterbox_amd.t3_turbo import T3TurboEngine
    dirs, hf, snap = hub
    monkeypatch.setattr(huggingface_hub, "hf_hub_download", hf)
    monkeypatch.setattr(huggingface_hub, "snapshot_download", snap)
    saved = {}
    import importlib.machinery
    ta = types.ModuleType("torchaudio")
    ta.__spec__ = importlib.machinery.ModuleSpec("torchaudio", None)  # transformers probes optional packages with importlib.util.find_spec
    ta.__version__ = "0.0.0+shim"

    def save(path, wav, sr, **kw):
        saved[str(path)] = (wav.detach().cpu(), sr)
        wavfile.write(str(path), sr, wav.detach().cpu().numpy().T.astype(np.float32))

    ta.save = save
    monkeypatch.setitem(sys.modules, "torchaudio", ta)
    # a random-init model never samples EOS: bound the AR loop (the examples do not pass max_new_tokens)
    for cls in (T3Engine, T3TurboEngine):
        orig = cls.generate

        def capped(self, *a, _orig=orig, **k):
            for key in ("max_new_tokens", "max_gen_len"):
                if key in k:
                    k[key] = min(k[key], max_tokens)
            return _orig(self, *a, **k)

        monkeypatch.setattr(cls, "generate", capped)
    monkeypatch.chdir(tmp_path)
    wavfile.write(str(tmp_path / "YOUR_FILE.wav"), 24000, synth.prompt_wav(seconds=7.0, sr=24000).numpy().astype(np.float32))  # the placeholder path the scenarios look for
    import example_scenarios
    getattr(example_scenarios, name)()
    return saved


def _check(saved, expect):
    assert set(saved) == set(expect), (sorted(saved), sorted(expect))
    for k, (wav, sr) in saved.items():
        assert sr == 24000 and wav.dim() == 2 and wav.shape[0] == 1 and wav.shape[1] > 2400 and bool(torch.isfinite(wav).all()), (k, tuple(wav.shape))
        assert float(wav.abs().max()) > 0


def test_example_tts(tmp_path, monkeypatch, hub):
    """the `tts` scenario: ChatterboxTTS.from_pretrained + generate, ChatterboxMultilingualTTS (French), voice cloning from YOUR_FILE.wav."""
    _check(_run_example("tts", tmp_path, monkeypatch, hub), ["test-1.wav", "test-2.wav", "test-3.wav"])


def test_example_tts_turbo(tmp_path, monkeypatch, hub):
    _check(_run_example("tts_turbo", tmp_path, monkeypatch, hub), ["test-turbo.wav"])


def test_example_tts_nano(tmp_path, monkeypatch, hub):
    _check(_run_example("tts_nano", tmp_path, monkeypatch, hub), ["test-nano.wav"])


def test_example_vc(tmp_path, monkeypatch, hub):
    """the `vc` scenario: source audio -> S3 tokens on the device -> S3Gen with the target voice analysed from a WAV file."""
    _check(_run_example("vc", tmp_path, monkeypatch, hub), ["testvc.wav"])
