"""Acceptance scenarios of the drop-in boundary: what a user of the reference's four example programs does, written against the
`chatterbox` import paths -- class per task, `from_pretrained`, `generate`, `torchaudio.save(path, wav, model.sr)`.

These are THIS repo's scripts (tests/test_examples_gpu.py runs them on the MI355X).  The reference's own example programs are not stored here;
tests/test_host_logic.py::test_reference_examples_use_only_the_api_we_export reads them where the reference is present (the authoring
container) and checks that every import, call and keyword argument they use is one the alias package serves and one of the scenarios below makes.

`CALLS` lists, per scenario, the (class, method, keyword arguments) it exercises -- the contract the CPU test compares with the reference's programs."""

CALLS = {
    "tts": [("ChatterboxTTS", "from_pretrained", ("device",)), ("ChatterboxTTS", "generate", ()), ("ChatterboxTTS", "generate", ("audio_prompt_path",)),
            ("ChatterboxMultilingualTTS", "from_pretrained", ("device",)), ("ChatterboxMultilingualTTS", "generate", ("language_id",))],
    "tts_turbo": [("ChatterboxTurboTTS", "from_pretrained", ("device",)), ("ChatterboxTurboTTS", "generate", ())],
    "tts_nano": [("ChatterboxTurboTTS", "from_pretrained", ("device", "nano")), ("ChatterboxTurboTTS", "generate", ())],
    "vc": [("ChatterboxVC", "from_pretrained", ()), ("ChatterboxVC", "generate", ("audio", "target_voice_path"))],
}


def _device():
    import torch
    return "cuda" if torch.cuda.is_available() else "cpu"


def tts(prompt_wav="YOUR_FILE.wav"):
    """English and multilingual synthesis with the built-in voice, then voice cloning from a WAV file."""
    import os
    import torchaudio
    from chatterbox.mtl_tts import ChatterboxMultilingualTTS
    from chatterbox.tts import ChatterboxTTS
    dev = _device()
    english = ChatterboxTTS.from_pretrained(device=dev)
    sentence = "The quick brown fox jumps over the lazy dog, twice, and then takes a nap."
    torchaudio.save("test-1.wav", english.generate(sentence), english.sr)
    many = ChatterboxMultilingualTTS.from_pretrained(device=dev)
    phrase = "Bonjour tout le monde, ceci est un essai."
    torchaudio.save("test-2.wav", many.generate(phrase, language_id="fr"), many.sr)
    if os.path.exists(prompt_wav):
        torchaudio.save("test-3.wav", english.generate(phrase, audio_prompt_path=prompt_wav), english.sr)


def tts_turbo():
    import torchaudio
    from chatterbox.tts_turbo import ChatterboxTurboTTS
    m = ChatterboxTurboTTS.from_pretrained(device="cuda")
    torchaudio.save("test-turbo.wav", m.generate("Well [chuckle] that went better than expected, shall we try the next one?"), m.sr)


def tts_nano():
    import torchaudio
    from chatterbox.tts_turbo import ChatterboxTurboTTS
    m = ChatterboxTurboTTS.from_pretrained(device="cuda", nano=True)
    torchaudio.save("test-nano.wav", m.generate("Well [chuckle] that went better than expected, shall we try the next one?"), m.sr)


def vc(source_wav="YOUR_FILE.wav", target_wav="YOUR_FILE.wav"):
    import torchaudio
    from chatterbox.vc import ChatterboxVC
    m = ChatterboxVC.from_pretrained(_device())
    torchaudio.save("testvc.wav", m.generate(audio=source_wav, target_voice_path=target_wav), m.sr)
