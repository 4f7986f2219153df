"""Chunked ("streaming") synthesis (-m gpu), SURVEY.md 8f N3.  The reference is non-streaming and its `finalize=False` hook raises,
so the chunk schedule is this build's own; its oracle is the SAME schedule restated here on the CPU oracle's stage functions
(O.flow_inference(hold_back=...), O.hift_inference(cache_source=...)), which are themselves pinned against the reference for the
non-chunked case.  Checked: (1) streamed audio == oracle-streamed audio, (2) the pieces concatenate to the full length, (3) away from
the chunk seams the last round reproduces the non-streaming synthesize() output, (4) first audio arrives after the first chunk."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
SAMP = dict(temperature=0.8, cfg_weight=0.5, repetition_penalty=1.2, min_p=0.05, top_p=1.0)


def _oracle_stream(O, s3_sd, tokens, ref, z, phase, noise, first, chunk, lookahead, fade, n_steps, growth=1.0):
    """The schedule of engine.synthesize_stream for ONE utterance that never samples EOS (fixed-length synthetic run)."""
    N, P = tokens.numel(), ref["prompt_token"].shape[1]
    pieces, emitted, tail, cache, n = [], 0, None, None, min(N, first + lookahead)
    ramp = torch.linspace(0.0, 1.0, fade + 2)[1:-1]
    while True:
        final = n >= N
        hold = 0 if final else 2 * lookahead
        mel = O.flow_inference(s3_sd, tokens[None, :n], torch.tensor([n]), ref, z[:, :, : 2 * (P + n)], n_steps, hold_back=torch.tensor([hold]))
        frames = 2 * n - hold
        wav, src = O.hift_inference(s3_sd, mel[:, :, :frames], phase, noise[:, :, : 480 * frames], cache_source=cache)
        wav = O.trim_fade(wav)[0]
        cache = src[:, :, : 480 * frames]
        avail = min(480 * frames, max(1, n - 1) * 960) if final else 480 * frames
        end = avail if final else max(emitted, avail - fade)
        new = wav[emitted:end].clone()
        if tail is not None and new.numel():
            k = min(tail.numel(), new.numel())
            new[:k] = tail[:k] * (1 - ramp[:k]) + new[:k] * ramp[:k]
        tail = None if final else wav[end: min(avail, end + fade)].clone()
        emitted = end
        pieces.append(new)
        if final:
            return pieces
        n = min(N, n + max(1, int(round(chunk))))
        chunk = chunk * growth


def _setup(dev, N, P):
    from chatterbox_amd import synth
    from chatterbox_amd.engine import ChatterboxEngine
    L = 2
    t3_sd, s3_sd = synth.t3_state_dict(L, 0), synth.s3gen_state_dict(0, n_mid=2, n_enc=1, n_up_enc=1)
    eng = ChatterboxEngine(t3_sd, s3_sd, dev, n_t3_layers=L)
    texts = [synth.text_tokens(10, seed=1), synth.text_tokens(17, seed=2)]
    cond, ref = synth.t3_cond(), synth.s3gen_ref(n_prompt_tokens=P)
    z = synth.randn((2, 80, 2 * (P + N)), seed=5)
    phase = (synth.rand((2, 9, 1), seed=6) * 2 - 1) * math.pi
    phase[:, 0] = 0
    noise = synth.randn((2, 9, 960 * N), seed=6)
    kw = dict(max_new_tokens=N, uniforms=synth.rand((2, N), seed=3), ban_eos=True, ban_from=6561, z=z.transpose(1, 2).contiguous(), phase=phase,
              noise=noise, n_cfm_timesteps=3, **SAMP)
    return eng, s3_sd, texts, cond, ref, z, phase, noise, kw


@pytest.mark.parametrize("overlap", [True, False], ids=["overlapped", "serial"])
def test_stream_matches_oracle_schedule(dev, overlap):
    """overlap=True (the default since round 6): T3 keeps decoding on its own stream while a round's flow + vocoder run; a round still sees exactly the
    tokens of the schedule, so the same oracle pins both forms."""
    from oracle import ref_torch as O
    N, P, first, chunk, look, fade = 20, 8, 6, 7, 3, 240
    eng, s3_sd, texts, cond, ref, z, phase, noise, kw = _setup(dev, N, P)
    rounds = list(eng.synthesize_stream(texts, cond, ref, first_chunk=first, chunk=chunk, lookahead=look, fade=fade, overlap=overlap, **kw))
    assert len(rounds) == 3 and rounds[0]["n_tokens"] == [9, 9] and rounds[-1]["final"] == [True, True]  # 9 -> 16 -> 20 tokens
    first_len = 480 * (2 * (first + look) - 2 * look) - fade
    assert [w.numel() for w in rounds[0]["wavs"]] == [first_len, first_len]          # first audio: 6 tokens' worth minus the held-back tail
    full, toks = eng.synthesize(texts, cond, ref, drop_last_token=True, **kw)
    for b in range(2):
        streamed = torch.cat([r["wavs"][b] for r in rounds])
        assert streamed.numel() == (N - 1) * 960 == full[b].numel()
        pieces = _oracle_stream(O, s3_sd, toks[b], ref, z[b:b + 1], phase[b:b + 1], noise[b:b + 1], first, chunk, look, fade, 3)
        assert [p.numel() for p in pieces] == [r["wavs"][b].numel() for r in rounds]
        rmse = (streamed - torch.cat(pieces)).pow(2).mean().sqrt().item()
        assert rmse <= 2e-3, f"utt {b}: streamed vs oracle-streamed RMSE {rmse:.3e}"
        # seams are continuous: no sample-to-sample jump beyond what the one-shot waveform itself contains (x2)
        jump = (streamed[1:] - streamed[:-1]).abs()
        assert jump.max() <= 2.0 * (full[b].cpu()[1:] - full[b].cpu()[:-1]).abs().max() + 1e-3


def test_stream_last_round_is_the_full_synthesis(dev):
    """The last round re-synthesises everything with the same noise; its excitation differs from the one-shot run only where the cached
    source was substituted, so beyond the vocoder's receptive field (< 8000 samples, measured on the oracle) the waveforms coincide."""
    N, P = 30, 8
    eng, s3_sd, texts, cond, ref, z, phase, noise, kw = _setup(dev, N, P)
    rounds = list(eng.synthesize_stream(texts, cond, ref, first_chunk=6, chunk=40, lookahead=3, fade=240, **kw))
    assert len(rounds) == 2
    full, _ = eng.synthesize(texts, cond, ref, drop_last_token=True, **kw)
    for b in range(2):
        streamed = torch.cat([r["wavs"][b] for r in rounds])
        cache_end = 480 * (2 * 9 - 6)
        a, c = streamed[cache_end + 8000:], full[b].cpu()[cache_end + 8000:]
        assert a.numel() > 10000 and (a - c).abs().max().item() <= 1e-5
