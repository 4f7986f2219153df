"""Golden vectors for (a) the END-TO-END path at the bench shape and (b) the F0 predictor at the lengths the waveform tolerances rest on,
produced by the UNMODIFIED reference (authoring container only).

    python tests/golden/make_golden_e2e.py

  e2e_b8   configs[2]: utterance 0 of the t3_l30_b8 golden (its 250 sampled speech tokens, as `ChatterboxMultilingualTTS.generate` post-processes
           them: drop_invalid_tokens, ids < 6561) through the reference S3Gen (P = 250 prompt tokens, 10 Euler steps with CFG) and HiFT:
           the CFM mel and windows of the waveform.  The GPU test runs `ChatterboxEngine.synthesize` on all 8 utterances in one device
           batch with the same injected randomness for utterance 0 and compares that utterance.
  f0       `ConvRNNF0Predictor` of the reference on the reference's own mels of s3gen_t1000 (500 frames) and vc_t3500 (3000 frames): the
           quantity whose error drives the full-inference waveform gap (phase = 2 pi h t f0).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from provenance import provenance  # noqa: E402
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from chatterbox_amd import synth  # noqa: E402
from chatterbox_amd.engine import drop_invalid_tokens  # noqa: E402
from oracle import ref_import, ref_torch as O  # noqa: E402
from make_golden import fingerprint  # noqa: E402
from make_golden_big import WAV_WIN, _run_ref_s3gen, wav_windows  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    S3 = ref_import.load_S3Gen()
    sd = synth.s3gen_state_dict(0)
    m = S3().eval()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected
    with torch.inference_mode():
        # ---- F0 at 500 and 3000 frames, on the reference's own mels
        f0 = {}
        for name in ("s3gen_t1000", "vc_t3500"):
            g = np.load(os.path.join(OUT, name + ".npz"))
            mel = torch.from_numpy(g["mel"][0])[None]  # (1, 80, frames)
            f = m.mel2wav.f0_predictor(mel)
            o = O.f0_predict(sd, mel)
            print(f"[f0] {name}: {mel.shape[2]} frames, f0 range {float(f.min()):.1f} .. {float(f.max()):.1f}; oracle-vs-reference max-abs "
                  f"{float((o - f).abs().max()):.3e}", flush=True)
            f0[name] = f[0].numpy().astype(np.float32)
        # ---- end to end, utterance 0 of the T3 golden
        t3 = np.load(os.path.join(OUT, "t3_l30_b8.npz"))
        toks = drop_invalid_tokens(torch.from_numpy(t3["tokens"][0]).long())
        N, P = int(toks.numel()), 250
        ref = synth.s3gen_ref(n_prompt_tokens=P)
        T = 2 * (P + N)
        z = synth.randn((1, 80, T), seed=105)
        phase = (synth.rand((1, 9, 1), seed=106) * 2 - 1) * np.pi
        phase[:, 0] = 0
        noise = synth.randn((1, 9, 960 * N), seed=106)
        mel, wav = _run_ref_s3gen(m, toks[None], ref, z, phase, noise, 10)
        starts = wav_windows(960 * N, 12)
        print(f"[e2e_b8] {N} valid tokens of {t3['tokens'].shape[1]}; mel std {mel.std():.3f}; wav rms {wav.pow(2).mean().sqrt():.4f}", flush=True)
    np.savez_compressed(os.path.join(OUT, "e2e_b8.npz"), provenance=provenance(os.path.basename(__file__)), N=N, P=P, tokens=toks.numpy(), mel=mel[0].numpy(),
                        wav_win=np.stack([wav[0, s:s + WAV_WIN].numpy() for s in starts]), win_start=starts,
                        wav_rms=float(wav.pow(2).mean().sqrt()), f0_t1000=f0["s3gen_t1000"], f0_t3500=f0["vc_t3500"], fp=fingerprint(sd))


if __name__ == "__main__":
    main()
