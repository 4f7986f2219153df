"""Golden vectors of the prompt-analysis front-end, produced by the UNMODIFIED reference (authoring container only).

    python tests/golden/make_golden_frontend.py

  frontend.npz:
    s3_logmel   S3Tokenizer.log_mel_spectrogram (reference s3tokenizer.py:128-168) of synth.prompt_wav(3 s, 16 kHz)
    mel24k      mel_spectrogram (reference s3gen/utils/mel.py:41-85) of synth.prompt_wav(3 s, 24 kHz)
    xvector     CAMPPlus.forward (reference s3gen/xvector.py) on mean-normalised fbank features (features produced by the oracle's
                Kaldi restatement and stored: the network body is what is pinned)
    ve_embed    VoiceEncoder.inference (reference voice_encoder.py:165-200) on the oracle's 40-bin mel (stored)
The third-party pieces the reference calls but does not contain (S3TokenizerV2.quantize, Kaldi fbank, librosa.stft, resamplers)
cannot be pinned here; they are restated in oracle/ref_frontend.py and labelled unpinned.
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
from oracle import ref_frontend as RF, ref_import  # noqa: E402
from make_golden import fingerprint  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    ref_import.install()
    from chatterbox.models.s3tokenizer.s3tokenizer import S3Tokenizer
    from chatterbox.models.s3gen.utils.mel import mel_spectrogram
    from chatterbox.models.s3gen.xvector import CAMPPlus
    from chatterbox.models.voice_encoder.voice_encoder import VoiceEncoder

    w16, w24 = synth.prompt_wav(3.0, 16000), synth.prompt_wav(3.0, 24000)
    with torch.inference_mode():
        tok = S3Tokenizer()  # the third-party base class is a stub: only the reference's own log_mel_spectrogram is exercised
        type(tok).device = property(lambda self: torch.device("cpu"))
        lm = tok.log_mel_spectrogram(w16[None])[0]                       # (128, T)
        e = (RF.s3_log_mel(w16) - lm).abs()
        print(f"s3 log-mel {tuple(lm.shape)}: oracle-vs-reference max {e.max():.3e} mean {e.mean():.3e}")
        m24 = mel_spectrogram(w24[None])[0]                               # (80, frames)
        e = (RF.mel_spectrogram_24k(w24)[0] - m24).abs()
        print(f"24 kHz mel {tuple(m24.shape)}: oracle-vs-reference max {e.max():.3e} mean {e.mean():.3e}")

        csd = synth.campplus_state_dict(0, prefix="")
        cam = CAMPPlus().eval()
        cam.load_state_dict(csd, strict=True)
        fb = RF.kaldi_fbank(w16)
        fb = fb - fb.mean(0, keepdim=True)
        xv = cam(fb[None])[0]
        e = (RF.campplus_forward(csd, fb[None])[0] - xv).abs().max()
        print(f"CAMPPlus x-vector: |x| max {xv.abs().max():.3f}, oracle-vs-reference max {e:.3e}")

        vsd = synth.voice_encoder_state_dict(0)
        ve = VoiceEncoder().eval()
        ve.load_state_dict(vsd, strict=True)
        vmel = RF.ve_melspectrogram(w16)
        emb = ve.inference(vmel[None], [vmel.shape[0]], rate=1.3)[0]
        e = (RF.ve_inference(vsd, vmel) - emb).abs().max()
        print(f"voice encoder: {vmel.shape[0]} frames, oracle-vs-reference max {e:.3e}")
    np.savez_compressed(os.path.join(OUT, "frontend.npz"), provenance=provenance(os.path.basename(__file__)), s3_logmel=lm.numpy(), mel24k=m24.numpy(), fbank=fb.numpy(), xvector=xv.numpy(),
                        ve_mel=vmel.numpy(), ve_embed=emb.numpy(), fp_cam=fingerprint(csd), fp_ve=fingerprint(vsd))


if __name__ == "__main__":
    main()
