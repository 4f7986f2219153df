"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Imports the *unmodified* reference modules from /root/reference (present in the
authoring container only, never on the GPU box) so that

  * oracle/ref_torch.py (our CPU restatement) can be validated against them, and
  * tests/golden/make_golden.py can generate the committed golden vectors.

The reference package cannot be imported as shipped (its __init__ needs
installed metadata plus librosa/perth/diffusers/... which are absent here), so
namespace stubs are registered for `chatterbox` / `chatterbox.models` and small
shim modules stand in for the missing third-party packages.  The shims restate
the published semantics of diffusers==0.29.0 `Attention` (AttnProcessor2_0) and
`GELU`, which the reference pins in pyproject.toml:24 and calls from
src/chatterbox/models/s3gen/matcha/transformer.py:5-14,196-204,110.  They are
"parity unpinned" pieces: no reference test pins them.
"""
import importlib.machinery
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = os.environ.get("CBX_REFERENCE_ROOT", "/root/reference")
REF_SRC = os.path.join(REF_ROOT, "src", "chatterbox")


def available() -> bool:
    return os.path.isdir(REF_SRC)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _pkg(name, path):
    m = _mod(name)
    m.__path__ = [path]
    return m


class _DiffusersAttention(nn.Module):
    """diffusers 0.29.0 `Attention` with the default AttnProcessor2_0, self-attention only."""

    def __init__(self, query_dim, heads=8, dim_head=64, dropout=0.0, bias=False,
                 cross_attention_dim=None, upcast_attention=False, **_):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(query_dim, inner, bias=bias)
        self.to_v = nn.Linear(query_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Dropout(dropout)])

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **_):
        B, T, _c = hidden_states.shape
        if attention_mask is not None:
            # prepare_attention_mask: (B,1,T) -> repeat_interleave(heads) -> (B, heads, 1, T)
            attention_mask = attention_mask.repeat_interleave(self.heads, dim=0)
            attention_mask = attention_mask.view(B, self.heads, -1, attention_mask.shape[-1])
        q = self.to_q(hidden_states)
        k = self.to_k(hidden_states)
        v = self.to_v(hidden_states)
        hd = q.shape[-1] // self.heads
        q = q.view(B, -1, self.heads, hd).transpose(1, 2)
        k = k.view(B, -1, self.heads, hd).transpose(1, 2)
        v = v.view(B, -1, self.heads, hd).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(B, -1, self.heads * hd).to(q.dtype)
        o = self.to_out[0](o)
        return self.to_out[1](o)


class _DiffusersGELU(nn.Module):
    def __init__(self, dim_in, dim_out, approximate="none"):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)
        self.approximate = approximate

    def forward(self, x):
        return F.gelu(self.proj(x), approximate=self.approximate)


class _Unused(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("not on the inference path")


def _get_activation(name):
    return {"silu": nn.SiLU, "swish": nn.SiLU, "mish": nn.Mish, "gelu": nn.GELU, "relu": nn.ReLU}[name]()


class _S3TokStub(nn.Module):
    """Stand-in for the third-party `s3tokenizer.S3TokenizerV2` (source not in the reference tree)."""

    def __init__(self, name="speech_tokenizer_v2_25hz", config=None):
        super().__init__()
        self._dummy = nn.Parameter(torch.zeros(1))


def _slaney_mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    """librosa.filters.mel (slaney scale + slaney norm) restated with numpy."""
    import numpy as np
    fmax = fmax or sr / 2.0

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        f_sp = 200.0 / 3
        mels = f / f_sp
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = np.log(6.4) / 27.0
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        f_sp = 200.0 / 3
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = np.log(6.4) / 27.0
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (w * enorm[:, None]).astype(np.float32)


class _S3TokConfig:
    n_mels = 128
    n_audio_state = 1280
    n_audio_head = 20
    n_audio_layer = 6
    n_codebook_size = 3 ** 8


_DONE = False


def install():
    """Register the namespace stubs + third-party shims (idempotent)."""
    global _DONE
    if _DONE:
        return
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT} (it only exists in the authoring container)")
    import transformers  # noqa: F401  (must precede the torchaudio stub)
    from transformers import LlamaModel, GPT2Model  # noqa: F401

    _pkg("chatterbox", REF_SRC)
    _pkg("chatterbox.models", os.path.join(REF_SRC, "models"))

    _pkg("diffusers", "/nonexistent")
    _pkg("diffusers.models", "/nonexistent")
    _pkg("diffusers.utils", "/nonexistent")
    _mod("diffusers.models.attention", GEGLU=_Unused, GELU=_DiffusersGELU, AdaLayerNorm=_Unused,
         AdaLayerNormZero=_Unused, ApproximateGELU=_Unused)
    _mod("diffusers.models.attention_processor", Attention=_DiffusersAttention)
    _mod("diffusers.models.lora", LoRACompatibleLinear=nn.Linear)
    _mod("diffusers.models.activations", get_activation=_get_activation)
    _mod("diffusers.utils.torch_utils", maybe_allow_in_graph=lambda c: c)
    _mod("conformer", ConformerBlock=_Unused)
    _mod("omegaconf", DictConfig=dict)
    _pkg("s3tokenizer", "/nonexistent")
    _mod("s3tokenizer.utils", padding=lambda *a, **k: None)
    _mod("s3tokenizer.model_v2", S3TokenizerV2=_S3TokStub, ModelConfig=_S3TokConfig)
    lib = _pkg("librosa", "/nonexistent")
    lib.filters = _mod("librosa.filters", mel=_slaney_mel)
    ta = _pkg("torchaudio", "/nonexistent")
    ta.transforms = _mod("torchaudio.transforms", Resample=_Unused)
    ta.compliance = _pkg("torchaudio.compliance", "/nonexistent")
    ta.compliance.kaldi = _mod("torchaudio.compliance.kaldi", fbank=None)
    _DONE = True


def load_T3():
    install()
    from chatterbox.models.t3.t3 import T3
    from chatterbox.models.t3.modules.t3_config import T3Config
    from chatterbox.models.t3.modules.cond_enc import T3Cond
    from chatterbox.models.t3 import llama_configs
    return T3, T3Config, T3Cond, llama_configs


def load_S3Gen():
    install()
    from chatterbox.models.s3gen.s3gen import S3Token2Wav
    return S3Token2Wav
