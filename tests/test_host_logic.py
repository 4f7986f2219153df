"""CPU (-m "not gpu"): host-side logic -- weight packing, token post-processing, text normalisation, sharding,
Conditionals round trip, and the C ABI surface (library loads, every declared symbol is exported)."""
import os
import re

import pytest
import torch
import torch.nn.functional as F

from chatterbox_amd import synth, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    import ctypes
    hdr = open(os.path.join(ROOT, "include", "cbx.h")).read()
    names = set(re.findall(r"\b(?:int|const char\*)\s+(cbx_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 15
    lib = ctypes.CDLL(os.path.join(ROOT, "chatterbox_amd", "libcbx_hip.so"))
    for n in sorted(names):
        assert hasattr(lib, n), f"libcbx_hip.so does not export {n}"
    from chatterbox_amd import _lib  # binding table covers the same set
    assert names == set(_lib._SIGS), names ^ set(_lib._SIGS)
    assert _lib.lib.cbx_abi_version() == _lib.ABI_VERSION == 2


def test_ops_fail_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from chatterbox_amd import ops
    with pytest.raises(Exception):
        ops.linear(torch.zeros(4, 16), torch.zeros(8, 16), torch.zeros(4, 8))  # CPU tensors are rejected, no fallback


@pytest.mark.parametrize("cin,cout,k,s,p", [(8, 6, 16, 8, 4), (8, 6, 11, 5, 3), (8, 6, 7, 3, 2)])
def test_conv_transpose_phase_packing(cin, cout, k, s, p):
    g = torch.Generator().manual_seed(1)
    x, w, b = torch.randn(2, cin, 13, generator=g), torch.randn(cin, cout, k, generator=g), torch.randn(cout, generator=g)
    wp, bp = weights.pack_conv_transpose(w, b, s, p)
    xp = F.pad(x, (1, 1)).transpose(1, 2)  # (B, T+2, cin): taps at input offsets -1, 0, +1
    cols = torch.cat([xp[:, j:j + 13] for j in range(3)], -1)
    y = (cols @ wp.t() + bp).reshape(2, 13 * s, cout).transpose(1, 2)
    assert torch.allclose(y, F.conv_transpose1d(x, w, b, stride=s, padding=p), atol=1e-5)


def test_conv_and_swiglu_packing_and_weight_norm_fold():
    g = torch.Generator().manual_seed(2)
    w = torch.randn(5, 4, 3, generator=g)
    x = torch.randn(1, 4, 9, generator=g)
    cols = torch.cat([F.pad(x, (2, 0)).transpose(1, 2)[:, j:j + 9] for j in range(3)], -1)
    assert torch.allclose(cols @ weights.pack_conv(w).t(), F.conv1d(F.pad(x, (2, 0)), w).transpose(1, 2), atol=1e-5)
    assert weights.pack_conv(w, cin_pad=16).shape == (5, 48)
    gate, up = torch.randn(64, 8, generator=g), torch.randn(64, 8, generator=g)
    pk = weights.pack_swiglu(gate, up)
    assert torch.equal(pk[:32], gate[:32]) and torch.equal(pk[32:64], up[:32]) and torch.equal(pk[64:96], gate[32:])
    sd = {}
    synth._wn_conv(sd, "c", 6, 4, 3, 0)
    ref = torch.nn.utils.parametrizations.weight_norm(torch.nn.Conv1d(4, 6, 3))
    ref.load_state_dict({**{k[2:]: v for k, v in sd.items()}, "bias": torch.zeros(6)})
    assert torch.allclose(weights.fold_weight_norm(sd, "c"), ref.weight, atol=1e-6)
    legacy = {"c.weight_g": sd["c.parametrizations.weight.original0"], "c.weight_v": sd["c.parametrizations.weight.original1"]}
    assert torch.allclose(weights.fold_weight_norm(legacy, "c"), ref.weight, atol=1e-6)


def test_synth_checkpoints_use_reference_key_layout():
    t3 = synth.t3_state_dict(2, 0)
    assert t3["tfmr.layers.1.mlp.down_proj.weight"].shape == (1024, 4096) and t3["speech_head.weight"].shape == (8194, 1024)
    assert t3["cond_enc.perceiver.pre_attention_query"].shape == (1, 32, 1024) and "tfmr.layers.2.mlp.up_proj.weight" not in t3
    assert len(t3) == 40 - 18 + 9 * 2  # 22 non-layer tensors + 9 per layer
    s3 = synth.s3gen_state_dict(0)
    assert s3["flow.decoder.estimator.up_blocks.0.0.block1.block.0.weight"].shape == (256, 512, 3)
    assert s3["mel2wav.ups.1.parametrizations.weight.original0"].shape == (256, 1, 1)
    assert s3["mel2wav.source_downs.0.weight"].shape == (256, 18, 30)
    assert len(s3) == 1449  # == len(S3Token2Wav().state_dict()) minus tokenizer.* / speaker_encoder.* (checked vs the reference)
    a, b = synth.t3_state_dict(1, 0), synth.t3_state_dict(2, 0)
    assert all(torch.equal(a[k], b[k]) for k in a)  # per-key seeding: shallower models are prefixes


def test_drop_invalid_tokens_and_text():
    pytest.importorskip("chatterbox_amd._lib")
    from chatterbox_amd.engine import drop_invalid_tokens
    x = torch.tensor([6561, 5, 7000, 17, 6562, 3])
    assert drop_invalid_tokens(x).tolist() == [5, 17]
    assert drop_invalid_tokens(torch.tensor([1, 2, 3])).tolist() == [1, 2, 3]
    from chatterbox_amd.text import punc_norm
    assert punc_norm("") == "You need to add some text for me to talk."
    assert punc_norm("hello   world") == "Hello world."
    assert punc_norm("wait… what?") == "Wait,  what?"  # same double space as the reference (replacement after whitespace squeeze)
    assert punc_norm("a – b") == "A - b."


def test_punc_norm_matches_reference_when_available():
    from oracle import ref_import
    if not ref_import.available():
        pytest.skip("reference tree not present (GPU box)")
    src = open(os.path.join(ref_import.REF_SRC, "mtl_tts.py")).read()
    start, end = src.index("def punc_norm"), src.index("@dataclass")
    ns = {}
    exec(src[start:end], ns)
    from chatterbox_amd.text import punc_norm
    for t in ["hello world...  how are you; fine — ok", "Déjà vu: “quoted” ‘text’ ", "x", "already done!", "multi\n line\ttext - dash"]:
        assert punc_norm(t) == ns["punc_norm"](t), t


def test_conditionals_roundtrip_and_api_surface(tmp_path):
    pytest.importorskip("chatterbox_amd._lib")
    from chatterbox_amd.api import (ChatterboxMultilingualTTS, Conditionals, T3Cond, _resolve_multilingual_t3_model)
    c = Conditionals(T3Cond(**synth.t3_cond()), synth.s3gen_ref())
    c.save(tmp_path / "conds.pt")
    d = Conditionals.load(tmp_path / "conds.pt")
    assert torch.equal(d.t3.speaker_emb, c.t3.speaker_emb) and torch.equal(d.gen["prompt_feat"], c.gen["prompt_feat"])
    assert d.gen["prompt_feat_len"] is None
    assert _resolve_multilingual_t3_model("v3") == "t3_mtl23ls_v3.safetensors"
    with pytest.raises(ValueError):
        _resolve_multilingual_t3_model("v9")
    m = ChatterboxMultilingualTTS.__new__(ChatterboxMultilingualTTS)
    m.conds = c
    with pytest.raises(ValueError):
        m.generate("hi", language_id="xx")
    assert ChatterboxMultilingualTTS.get_supported_languages()["sw"] == "Swahili"
    import chatterbox.mtl_tts as compat
    assert compat.ChatterboxMultilingualTTS is ChatterboxMultilingualTTS


def test_shard_range():
    from chatterbox_amd.dist import shard_range
    for n in (0, 1, 7, 256, 257):
        for w in (1, 2, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
