"""CPU (-m "not gpu"): host-side logic -- weight packing, token post-processing, text normalisation, sharding,
Conditionals round trip, and the C ABI surface (library loads, every declared symbol is exported)."""
import os
import re
import sys

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
    assert _lib.lib.cbx_abi_version() == _lib.ABI_VERSION == 16


def test_ctypes_structs_match_the_c_header(tmp_path):
    """The ctypes mirrors of cbx_gemm_t / cbx_gemv_t / cbx_sampler_t must have the size and field offsets the C compiler gives
    include/cbx.h (a mismatch would silently shift every later argument)."""
    import ctypes
    import shutil
    import subprocess
    from chatterbox_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    structs = {"cbx_gemm_t": _lib.GemmParams, "cbx_gemm_pl_t": _lib.GemmPlParams, "cbx_gemv_t": _lib.GemvParams, "cbx_t3_layer_t": _lib.T3Layer, "cbx_t3_step_t": _lib.T3Step,
               "cbx_sampler_t": _lib.SamplerParams, "cbx_decode_attn_t": _lib.DecodeAttnParams, "cbx_t3_prefill_t": _lib.T3Prefill, "cbx_gpt2_layer_t": _lib.Gpt2Layer, "cbx_gpt2_prefill_t": _lib.Gpt2Prefill,
               "cbx_planes_t": _lib.PlanesRef, "cbx_cfm_tblock_t": _lib.CfmTBlock, "cbx_cfm_stage_t": _lib.CfmStage, "cbx_cfm_t": _lib.CfmSolve,
               "cbx_hift_resblock_t": _lib.HiftResblock, "cbx_hift_t": _lib.HiftDecode,
               "cbx_conformer_t": _lib.Conformer, "cbx_s3enc_t": _lib.S3Encode, "cbx_hift_f0_t": _lib.HiftF0,
               "cbx_gemv_row_t": _lib.GemvRowParams, "cbx_attn_parts_t": _lib.AttnPartsParams}  # ABI v14
    lines = []
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "cbx.h"\nint main(void) {\n' + "\n".join(lines) + "\nreturn 0; }\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_gemm_precision_scope_nests_and_restores():
    from chatterbox_amd import ops
    assert ops.GEMM_PRECISION == 0
    with ops.gemm_precision(3):
        assert ops.GEMM_PRECISION == 3
        with ops.gemm_precision(1):
            assert ops.GEMM_PRECISION == 1
        assert ops.GEMM_PRECISION == 3
    assert ops.GEMM_PRECISION == 0


def test_gemm_precision_scope_is_per_host_thread():
    """synthesize_pipelined enqueues T3 (exact fp32) on a worker thread while the flow + vocoder are enqueued inside gemm_precision(16) on the caller's:
    the scope must not leak across threads (it was a module global until round 5)."""
    import threading
    from chatterbox_amd import ops
    seen = {}
    with ops.gemm_precision(16):
        def other():
            seen["in_thread"] = ops.GEMM_PRECISION
            with ops.gemm_precision(6):
                seen["nested"] = ops.GEMM_PRECISION
        th = threading.Thread(target=other)
        th.start()
        th.join()
        assert ops.GEMM_PRECISION == 16
    assert seen == {"in_thread": 0, "nested": 6} and ops.GEMM_PRECISION == 0


def test_range_checked_repeats_at_bf16x6_and_restores(monkeypatch):
    """engine._range_checked (host logic of the default f16x3 numerics): a tripped range flag repeats the S3Gen work once at precision 6
    with a warning, the engines' precisions are restored, an untripped flag or check=False runs once."""
    import types
    import warnings
    from chatterbox_amd import engine, ops
    eng = types.SimpleNamespace(flow=types.SimpleNamespace(precision=16), hift=types.SimpleNamespace(precision=16))
    seen = []

    def run():
        seen.append((eng.flow.precision, eng.hift.precision))
        return len(seen)
    flags = iter([True])
    monkeypatch.setattr(ops, "range_flag_tripped", lambda: next(flags, False))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert engine._range_checked(eng, run) == 2
    assert seen == [(16, 16), (6, 6)] and (eng.flow.precision, eng.hift.precision) == (16, 16)
    assert any("fp16 range" in str(x.message) for x in w)
    seen.clear()
    assert engine._range_checked(eng, run) == 1 and seen == [(16, 16)]          # flag not raised
    seen.clear()
    monkeypatch.setattr(ops, "range_flag_tripped", lambda: True)
    assert engine._range_checked(eng, run, check=False) == 1 and seen == [(16, 16)]  # caller checks later
    eng.flow.precision = eng.hift.precision = 6                                    # nothing to fall back from
    seen.clear()
    assert engine._range_checked(eng, run) == 1 and seen == [(6, 6)]
    eng.flow.precision, eng.hift.precision = 16, 1                                 # only the f16x3 engine changes mode
    seen.clear()
    engine._range_checked(eng, run)
    assert seen == [(16, 1), (6, 1)] and (eng.flow.precision, eng.hift.precision) == (16, 1)
    # precision 16 is a legal scope and labelled in bench.py's roofline objects
    with ops.gemm_precision(16):
        assert ops.GEMM_PRECISION == 16


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
    from chatterbox_amd.text import punc_norm, punc_norm_en, punc_norm_turbo
    cases = ["hello world...  how are you; fine — ok", "Déjà vu: “quoted” ‘text’ ", "x", "already done!", "multi\n line\ttext - dash",
             "你好。", "こんにちは、", "ends with cjk comma，", ""]
    for fname, ours in (("mtl_tts.py", punc_norm), ("tts.py", punc_norm_en), ("tts_turbo.py", punc_norm_turbo)):
        src = open(os.path.join(ref_import.REF_SRC, fname)).read()
        start = src.index("def punc_norm")
        end = src.index("@dataclass", start)
        ns = {}
        exec(src[start:end], ns)
        for t in cases:
            assert ours(t) == ns["punc_norm"](t), (fname, t)


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


def test_turbo_norm_loudness_is_exposed_and_never_raises(capsys):
    """reference ChatterboxTurboTTS.norm_loudness (tts_turbo.py:223-239): a gain towards -27 LUFS when pyloudnorm is there, otherwise
    the input comes back with the reference's warning -- never an exception."""
    import numpy as np
    from chatterbox_amd import api
    wav = (0.1 * np.sin(np.arange(24000 * 2) * 0.05)).astype(np.float32)
    out = api.ChatterboxTurboTTS.norm_loudness(None, wav, 24000)
    try:
        import pyloudnorm  # noqa: F401
        assert out.shape == wav.shape and np.isfinite(out).all()
    except ImportError:
        assert out is wav and "norm_loudness" in capsys.readouterr().out


def test_reference_citations_point_into_existing_files():
    """Every `path.py:line[-line]` citation of the reference in the header, the docs and the package must name a file of /root/reference and
    lines inside it (skipped where the reference tree is absent, e.g. on the GPU box)."""
    import glob
    import re
    if not os.path.isdir("/root/reference/src/chatterbox"):
        pytest.skip("reference tree not present")
    ref = {}
    for root, _, files in os.walk("/root/reference"):
        for f in files:
            if f.endswith(".py"):
                ref.setdefault(f, []).append(os.path.join(root, f))
    own = {os.path.basename(f) for f in glob.glob(os.path.join(ROOT, "chatterbox_amd", "*.py")) + glob.glob(os.path.join(ROOT, "tests", "*.py")) +
           glob.glob(os.path.join(ROOT, "oracle", "*.py"))} | {"bench.py"}
    pat = re.compile(r"([A-Za-z0-9_/]+\.py):(\d+)(?:-(\d+))?")
    files = ["include/cbx.h", "DESIGN.md", "INTEGRATION.md"] + [os.path.relpath(f, ROOT) for f in glob.glob(os.path.join(ROOT, "chatterbox_amd", "*.py")) +
                                                                 glob.glob(os.path.join(ROOT, "chatterbox_amd", "csrc", "*.hip"))]
    bad, n = [], 0
    for fn in files:
        for i, line in enumerate(open(os.path.join(ROOT, fn)), 1):
            for m in pat.finditer(line):
                path, hi = m.group(1), int(m.group(3) or m.group(2))
                base = os.path.basename(path)
                if base not in ref:
                    if base not in own:
                        bad.append((fn, i, m.group(0), "no such reference file"))
                    continue
                cands = [q for q in ref[base] if q.endswith(path)] or ([] if base in own else ref[base])
                if not cands:
                    continue  # e.g. a citation of this repo's own t3.py / api.py
                n += 1
                if not any(hi <= sum(1 for _ in open(c)) for c in cands):
                    bad.append((fn, i, m.group(0), "past the end of the file"))
    assert n > 100 and not bad, bad[:10]


def test_shard_range():
    from chatterbox_amd.dist import shard_range
    for n in (0, 1, 7, 256, 257):
        for w in (1, 2, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_bench_roofline_assembly_is_pure_and_consistent():
    """bench.py's roofline objects from synthetic timer summaries (no GPU): fractions, per-launch figures, PMC lookup."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    summ = {"gemm_split": dict(launches=100, ms=5.0, flops=100 * 8e9, bytes=100 * 8e7),
            "flash_attn_f32": dict(launches=10, ms=1.3, flops=10 * 3.1e10, bytes=10 * 1.2e8),
            "gemm_f32": dict(launches=4, ms=0.8, flops=4 * 1.2e10, bytes=4 * 5e7)}
    gemv = {"qkv": dict(ms=0.42, launches=60, per_step=30, bytes=60 * 12582912.0),
            "gate_up": dict(ms=0.72, launches=60, per_step=30, bytes=60 * 33554432.0)}
    r = bench.roofline_entries(summ, elapsed=2.0, steps=3, timed_steps=1, s3_prec=3, n_decode=249, gemv=gemv)
    assert set(r) == {"gemm_split", "flash_attn_f32", "gemm_f32", "gemv_f32"}
    g = r["gemm_split"]
    assert g["bound"] == "mfma" and g["peak"] == bench.MFMA_BF16_PEAK_TFLOPS and abs(g["fp32_equivalent_tflops"] - 160.0) < 1e-6
    assert abs(g["achieved"] - 480.0) < 1e-6 and abs(g["frac"] - 0.192) < 1e-9 and g["avg_launch_us"] == 50.0
    assert r["gemm_f32"]["peak"] == bench.MFMA_F32_PEAK_TFLOPS
    v = r["gemv_f32"]
    assert v["bound"] == "hbm" and v["unit"] == "GB/s" and abs(v["achieved"] - (60 * 12582912.0 + 60 * 33554432.0) / 1.14e-3 / 1e9) < 0.1
    assert abs(v["frac"] - v["achieved"] / 8000.0) < 1e-4 and set(v["by_projection"]) == {"qkv", "gate_up"}
    # exact mode prices the same classes against the fp32 MFMA peak
    r1 = bench.roofline_entries({"gemm_f32": summ["gemm_f32"], "flash_attn_f32": summ["flash_attn_f32"]}, 2.0, 3, 1, 1, 249, None)
    assert r1["flash_attn_f32"]["peak"] == bench.MFMA_F32_PEAK_TFLOPS and "fp32_equivalent_tflops" not in r1["flash_attn_f32"]
    r16 = bench.roofline_entries(summ, elapsed=2.0, steps=3, timed_steps=1, s3_prec=16, n_decode=249, gemv=None)
    assert "f16x3" in r16["gemm_split"]["kernel"] and abs(r16["gemm_split"]["achieved"] - 480.0) < 1e-6  # 3 products, like bf16x3
    tr, src = bench.pmc_traffic("gemm_split_kernel")
    assert tr is None or (tr > 1e6 and "profiles/" in src)


def test_safetensors_reader_writer_and_conds_container(tmp_path):
    """formats.py: the container parser interoperates with the `safetensors` package both ways; voices round-trip without pickle."""
    from safetensors.torch import load_file, save_file
    from chatterbox_amd import formats
    from chatterbox_amd.api import Conditionals, T3Cond
    g = torch.Generator().manual_seed(0)
    sd = {"a.weight": torch.randn(7, 5, generator=g), "b": torch.randint(0, 100, (3,), generator=g), "c.half": torch.randn(4, generator=g).half(),
          "d.bf16": torch.randn(6, generator=g).bfloat16(), "e.empty": torch.zeros(0, 3), "f.u8": torch.arange(5, dtype=torch.uint8)}
    save_file(sd, str(tmp_path / "ref.safetensors"), metadata={"k": "v"})
    got, meta = formats.read_safetensors(tmp_path / "ref.safetensors", with_metadata=True)
    assert meta == {"k": "v"} and set(got) == set(sd) and all(torch.equal(got[k], sd[k]) and got[k].dtype == sd[k].dtype for k in sd)
    formats.write_safetensors(sd, tmp_path / "ours.safetensors", metadata={"x": 1})
    back = load_file(str(tmp_path / "ours.safetensors"))
    assert all(torch.equal(back[k], sd[k]) for k in sd)
    c = Conditionals(T3Cond(**synth.t3_cond()), synth.s3gen_ref())
    c.save(tmp_path / "voice.safetensors")
    d = Conditionals.load(tmp_path / "voice.safetensors")
    assert torch.equal(d.t3.speaker_emb, c.t3.speaker_emb) and torch.equal(d.t3.cond_prompt_speech_tokens, c.t3.cond_prompt_speech_tokens)
    assert d.t3.cond_prompt_speech_tokens.dtype == torch.int64 and torch.equal(d.gen["prompt_feat"], c.gen["prompt_feat"])
    assert d.gen["prompt_feat_len"] is None and d.t3.clap_emb is None and float(d.t3.emotion_adv) == 0.5
    fp1, fp2 = formats.fingerprint(sd), formats.fingerprint({**sd, "a.weight": sd["a.weight"] + 1})
    assert fp1 != fp2 and fp1 == formats.fingerprint(dict(reversed(list(sd.items()))))
    formats.save_packed({"x": torch.ones(3)}, tmp_path / "p.cbxpack", fp1, "kind-a")
    assert formats.load_packed(tmp_path / "p.cbxpack", fp1, "kind-a") is not None
    assert formats.load_packed(tmp_path / "p.cbxpack", fp2, "kind-a") is None and formats.load_packed(tmp_path / "p.cbxpack", fp1, "kind-b") is None


def test_safetensors_zero_dim_tensors_and_scalar_emotion_roundtrip(tmp_path):
    """0-d tensors (num_batches_tracked, a float emotion_adv) must serialise; a stale packed-layout version must not load."""
    from safetensors.torch import load_file
    from chatterbox_amd import formats
    from chatterbox_amd.api import Conditionals, T3Cond
    sd = {"bn.num_batches_tracked": torch.tensor(7), "s": torch.tensor(0.25), "v": torch.arange(3.0)}
    formats.write_safetensors(sd, tmp_path / "z.safetensors")
    back = load_file(str(tmp_path / "z.safetensors"))
    assert all(torch.equal(back[k], sd[k]) and back[k].shape == sd[k].shape for k in sd)
    cond = synth.t3_cond()
    cond["emotion_adv"] = 0.5  # the dataclass default is a python float
    c = Conditionals(T3Cond(**cond), synth.s3gen_ref())
    c.save(tmp_path / "voice.safetensors")
    assert float(Conditionals.load(tmp_path / "voice.safetensors").t3.emotion_adv) == 0.5
    fp = formats.fingerprint(sd)
    formats.save_packed({"x": torch.ones(3)}, tmp_path / "p.cbxpack", fp, formats.packed_kind("k"))
    assert formats.load_packed(tmp_path / "p.cbxpack", fp, formats.packed_kind("k")) is not None
    old = formats.PACKED_LAYOUT_VERSION
    try:
        formats.PACKED_LAYOUT_VERSION = old + 1
        assert formats.load_packed(tmp_path / "p.cbxpack", fp, formats.packed_kind("k")) is None
    finally:
        formats.PACKED_LAYOUT_VERSION = old
    # the fingerprint sees an edit anywhere in the head / tail 4 KiB, not only at 64 sampled positions
    big = {"w": torch.zeros(100000)}
    big2 = {"w": big["w"].clone()}
    big2["w"][99999 - 517] = 1.0
    assert formats.fingerprint(big) != formats.fingerprint(big2)


def test_turbo_prompt_uses_the_15s_encoder_cut():
    """ADVICE r02 / reference tts_turbo.py:112,258: Turbo tokenises up to 15 s of the 16 kHz prompt (375 tokens), English / MTL 6 s."""
    import numpy as np
    from chatterbox_amd import api, frontend as fe

    class FakeAnalyzer(fe.PromptAnalyzer):
        def __init__(self):
            self.cuts = []
            self.ve = type("VE", (), {"embeds_from_wavs": staticmethod(lambda wavs, sample_rate: torch.zeros(1, 256))})()
            self.tokenizer = self._tok

        def _tok(self, wav, max_len=None):
            self.cuts.append(wav.shape[-1])
            n = min(wav.shape[-1] // 640, max_len)
            return torch.zeros(1, n, dtype=torch.long), torch.tensor([n])

    a = FakeAnalyzer()
    w16 = np.zeros(20 * fe.S3_SR, dtype=np.float32)
    _, tok = a.t3_prompt(w16, 375, enc_cond_len=api.ChatterboxTurboTTS.ENC_COND_LEN)
    assert a.cuts[-1] == 15 * fe.S3_SR and tok.shape[1] == 375
    _, tok = a.t3_prompt(w16, 150)
    assert a.cuts[-1] == 6 * fe.S3_SR and tok.shape[1] == 150


def _example_api_calls(src):
    """(imports {name: module}, calls [(class, method, sorted keyword names)]) a program makes on names imported from `chatterbox.*`."""
    import ast
    tree = ast.parse(src)
    imports, owner, calls = {}, {}, []
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] == "chatterbox":
            for a in node.names:
                imports[a.asname or a.name] = node.module
    for node in ast.walk(tree):  # variables bound to Class.from_pretrained(...)
        if isinstance(node, ast.Assign) and isinstance(node.value, ast.Call) and isinstance(node.value.func, ast.Attribute) \
                and isinstance(node.value.func.value, ast.Name) and node.value.func.value.id in imports:
            for t in node.targets:
                if isinstance(t, ast.Name):
                    owner[t.id] = node.value.func.value.id
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and isinstance(node.func.value, ast.Name):
            base = node.func.value.id
            cls = base if base in imports else owner.get(base)
            if cls:
                calls.append((cls, node.func.attr, tuple(sorted(k.arg for k in node.keywords if k.arg))))
    return imports, calls


def test_reference_examples_use_only_the_api_we_export():
    """The drop-in boundary against the reference's own example programs, WITHOUT keeping their text in this repo: wherever the
    reference is present (this container; not the GPU box) each example_*.py is parsed and every `chatterbox.*` import, every method it calls on
    those classes and every keyword argument must (1) exist in the alias package with that parameter name and (2) be exercised by the scenario
    of tests/example_scenarios.py that stands in for it on the MI355X (tests/test_examples_gpu.py).  Everywhere: the scenarios' own CALLS table
    matches their code."""
    import importlib
    import inspect
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import example_scenarios as sc
    src_sc = open(os.path.join(ROOT, "tests", "example_scenarios.py"), encoding="utf-8").read()
    import ast
    fns = {n.name: ast.get_source_segment(src_sc, n) for n in ast.parse(src_sc).body if isinstance(n, ast.FunctionDef)}
    for name, want in sc.CALLS.items():
        _, calls = _example_api_calls(fns[name])
        assert sorted(set(calls)) == sorted(set((c, m, tuple(sorted(k))) for c, m, k in want)), (name, calls)
    pairs = {"example_tts.py": "tts", "example_tts_turbo.py": "tts_turbo", "example_tts_nano.py": "tts_nano", "example_vc.py": "vc"}
    if not os.path.isdir("/root/reference"):
        pytest.skip("the reference is only present in the authoring container")
    for ref_name, scen in pairs.items():
        imports, calls = _example_api_calls(open(os.path.join("/root/reference", ref_name), encoding="utf-8").read())
        assert calls, ref_name
        ours = set((c, m, tuple(sorted(k))) for c, m, k in sc.CALLS[scen])
        for cls, module in imports.items():
            assert hasattr(importlib.import_module(module), cls), f"{ref_name}: {module}.{cls} is not exported"
        for cls, meth, kws in calls:
            fn = getattr(getattr(importlib.import_module(imports[cls]), cls), meth)
            params = inspect.signature(fn).parameters
            assert all(k in params for k in kws), f"{ref_name}: {cls}.{meth} has no parameter(s) {[k for k in kws if k not in params]}"
            assert (cls, meth, kws) in ours, f"{ref_name}: {cls}.{meth}({', '.join(kws)}) is not exercised by scenario {scen!r}"


def test_stream_token_schedule_constant_and_growing_chunks():
    """engine.stream_token_schedule = the rounds synthesize_stream runs (tests/test_stream_gpu.py pins the constant-chunk form on the GPU):
    first_chunk + lookahead tokens, then chunk * growth^k more per round; the work of a round is proportional to prompt + tokens so far."""
    from chatterbox_amd.engine import stream_token_schedule as sched
    assert sched(20, 6, 7, 3) == [9, 16, 20] and sched(30, 6, 40, 3) == [9, 30]          # the two GPU stream tests
    assert sched(250, 25, 50, 3) == [28, 78, 128, 178, 228, 250]
    assert sched(250, 25, 50, 3, 2.0) == [28, 78, 178, 250]
    assert sched(5, 25, 50, 3) == [5] and sched(28, 25, 50, 3) == [28] and sched(29, 25, 50, 3) == [28, 29]
    P = 250                                                                             # S3Gen prompt tokens of the bench
    cost = lambda s: sum(P + n for n in s) / (P + 250)                                  # flow work in units of the one-shot synthesis
    assert cost(sched(250, 25, 50, 3)) > 4.7 and cost(sched(250, 25, 50, 3, 2.0)) < 3.1
    for g in (1.0, 1.5, 2.0, 3.0):
        s = sched(1000, 25, 50, 3, g)
        assert s[0] == 28 and s[-1] == 1000 and all(b > a for a, b in zip(s, s[1:]))


def test_decode_autotuner_child_failure_keeps_the_geometry():
    """chatterbox_amd/autotune.py: the candidates are measured in a child process; a child that dies (here: there is no GPU, its first assertion
    fails) must come back as an error report -- never as an exception or a changed geometry in the serving process.  Also: the candidate
    tables only hold knobs the engine / library know, and the split into tune keys / library knobs is total."""
    import torch
    from chatterbox_amd import autotune as at
    from chatterbox_amd.t3 import T3Engine
    for v in at.TILE_VARIANTS + at.ATTN_VARIANTS:
        t, k = at.split_variant(v)
        assert set(t) <= set(T3Engine._TUNE) and set(k) <= set(at.LIB_KNOBS) and {**t, **k} == v
    assert at.TILE_VARIANTS[0] == {} and at.env_knobs() == at.LIB_KNOBS or os.environ.get("CBX_DA_PIPE") or os.environ.get("CBX_DA_U")
    if torch.cuda.is_available():
        pytest.skip("the failure path is what this test is about: it needs a box without a GPU")
    rep = at.tune_in_child(1, 1, 8, 1, 1, 0.01, False, 0, {}, dict(at.LIB_KNOBS), timeout=300.0)
    assert "error" in rep and "best" not in rep, rep
    eng = T3Engine.__new__(T3Engine)  # the adoption logic of T3Engine.autotune without a model: an error report adopts nothing
    eng.decode_mode, eng.dev, eng.L, eng._state, eng.tune, eng.knobs = "v2", torch.device("cuda", 0), 1, {}, dict(T3Engine._TUNE), dict(at.LIB_KNOBS)
    r2 = T3Engine.autotune(eng, B=1, ctx=8, steps=1, reps=1, timeout=300.0)
    assert "error" in r2 and eng.tune == T3Engine._TUNE and eng.knobs == at.LIB_KNOBS and eng.autotune_report is r2
    r3 = T3Engine.autotune(eng, B=16)  # 32 rows: not the packed <= 16-row path -- nothing to tune, nothing spawned
    assert r3.get("skipped") and eng.tune == T3Engine._TUNE


def test_decode_autotuner_adoption_rules(monkeypatch):
    """T3Engine.autotune's adoption logic on a canned child report: without a validate callback only `best` (bit-identical) is adopted; with one,
    `best_any` (a reordering geometry) is applied, kept if the callback accepts it, and replaced by `best` if it refuses or raises."""
    import torch
    from chatterbox_amd import autotune as at
    from chatterbox_amd.t3 import T3Engine
    best, best_any = dict(qkv_tc=12, da_pipe=1), dict(od_tc=4, d_ks2=1, d_nw2=8, da_pipe=3)
    monkeypatch.setattr(at, "tune_in_child", lambda *a, **k: dict(best=dict(best), best_any=dict(best_any), candidates=[]))
    applied = []

    def mk():
        eng = T3Engine.__new__(T3Engine)
        eng.decode_mode, eng.dev, eng.L, eng._state, eng.tune, eng.knobs = "v2", torch.device("cuda", 0), 1, {}, dict(T3Engine._TUNE), dict(at.LIB_KNOBS)
        eng.apply_variant = lambda t, k=None: (applied.append((dict(t), dict(k))), setattr(eng, "tune", dict(t)), setattr(eng, "knobs", dict(k)))
        eng._prepare_tune = lambda: None
        return eng

    eng = mk()
    rep = eng.autotune(B=8)
    assert rep["adopted"] == best and eng.tune["qkv_tc"] == 12 and eng.tune["d_ks2"] == T3Engine._TUNE["d_ks2"] and eng.knobs["da_pipe"] == 1
    eng = mk()
    rep = eng.autotune(B=8, validate=lambda: eng.tune["d_ks2"] == 1 and eng.knobs["da_pipe"] == 3)  # sees best_any applied
    assert rep["best_any_validated"] is True and rep["adopted"] == best_any and eng.tune["od_tc"] == 4 and eng.tune["qkv_tc"] == 0
    for refuse in (lambda: False, lambda: 1 // 0):
        eng = mk()
        rep = eng.autotune(B=8, validate=refuse)
        assert rep["best_any_validated"] is False and rep["adopted"] == best
        assert eng.tune["qkv_tc"] == 12 and eng.tune["od_tc"] == 0 and eng.tune["d_ks2"] == T3Engine._TUNE["d_ks2"] and eng.knobs["da_pipe"] == 1
    assert "validate_error" in rep


def test_c_step_descriptor_accepts_a_state_without_split_workspace():
    """A decode state with >= 128 (row, head) pairs (B = 8: 16 rows x 16 heads) owns no split-context workspace: the C step descriptor must
    carry NULL / 0 for it (round 4's first hardware run died on `None.data_ptr()` here -- the emulator tests only built small states)."""
    import torch
    from chatterbox_amd import ops
    from chatterbox_amd._lib import T3Step
    g = ops.DecodeAttnGeom(torch.device("cpu"), split=False)
    assert g.ws is None and ops._p(g.ws) is None
    d = T3Step()
    d.da_ws, d.da_cnt, d.da_pairs = ops._p(g.ws), ops._p(g.cnt), (g.max_pairs if g.ws is not None else 0)
    assert not d.da_ws and not d.da_cnt and d.da_pairs == 0
    g2 = ops.DecodeAttnGeom(torch.device("cpu"), split=True)
    assert g2.ws.numel() == 128 * 8 * 66 and g2.cnt.dtype == torch.int32 and not bool(g2.cnt.any())


def test_green_allow_list_is_canonical_and_contains_the_default_geometry():
    """chatterbox_amd/decode_green.json (the geometries bench.py may run or adopt): every entry made of knobs the engine / library know, in canon
    form against the FROZEN round-3 base (so entries keep their meaning when defaults move); the frozen base and the engines' current default
    geometry are on it; every geometry the autotuner can compose from the default is on it too (else bench.py would skip it -- allowed, but then
    the candidate list is stale)."""
    import json
    from chatterbox_amd import autotune as at
    from chatterbox_amd.t3 import T3Engine
    green = at.green_variants()
    assert () in green and at.canon({}) == () and at.canon(dict(qkv_tc=0, da_pipe=0, da_u=4)) == ()
    assert at.canon(dict(da_pipe=7, pre_epi=1, qkv_tc=0)) == (("da_pipe", 7), ("pre_epi", 1))
    assert at.canon(dict(qkv_ks=0, qkv_ct=2)) == () and at.canon(dict(qkv_ks=4, qkv_ct=2)) == (("qkv_ct", 2), ("qkv_ks", 4))
    assert set(at.BASE_TUNE) == set(T3Engine._TUNE), "the frozen base knows every tune key"
    known = set(T3Engine._TUNE) | set(at.LIB_KNOBS)
    doc = json.load(open(at.GREEN_FILE))
    assert isinstance(doc["green"], list) and doc.get("source"), "decode_green.json names the hardware run it was written from"
    for v in doc["green"]:
        assert set(v) <= known and at.canon(v) in green, v
    assert at.canon(T3Engine._TUNE, at.LIB_KNOBS) in green, "the default geometry must be hardware-verified"
    missing = [at.canon(t, k) for t, k in at.composed_candidates(T3Engine._TUNE, at.LIB_KNOBS) if at.canon(t, k) not in green]
    assert not missing, f"{len(missing)} composable candidates are not on the allow-list, e.g. {missing[:3]}"


def test_stage_seams_reject_an_empty_descriptor():
    """The stage-level entry points of ABI v12 validate their descriptor before the first launch (no GPU needed to see that)."""
    import ctypes
    from chatterbox_amd import _lib
    for fn, cls in ((_lib.lib.cbx_cfm_solve, _lib.CfmSolve), (_lib.lib.cbx_hift_decode, _lib.HiftDecode), (_lib.lib.cbx_s3gen_encode, _lib.S3Encode),
                    (_lib.lib.cbx_hift_f0_source, _lib.HiftF0)):
        assert fn(ctypes.byref(cls()), None) == -22 and b"null descriptor" in _lib.lib.cbx_last_error()


def test_c_blocks_of_integration_md_compile(tmp_path):
    """Every ```c block of INTEGRATION.md is a translation unit against include/cbx.h (the header is plain C; the sketch of a C host stays in step with it)."""
    import re
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    blocks = re.findall(r"```c\n(.*?)```", open(os.path.join(ROOT, "INTEGRATION.md")).read(), flags=re.S)
    assert blocks, "INTEGRATION.md holds at least the C host sketch"
    for i, b in enumerate(blocks):
        src = tmp_path / f"block{i}.c"
        src.write_text(b)
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])
