"""Stage-level C entry points of the flow decoder and the vocoder (-m gpu; ABI v12): `cbx_cfm_solve` and `cbx_hift_decode` sequence the same
kernel-level launches, with the same arguments, as `FlowEngine.cfm` / `HiFTEngine.decode` issue one by one -- so their results must be BIT-identical
to the Python sequencing (which the golden / oracle tests of tests/test_models_gpu.py pin against the reference).  The bodies take the device as an
argument: tests/test_simt_kernels.py runs them on the SIMT emulator at smaller shapes.  The file sorts after every other `-m gpu` file on purpose.
"""

import pytest
import torch

pytestmark = pytest.mark.gpu
# All four entry points are the engines' DEFAULT path since round 5 (first hardware run of cbx_s3gen_encode / cbx_hift_f0_source: round 5, call A,
# profiles/r05_a_seams_first_hardware_run.log -- green), so every S3Gen / HiFT golden of tests/test_models_gpu.py passes through them as well.


@pytest.mark.parametrize("meanflow,T,fused_qkv,fused_ln", [(False, 152, True, 1), (False, 150, True, 2), (True, 152, True, 1), (False, 152, False, 2),
                                                           (False, 152, True, 0)])
def test_cfm_solve_through_the_c_entry_point_equals_the_python_sequence(dev, meanflow, T, fused_qkv, fused_ln, n_mid=2, B=3, n_steps=3):
    """solve_euler (CFG or meanflow) on the plane-format estimator: cbx_cfm_solve against FlowEngine.cfm's own launch sequence, for a ragged batch, the fused
    and the separate q | k | V^T projection (T % 4 != 0 forces the separate one), LayerNorm from the GEMM epilogues and as launches of its own."""
    from chatterbox_amd import ops, synth
    from chatterbox_amd.s3gen import FlowEngine
    sd = synth.s3gen_state_dict(0, meanflow=meanflow, n_mid=n_mid, n_enc=1, n_up_enc=1)
    eng = FlowEngine(sd, dev, meanflow=meanflow)
    eng.fused_qkv, eng.fused_ln = fused_qkv, fused_ln
    mu, cond, z = (synth.randn((B, T, 80), seed=s).to(dev) for s in (1, 2, 3))
    cond[:, T // 3:] = 0
    spk = synth.randn((B, 80), seed=4).to(dev)
    lens = torch.tensor([T, T - 7, max(2, T // 2)][:B], dtype=torch.int32, device=dev)
    out, calls, inner = {}, [], eng._cfm_solve_c
    eng._cfm_solve_c = lambda *a, **k: (calls.append(1), inner(*a, **k))[1]
    for seam in (False, True):
        eng.c_seam = seam
        with ops.gemm_precision(16), torch.inference_mode():
            assert eng._planes_ok((1 if meanflow else 2) * B, T)
            out[seam] = eng.cfm(mu, lens, spk, cond, z, n_steps=n_steps).clone()
    assert len(calls) == 1, "the second pass went through cbx_cfm_solve"
    assert torch.isfinite(out[True]).all() and out[True].abs().max() > 0
    assert torch.equal(out[True], out[False]), f"max |diff| {(out[True] - out[False]).abs().max().item():.3e}"


@pytest.mark.parametrize("ragged,fade,precision", [(True, False, 16), (False, True, 16), (True, True, 1)])
def test_hift_decode_through_the_c_entry_point_equals_the_python_sequence(dev, ragged, fade, precision, T=20, B=3):
    """HiFTGenerator.decode: cbx_hift_decode against HiFTEngine.decode's own launch sequence (ragged batch / full rows, with and without trim_fade,
    f16x3 and exact fp32 convs)."""
    from chatterbox_amd import synth
    from chatterbox_amd.hift import HiFTEngine
    eng = HiFTEngine(synth.s3gen_state_dict(0), dev, precision=precision)
    mel = (synth.randn((B, T, 80), seed=9) * 1.5 - 4.0).to(dev)
    s = torch.tanh(synth.randn((B, 480 * T), seed=10)).to(dev)
    lens = torch.tensor([T, max(1, T - 7), max(1, T // 3)][:B], dtype=torch.int32, device=dev) if ragged else None
    from chatterbox_amd import ops
    out, calls, inner = {}, [], eng._decode_c
    eng._decode_c = lambda *a, **k: (calls.append(1), inner(*a, **k))[1]
    for seam in (False, True):
        eng.c_seam = seam
        with ops.gemm_precision(precision):
            out[seam] = eng.decode(mel, s, lens=lens, fade=fade).clone()
    assert len(calls) == 1, "the second pass went through cbx_hift_decode"
    assert torch.isfinite(out[True]).all() and out[True].abs().max() > 0
    assert torch.equal(out[True], out[False]), f"max |diff| {(out[True] - out[False]).abs().max().item():.3e}"


def test_s3gen_encode_through_the_c_entry_point_equals_the_python_sequence(dev, B=3, N=60, n_enc=2, n_up=2):
    """UpsampleConformerEncoder.forward + encoder_proj (flash rel-pos form): cbx_s3gen_encode against FlowEngine._encode_rows' own launch sequence, ragged batch."""
    from chatterbox_amd import ops, synth
    from chatterbox_amd.s3gen import FlowEngine
    eng = FlowEngine(synth.s3gen_state_dict(0, n_mid=1, n_enc=n_enc, n_up_enc=n_up), dev)
    tok = synth.speech_tokens(B * N, seed=3).view(B, N).to(dev)
    lens = torch.tensor([N, max(1, N - 3), max(1, N // 2)][:B], dtype=torch.int32, device=dev)
    out, calls, inner = {}, [], eng._encode_c
    eng._encode_c = lambda *a, **k: (calls.append(1), inner(*a, **k))[1]
    for seam in (False, True):
        eng.c_seam = seam
        with ops.gemm_precision(16), torch.inference_mode():
            out[seam] = eng.encode(tok, lens).clone()
    assert len(calls) == 1, "the second pass went through cbx_s3gen_encode"
    assert torch.isfinite(out[True]).all() and out[True].abs().max() > 0
    assert torch.equal(out[True], out[False]), f"max |diff| {(out[True] - out[False]).abs().max().item():.3e}"


def test_hift_f0_source_through_the_c_entry_point_equals_the_python_sequence(dev, B=3, T=20):
    """The front half of HiFTGenerator.inference (F0 predictor + source module): cbx_hift_f0_source against HiFTEngine.f0_predict + source, ragged batch."""
    from chatterbox_amd import synth
    from chatterbox_amd.hift import HiFTEngine
    eng = HiFTEngine(synth.s3gen_state_dict(0), dev)
    mel = (synth.randn((B, T, 80), seed=9) * 1.5 - 4.0).to(dev)
    phase, noise = synth.rand((B, 9), seed=5) * 6.28 - 3.14, synth.randn((B, 9, 480 * T), seed=6)
    phase[:, 0] = 0
    lens = torch.tensor([T, max(1, T - 7), max(1, T // 2)][:B], dtype=torch.int32, device=dev)
    eng.decode = lambda mel, s, lens=None, fade=True: s.clone()  # the front half only: inference() hands the source to decode()
    out, calls, inner = {}, [], eng._f0_source_c
    eng._f0_source_c = lambda *a, **k: (calls.append(1), inner(*a, **k))[1]
    for seam in (False, True):
        eng.c_seam = seam
        out[seam] = eng.inference(mel, phase=phase.to(dev), noise=noise.to(dev), lens=lens)[1].clone()
    assert len(calls) == 1, "the second pass went through cbx_hift_f0_source"
    assert torch.isfinite(out[True]).all() and out[True].abs().max() > 0
    assert torch.equal(out[True], out[False]), f"max |diff| {(out[True] - out[False]).abs().max().item():.3e}"


def test_t3_token_loop_in_c_equals_the_python_replay_loop_and_stops_at_eos(dev):
    """cbx_t3_loop_* (ABI v14: the library captures cbx_t3_decode_step in a hipGraph and replays it; the reference's loop t3.py:338-386 incl. its EOS test):
    (1) T3Engine.generate through the C loop samples the tokens of the Python replay loop over a torch-captured graph (and of the eager step), for a ragged
    batch; (2) chunked use (async generate + advance) on the C loop equals the one-shot run; (3) cbx_t3_loop_run polls the done flags: with every utterance
    flagged done it stops at the first poll."""
    import ctypes
    from chatterbox_amd import synth
    from chatterbox_amd._lib import check, lib
    from chatterbox_amd.t3 import T3Engine
    L, steps = 2, 21
    eng = T3Engine(synth.t3_state_dict(L, 0), dev)
    assert eng.c_loop and eng.c_step
    texts = [synth.text_tokens(9, seed=1), synth.text_tokens(15, seed=2), synth.text_tokens(12, seed=3)]
    cond, u = synth.t3_cond(), synth.rand((3, steps), seed=5)
    kw = dict(max_new_tokens=steps, uniforms=u, ban_eos=True, ban_from=6561, temperature=0.8, cfg_weight=0.5, repetition_penalty=1.2, min_p=0.05, top_p=1.0)
    calls, inner = [], eng._run_c_loop
    eng._run_c_loop = lambda *a, **k: (calls.append(a[1]), inner(*a, **k))[1]
    got = [t.tolist() for t in eng.generate(cond, texts, **kw)]
    assert calls == [steps - 1], "generate() ran its token loop through cbx_t3_loop_run"
    eng.c_loop = False
    for st in eng._state.values():
        st.pop("cstep", None)
    ref_graph = [t.tolist() for t in eng.generate(cond, texts, **kw)]
    ref_eager = [t.tolist() for t in eng.generate(cond, texts, use_graph=False, **kw)]
    assert got == ref_graph == ref_eager
    eng.c_loop = True
    h = eng.generate(cond, texts, async_mode=True, run_steps=6, **kw)
    eng.advance(h, 7)
    eng.advance(h, 100)
    assert [t.tolist() for t in eng.collect(h)] == got
    # (3) the EOS poll: flag everything done, ask for 12 steps with a poll every 4 -> the loop stops after the first poll
    st = h["st"]
    with torch.inference_mode():
        for k in ("step", "n_generated"):
            st[k].zero_()
        st["done"].fill_(1)
    torch.cuda.synchronize()
    ran = ctypes.c_int(-1)
    check(lib.cbx_t3_loop_run(eng._c_loop(st), 12, 4, torch.cuda.current_stream().cuda_stream, ctypes.byref(ran)), "cbx_t3_loop_run")
    assert ran.value == 4 and int(st["step"].sum()) == 0, "finished utterances are no-ops in the sampler; the loop ended at the first poll"
    with torch.inference_mode():
        st["done"].zero_()
