#!/usr/bin/env python
"""bench.py -- headline metric of BASELINE.json: audio-seconds per wall-second (xRT) + p50 first-audio latency of
generate() = T3 -> S3Gen (10-step CFM, CFG) -> HiFT, Multilingual-V3 500M architecture, batch 8 per GPU.

    python bench.py --gpus N --steps K --warmup W        (N > 1: one rank per GPU over RCCL -- either the caller launched the ranks with
                                                          torch.distributed.run, or this script re-runs itself under it)

A "step" is one pass of the whole hot path over one batch of synthetic utterances: 64 text tokens,
250 speech tokens (10 s of audio, EOS banned so the length is fixed), 150-token T3 voice prompt, 250-token / 500-frame
S3Gen prompt, seeded random-init weights in the reference's checkpoint layout (no network => no pretrained weights).
Prints ONE JSON line (rank 0).  The headline `value` runs S3Gen in the default fp32-level numerics mode (f16x3: two fp16 planes per fp32
operand, range-checked on the device; T3 is exact fp32 MFMA); the other fp32-level mode (bf16x6) and the opt-in bf16x3 fast mode are
measured by the same run and reported beside it, clearly labelled.  `roofline` is the
dominant kernel class (T3 decode weight streaming); `decode_step` is the whole decode step (weights + KV cache) timed with HIP events
INSIDE the timed region; `cpu_baseline` is the reference itself (kind "reference", when /root/reference is present) or the oracle
(kind "port", a CPU restatement pinned against the reference) on ONE utterance of the benched workload.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, exact fp32
MFMA_BF16_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_bf16, dense (the split-operand kernels issue 3 or 6 of these per fp32 product)
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU per step (default 8; vc60: 1)")
    ap.add_argument("--tokens", type=int, default=250, help="speech tokens per utterance (25/s)")
    ap.add_argument("--text-tokens", type=int, default=64)
    ap.add_argument("--t3-layers", type=int, default=30)
    ap.add_argument("--workload", default="mtl", choices=["mtl", "turbo", "nano", "vc60"],
                    help="mtl = configs[2] (the headline metric); turbo / nano = configs[1] / configs[0] architectures (GPT-2 T3, 2-step meanflow); "
                         "vc60 = configs[4]: voice conversion of 60 s utterances (S3 tokenizer -> S3Gen at T = 3500 mel frames -> HiFT; no T3), "
                         "--batch utterances per step (default there: 1)")
    ap.add_argument("--vc-seconds", type=float, default=60.0, help="vc60: length of the source utterance")
    ap.add_argument("--schedule", default="auto", choices=["auto", "pipelined", "serial"],
                    help="how the K timed steps (batches) run.  auto (default) = pipelined, falling back to serial (and saying so) if the throughput schedule fails on a "
                         "single-GPU run; an EXPLICIT pipelined fails instead of falling back.  pipelined (the headline schedule since round 5; the throughput schedule, engine.synthesize_pipelined): T3 of batch "
                         "k + 1 on a high-priority HIP stream beside the CFM + vocoder of batch k on a second one, both on their co-resident kernel forms, the "
                         "T3 launches enqueued by a second host thread; fill and drain are inside the timed region; same work, same results, about twice the "
                         "per-batch latency.  serial: the K batches strictly one after the other (the latency schedule; rounds 1-4's headline).  Whichever "
                         "is the headline, the other one is measured after the timed region and reported beside it.")
    ap.add_argument("--pipelined", action="store_true", help="= --schedule pipelined (kept for compatibility)")
    ap.add_argument("--serial", action="store_true", help="= --schedule serial (kept for compatibility)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tokens", type=int, default=None, help="speech tokens of the CPU-baseline utterance (default: --tokens, i.e. the benched workload)")
    ap.add_argument("--roofline-kernel", default="auto", choices=["auto", "gemm_f32", "gemm_split", "flash_attn_f32", "gemv_f32", "gemm_planes", "flash_attn_planes"],
                    help="kernel class reported as `roofline` (auto: the one with the largest share of a step); the others go to "
                         "`roofline_secondary`.  All three are timed with HIP events on the launch stream")
    ap.add_argument("--s3gen-precision", type=int, default=None, choices=[1, 3, 6, 16],
                    help="numerics of the S3Gen GEMMs / attention: 1 exact fp32 MFMA, 16 f16x3 (default: fp32-level error, every fp32 parity "
                         "tolerance holds; fp16 operand range, checked on the device), 6 bf16x6 (fp32-level, fp32 range), 3 bf16x3 (opt-in "
                         "fast mode, bf16-mode tolerances).  T3 is always exact fp32")
    ap.add_argument("--no-alt-precisions", action="store_true",
                    help="skip the extra steps measured after the timed region at the other S3Gen precisions (default: bf16x3 is measured "
                         "and reported under audio_s_per_wall_s_at_other_precisions; --all-precisions adds exact fp32)")
    ap.add_argument("--all-precisions", action="store_true")
    ap.add_argument("--no-fast-mode", action="store_true", help="skip the full fast-mode measurement (S3Gen bf16x3 + T3 bf16 decode weights)")
    ap.add_argument("--no-streaming", action="store_true", help="skip the chunked-synthesis latency measurement after the timed region")
    ap.add_argument("--config3", action="store_true",
                    help="after the timed region also run configs[3]: 256 utterances strong-sharded over the ranks (32 per GPU at 8 GPUs); "
                         "default on when WORLD_SIZE == 8")
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--no-autotune", action="store_true",
                    help="keep the built-in decode-step geometry instead of measuring the candidates on this GPU first (T3Engine.autotune: child "
                         "process, bit-identical candidates only)")
    ap.add_argument("--no-parity", action="store_true", help="skip the GPU-vs-CPU parity block (needs the CPU baseline utterance)")
    ap.add_argument("--force-rccl", action="store_true",
                    help="initialise the RCCL process group even at world size 1 and run C1 (broadcast of the Conditionals) and C2 (gather of the "
                         "waveforms) through it on device tensors: what a single GPU can exercise of the multi-GPU path (tests/test_rccl_world1_gpu.py)")
    ap.add_argument("--selftest-rendezvous", action="store_true",
                    help="launcher / collective self-test WITHOUT kernels (gloo, host tensors): the N ranks rendezvous, C1 (broadcast of the "
                         "Conditionals) and C2 (gather of waveforms) fire on synthetic payloads of the benched shapes, rank 0 prints the JSON "
                         "line with value = null.  Used by tests/test_bench_launcher.py on the CPU; it measures nothing")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 1 if args.workload == "vc60" else 8
    return args


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed environment: re-run this very command line under
    `python -m torch.distributed.run` (one process per GPU, rendezvous on 127.0.0.1) and pass its exit code on.  When the
    driver (or a user) already launched the ranks, WORLD_SIZE is set and this is a no-op."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, args.gpus))))
    return subprocess.call(cmd, env=env)


def selftest_rendezvous(args, rank, world):
    """No kernels, no GPU: exercises exactly the multi-rank plumbing of main() -- init, C1, per-rank work list, barrier-bracketed
    timed region with max-over-ranks, C2 inside it, one JSON line from rank 0 -- on host tensors over gloo."""
    import torch.distributed as dist
    from chatterbox_amd import dist as cdist, synth
    t3c, gen = (synth.t3_cond(), synth.s3gen_ref()) if rank == 0 else (None, None)
    t3c, gen = cdist.broadcast_conditionals(t3c, gen, src=0)                                     # C1
    c1_ok = bool(torch.equal(t3c["cond_prompt_speech_tokens"], synth.t3_cond()["cond_prompt_speech_tokens"]))
    B, N = args.batch, args.tokens
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    got = 0
    for _ in range(args.steps):
        wavs = [torch.full(((N - 1) * 960,), float(rank * B + b)) for b in range(B)]
        allw = cdist.gather_waveforms(wavs, dst=0)                                                # C2
        if rank == 0:
            assert len(allw) == B * world and all(float(w[0]) == i for i, w in enumerate(allw)), "C2: wrong order / count"
            got += len(allw)
    # the throughput schedule's collective order: batches posted to the background gatherer by a generator that is fed by a second host thread
    gat = cdist.AsyncGatherer(dst=0)
    jobs = [dict(text_tokens=[None] * B) for _ in range(args.steps)]
    a_p, lat_p = pipelined_batches(_StubPipelineEngine(rank, (N - 1) * 960), jobs, gat)
    per_batch = gat.close()
    pipe_ok = len(lat_p) == args.steps
    if rank == 0:
        pipe_ok &= len(per_batch) == args.steps
        for k, allw in enumerate(per_batch):
            want = [float(1000 * r + 10 * k + b) for r in range(world) for b in range(B)]
            pipe_ok &= allw is not None and [float(w[0]) for w in allw] == want and all(w.numel() == (N - 1) * 960 + 16 * (i % B) for i, w in enumerate(allw))
    else:
        pipe_ok &= all(x is None for x in per_batch) or world == 1
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        okt = torch.tensor([int(pipe_ok)])
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        pipe_ok = bool(int(okt[0]))
    if rank == 0:
        print(json.dumps({"metric": "audio-sec/wall-sec (xRT) + p50 first-audio latency, Multilingual-V3 500M", "value": None,
                          "unit": "audio-s/wall-s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * float(el[0]) / max(1, args.steps), 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "none",
                          "selftest": {"what": "rendezvous + C1 + C2 only (gloo, host tensors): NO kernels ran, nothing is measured",
                                       "c1_ok": c1_ok, "c2_waveforms_gathered": got, "ranks": world,
                                       "pipelined_c2_ok": bool(pipe_ok), "pipelined_batches": len(lat_p)},
                          "config": {"workload": "selftest", "global_batch": B * world, "parallelism": f"dp{world}"}}), flush=True)


def pipelined_batches(eng, jobs, gatherer, **kw):
    """The throughput schedule over `jobs` (engine.synthesize_pipelined): -> (audio seconds of this rank, per-batch latencies).  C2 is POSTED per batch to
    `gatherer` (dist.AsyncGatherer: a background thread runs the collectives in batch order) instead of blocking the generator between batches; the caller
    closes the gatherer before the barrier that ends its timed region."""
    a, ls = 0.0, []
    for host, st_, lat in eng.synthesize_pipelined(jobs, **kw):
        a += sum(w.numel() for w in host) / 24000.0
        ls.append(lat)
        gatherer.post(host)  # C2, off the critical path
    return a, ls


class _StubPipelineEngine:
    """Host-only stand-in for ChatterboxEngine.synthesize_pipelined with the same threading shape (a worker thread produces batches, the generator
    yields them in order): what tests/test_bench_launcher.py drives through pipelined_batches + AsyncGatherer over gloo.  Computes nothing."""

    def __init__(self, rank, samples):
        self.rank, self.samples = rank, samples

    def synthesize_pipelined(self, jobs, **kw):
        import queue
        import threading
        q = queue.Queue()

        def worker():
            for k, job in enumerate(jobs):
                time.sleep(0.002 * ((k + self.rank) % 3))  # ranks finish their batches at different times
                q.put([torch.full((self.samples + 16 * b,), float(1000 * self.rank + 10 * k + b)) for b in range(len(job["text_tokens"]))])
        th = threading.Thread(target=worker, daemon=True)
        th.start()
        for k, job in enumerate(jobs):
            t0 = time.perf_counter()
            yield q.get(), None, time.perf_counter() - t0
        th.join()


def cpu_baseline(t3_sd, s3_sd, args, n_layers):
    """ONE utterance of the benched workload (same text length, same number of speech tokens, same prompts) on the host cores:
    the UNMODIFIED reference modules when /root/reference is present (kind "reference"; never the case on the GPU box), else the
    oracle (kind "port": oracle/ref_torch.py, a CPU fp32 restatement pinned against the reference by tests/golden).  The reference
    cannot batch, so B > 1 on the CPU is a serial loop of such calls: xRT is the same for any B."""
    from chatterbox_amd import synth
    from oracle import ref_import, ref_torch as O
    n = args.cpu_tokens or args.tokens
    torch.set_num_threads(max(1, min(args.cpu_threads, os.cpu_count() or 1)))  # tiny decode matmuls crawl on 256 threads
    tt = synth.text_tokens(args.text_tokens)
    u = synth.rand((n,), seed=7)
    ref = synth.s3gen_ref()
    T = 2 * (250 + n)
    z, noise, phase = synth.randn((1, 80, T), seed=5), synth.randn((1, 9, 960 * n), seed=6), torch.zeros(1, 9, 1)
    kind = "reference" if ref_import.available() else "port"
    with torch.inference_mode():
        if kind == "reference":
            t0, t1, t2, toks, mel, wav = _cpu_reference(ref_import, O, t3_sd, s3_sd, n_layers, tt, u, ref, z, noise, phase, n)
        else:
            t0 = time.perf_counter()
            toks = O.t3_inference(t3_sd, n_layers, synth.t3_cond(), torch.stack([tt, tt]), n, u, ban_eos=True)
            t1 = time.perf_counter()
            wav, mel = O.s3gen_inference(s3_sd, toks.clamp(max=6560)[None], torch.tensor([n]), ref, z, phase, noise, 10)
            t2 = time.perf_counter()
            toks, mel, wav = toks.view(-1), mel[0].t(), wav[0]
    audio_s = n / 25.0
    artefacts = dict(tt=tt, u=u, ref=ref, z=z, noise=noise, phase=phase, n=n, tokens=toks.view(-1)[:n].clone(), mel=mel.clone(), wav=wav.view(-1).clone())
    return artefacts, dict(value=round(audio_s / (t2 - t0), 4), unit="audio-s/wall-s", cores=torch.get_num_threads(), kind=kind,
                sample=f"1 utterance of the benched workload: {args.text_tokens} text tokens, {n} speech tokens ({audio_s:.1f} s audio), 10 s voice "
                       f"prompt, 10-step CFG CFM; T3 {t1 - t0:.1f} s + S3Gen/HiFT {t2 - t1:.1f} s on {torch.get_num_threads()} threads "
                       f"(the reference cannot batch: B > 1 = serial loop, same xRT)")


def _cpu_reference(ref_import, O, t3_sd, s3_sd, n_layers, tt, u, ref, z, noise, phase, n):
    """Time the unmodified reference (T3.inference -> S3Token2Wav.flow_inference -> hift_inference) with injected RNG."""
    import torch.distributions.uniform as U
    from chatterbox_amd import synth
    T3, T3Config, T3Cond, lc = ref_import.load_T3()
    lc.LLAMA_CONFIGS["Llama_520M"]["num_hidden_layers"] = n_layers
    m = T3(T3Config.multilingual()).eval()
    m.load_state_dict(t3_sd, strict=True)
    S3 = ref_import.load_S3Gen()
    g = S3().eval()
    g.load_state_dict(s3_sd, strict=False)
    ci = synth.t3_cond()
    cond = T3Cond(speaker_emb=ci["speaker_emb"], cond_prompt_speech_tokens=ci["cond_prompt_speech_tokens"], emotion_adv=ci["emotion_adv"])
    step = [0]
    o_mn, o_rl, o_us = torch.multinomial, torch.randn_like, U.Uniform.sample

    def fake_multinomial(probs, num_samples=1, **kw):
        pr = probs[0].clone()
        pr[O.STOP_SPEECH] = 0.0
        tok = O.sample_inverse_cdf(pr, u[step[0]])
        step[0] += 1
        return torch.tensor([[tok]])

    try:
        torch.multinomial = fake_multinomial
        t0 = time.perf_counter()
        toks = m.inference(t3_cond=cond, text_tokens=torch.stack([tt, tt]), max_new_tokens=n, temperature=0.8, cfg_weight=0.5,
                           repetition_penalty=1.2, min_p=0.05, top_p=1.0)
        t1 = time.perf_counter()
        torch.randn_like = lambda t, **kw: z.clone()
        mel = g.flow_inference(toks[:, :n].clamp(max=6560), ref_dict=dict(ref), n_cfm_timesteps=10, finalize=True)
        torch.randn_like = lambda t, **kw: noise.clone() if t.shape == noise.shape else torch.zeros_like(t)
        U.Uniform.sample = lambda self, sample_shape=torch.Size(): phase.clone()
        out = g.hift_inference(mel)
        t2 = time.perf_counter()
    finally:
        torch.multinomial, torch.randn_like, U.Uniform.sample = o_mn, o_rl, o_us
    wav = out[0] if isinstance(out, (tuple, list)) else out
    wav = wav.view(1, -1).clone()
    wav[:, : len(g.trim_fade)] *= g.trim_fade  # S3Token2Wav.inference applies it after hift_inference (s3gen.py:360); outside the timed span
    return t0, t1, t2, toks.view(-1)[:n], mel[0].t(), wav.view(-1)


def gpu_parity(eng, art, kind):
    """BASELINE.md section 4, "parity in the same run": the GPU synthesises the very utterance the CPU baseline just produced (same text,
    voice, uniforms, CFM noise z, vocoder phase / noise) and the line carries how far the two are apart: sampled tokens identical, mel L1
    (mean |d| over the generated frames; fp32 tolerance 5e-6 at T = 1000, tests/test_baseline_shapes_gpu.py) and waveform RMSE."""
    from chatterbox_amd import synth
    dev, n = eng.dev, art["n"]
    samp = dict(temperature=0.8, cfg_weight=0.5, repetition_penalty=1.2, min_p=0.05, top_p=1.0)
    toks = eng.t3.generate(synth.t3_cond(), [art["tt"]], max_new_tokens=n, uniforms=art["u"][None].to(dev), ban_eos=True, **samp)[0].cpu().view(-1)[:n]
    equal = bool(torch.equal(toks, art["tokens"].cpu()))
    # flow + vocoder on the CPU's own tokens, so that a token difference (if any) does not hide the S3Gen comparison
    st = art["tokens"].clamp(max=6560)
    wavs, mel = eng.vocode([st], art["ref"], z=art["z"].transpose(1, 2).contiguous().to(dev), phase=art["phase"], noise=art["noise"],
                           n_cfm_timesteps=10)
    m_gpu, m_cpu = mel[0].float().cpu(), art["mel"].float()
    T = min(m_gpu.shape[0], m_cpu.shape[0])
    w_gpu, w_cpu = wavs[0].float().cpu().view(-1), art["wav"].float().view(-1)
    L = min(w_gpu.numel(), w_cpu.numel())
    return dict(vs=kind, tokens_equal=equal, n_tokens=int(n), mel_l1=float((m_gpu[:T] - m_cpu[:T]).abs().mean()),
                mel_max_abs=float((m_gpu[:T] - m_cpu[:T]).abs().max()), wav_rmse=float((w_gpu[:L] - w_cpu[:L]).pow(2).mean().sqrt()),
                wav_peak=float(w_cpu[:L].abs().max()), frames=int(T), samples=int(L),
                note="one utterance of the benched workload, identical injected randomness on both sides; S3Gen compared on the CPU's tokens")


def pmc_traffic(kernel_substr, source="flow_only"):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/: separate FETCH_SIZE and WRITE_SIZE runs of
    scripts/flow_only.py resp. scripts/prof_t3_eager.py -- PMC cannot be collected inside the timed run, and rocprofv3 --pmc
    does not survive hipGraph replays).  FETCH_SIZE is doubled: on gfx950 it reports half of the bytes of a 16-B-per-lane
    coalesced read (MI355X_MICROARCH.md, HBM section); units are KiB."""
    import csv
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = next((q for q in (os.path.join(ROOT, "profiles", f"{r}_{source}_pmc_{c}.csv") for r in ("r06", "r05", "r04", "r03", "r02", "r01")) if os.path.exists(q)), None)
        if f is None:
            return None, None
        used = os.path.basename(f)[:3]
        n = 0
        for r in csv.DictReader(open(f)):
            subs = kernel_substr if isinstance(kernel_substr, tuple) else (kernel_substr,)
            if any(k in r["kernel"] for k in subs) and r["counter"] == c:  # several template instances: launch-weighted mean
                tot[c] = tot.get(c, 0.0) + float(r["sum"]) * 1024.0
                n += int(r["launches"])
        if n:
            tot[c] /= n
    if len(tot) != 2:
        return None, None
    return 2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"], (f"profiles/{used}_{source}_pmc_{{FETCH,WRITE}}_SIZE.csv (2*FETCH_SIZE + WRITE_SIZE per launch; "
                                                          "a committed rocprofv3 --pmc pass of the same kernels, not re-measured in this run)")


def roofline_entries(summ, elapsed, steps, timed_steps, s3_prec, n_decode, gemv, gemv_kind="llama"):
    """One roofline object per timed kernel class.  GEMM / attention: every launch of the last timed step, HIP events on the launch
    stream.  gemv: the decode step's projections replayed from a hipGraph (the form they run in inside the timed region, where
    events cannot see individual graph nodes) -- one graph per projection type sweeping the 30 layers' weights, events around it."""
    out = {}
    nprod = {3: 3, 6: 6, 16: 3}.get(s3_prec, 1)
    pname = {3: "bf16x3", 6: "bf16x6", 16: "f16x3"}.get(s3_prec, "")
    names = {"gemm_f32": ("gemm_f32_kernel (exact fp32 MFMA implicit GEMM: T3 prefill, shapes the split kernel does not serve)",
                          "gemm_f32_kernel", 1),
             "gemm_split": ("gemm_split_kernel (implicit-GEMM linear/conv of S3Gen, %s)" % pname, "gemm_split_kernel", nprod),
             "flash_attn_f32": (("flash_attn_split_kernel (%s)" % pname) if nprod > 1 else "flash_attn_f32_kernel",
                                "flash_attn_split_kernel" if nprod > 1 else "flash_attn_f32_kernel", nprod),
             "gemm_planes": ("gemm_pl_kernel (plane-format f16x3 implicit GEMM of the CFM estimator: fp16 planes DMA'd global -> LDS, no operand "
                             "conversion)", "gemm_pl_kernel", 3),
             "flash_attn_planes": ("flash_attn_pl2_kernel (plane-format f16x3 attention of the CFM transformer blocks)", "flash_attn_pl", 3)}
    for kind, (kname, sub, npr) in names.items():
        ks = summ.get(kind)
        if not ks or ks["ms"] <= 0:
            continue
        tf = ks["flops"] / (ks["ms"] * 1e-3) / 1e12
        if kind == "flash_attn_f32" and nprod > 1:
            # T3 prefill attention stays exact; it is < 1 % of the attention FLOPs of a step and is folded into this line
            pass
        peak = MFMA_BF16_PEAK_TFLOPS if npr > 1 else MFMA_F32_PEAK_TFLOPS
        issued = tf * npr
        e = dict(bound="mfma", kernel=kname, achieved=round(issued, 2), peak=peak, unit="TFLOP/s", frac=round(issued / peak, 4),
                 traffic=None, launches=ks["launches"], avg_launch_us=round(1e3 * ks["ms"] / ks["launches"], 2),
                 flops_per_launch=round(ks["flops"] / ks["launches"], 0),
                 algorithmic_bytes_per_launch=round(ks["bytes"] / ks["launches"], 0),
                 share_of_step=round(ks["ms"] * 1e-3 / timed_steps / (elapsed / steps), 3))
        if npr > 1:
            e["fp32_equivalent_tflops"] = round(tf, 2)
            e["note"] = (f"achieved = algorithmic fp32 FLOPs x {npr} 16-bit MFMA products per fp32 product (the work the matrix cores "
                         f"execute), priced against the dense bf16 / fp16 peak; {round(tf, 1)} TFLOP/s fp32-equivalent = "
                         f"{round(tf / MFMA_F32_PEAK_TFLOPS, 2)}x the exact-fp32 MFMA peak.  f16x3 issues half the products of bf16x6 for the "
                         f"same fp32-level result, so its fraction of the 16-bit peak is lower while its time per launch is 1.45x shorter")
        tr, src = pmc_traffic(sub, "t3_eager" if kind == "gemm_f32" else "flow_only")
        e["traffic"], e["traffic_source"] = (round(tr, 0) if tr else None), src
        out[kind] = e
    if gemv:
        tot_ms = sum(v["ms"] for v in gemv.values())
        tot_b = sum(v["bytes"] for v in gemv.values())
        tot_n = sum(v["launches"] for v in gemv.values())
        per_step_ms = sum(v["ms"] / v["launches"] * v["per_step"] for v in gemv.values())
        gbs = tot_b / (tot_ms * 1e-3) / 1e9
        gname = ("gemv_kernel / gemv_ct_kernel (T3 decode weight streaming: q/k/v, o, gate|up, down projections; M = 2*batch rows)" if gemv_kind == "llama" else
                 "gemv_row_kernel (GPT-2 T3 decode at batch 1: c_attn, attention c_proj, c_fc, mlp c_proj on row-major weights, one activation row)")
        e = dict(bound="hbm", kernel=gname,
                 achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4), traffic=None,
                 launches=tot_n, avg_launch_us=round(1e3 * tot_ms / tot_n, 2), algorithmic_bytes_per_launch=round(tot_b / tot_n, 0),
                 share_of_step=round(per_step_ms * 1e-3 * n_decode / (elapsed / steps), 3),
                 by_projection={k: dict(avg_launch_us=round(1e3 * v["ms"] / v["launches"], 2),
                                        GBps=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                                        bytes_per_launch=round(v["bytes"] / v["launches"], 0)) for k, v in gemv.items()},
                 note="average launch duration INCLUDING the dependent-launch boundary (~1.2 us): events bracket hipGraph replays of "
                      "30 back-to-back launches (one per layer, 1 GB of distinct weights per sweep, so nothing is served from the "
                      "256 MB Infinity Cache); rocprofv3's per-kernel average excludes that boundary")
        tr, src = pmc_traffic(("gemv_kernel", "gemv_ct_kernel"), "t3_eager") if gemv_kind == "llama" else pmc_traffic(("gemv_row_kernel",), "turbo_eager")
        e["traffic"], e["traffic_source"] = (round(tr, 0) if tr else None), src
        out["gemv_f32"] = e
    return out


@torch.inference_mode()
def gemv_sweeps(t3, rows, reps=6):
    """HIP-event timing of the decode projections in the form they run in: hipGraph replays.  One graph per projection type, each
    sweeping the 30 layers (distinct weights -> cold HBM streams, as in a real decode step)."""
    from chatterbox_amd import ops
    dev, tn = t3.dev, t3.tune
    f = lambda *s: torch.randn(*s, device=dev)
    if t3.decode_mode == "v2" and rows <= 16:  # packed operands, RMSNorm / residual / partial sums folded into the GEMVs
        r16 = (rows + 15) // 16 * 16
        x, x2, att, g = f(r16, t3.D), f(r16, t3.D), f(r16, t3.D), f(r16, t3.F)
        qkv, gg = torch.empty(rows, 3 * t3.D, device=dev), torch.empty(r16, t3.F, device=dev)
        dks = tn["d_ks2"]
        t3._prepare_tune()
        qtc, odtc = t3._tiles()  # output columns per workgroup of the q/k/v resp. o / down projections (T3Engine.tune, CBX_T3_TUNE)
        qt, ht = (0 if qtc == 16 else qtc), (0 if odtc == 16 else odtc)
        pd = f(max(dks, 1), r16, t3.D) * 0.1
        pk = dict(w_packed=True, x_packed=True, M=rows, flags=t3._gf())  # the adopted geometry's GEMV flags (cbx_gemv_t.flags)
        red = dict(xpart=pd, x_out=x2) if dks > 1 else {}
        qks, qct = t3._qks(), int(tn.get("qkv_ct") or 3)
        qparts, qssq = torch.empty(max(qks, 1), rows, 3 * t3.D, device=dev), torch.zeros(max(qks, 1), 16, device=dev)
        calls = {"qkv": (lambda lw: ops.gemv(x, lw["wqkv_pk"], qparts, N=3 * t3.D, K=t3.D, nw=8, norm_w=lw["ln1"], col_tiles=qct, ksplit=qks, ssq_out=qssq, **red, **pk)) if qks > 1
                 else (lambda lw: ops.gemv(x, t3._image(lw, "wqkv", qtc), qkv, N=3 * t3.D, K=t3.D, nw=8, norm_w=lw["ln1"], half_tile=qt, **red, **pk)),
                 "o": lambda lw: ops.gemv(att, t3._image(lw, "wo", odtc), x2, N=t3.D, K=t3.D, nw=tn["o_nw2"], res=x2, out_packed=True,
                                          half_tile=ht, **pk),
                 "gate_up": lambda lw: ops.gemv(x, lw["wgu_pk"], gg, N=t3.F, K=t3.D, swiglu=True, nw=tn["gu_nw"], norm_w=lw["ln2"],
                                                out_packed=True, **pk),
                 "down": (lambda lw: ops.gemv(g, t3._image(lw, "wd", odtc), pd, N=t3.D, K=t3.F, ksplit=dks, nw=tn["d_nw2"], out_packed=True,
                                              half_tile=ht, **pk)) if dks > 1
                 else (lambda lw: ops.gemv(g, t3._image(lw, "wd", odtc), x2, N=t3.D, K=t3.F, nw=tn["d_nw2"], res=x2, out_packed=True, half_tile=ht, **pk))}
        wbytes = {"qkv": lambda lw: lw["wqkv_pk"].numel() * 4, "o": lambda lw: lw["wo_pk"].numel() * 4,
                  "gate_up": lambda lw: lw["wgu_pk"].numel() * 4, "down": lambda lw: lw["wd_pk"].numel() * 4}
    else:
        h, att, g = f(rows, t3.D), f(rows, t3.D), f(rows, t3.F)
        qkv, gg = torch.empty(rows, 3 * t3.D, device=dev), torch.empty(rows, t3.F, device=dev)
        po, pd = torch.empty(tn["o_ks"], rows, t3.D, device=dev), torch.empty(tn["d_ks"], rows, t3.D, device=dev)
        calls = {"qkv": lambda lw: ops.gemv(h, lw["wqkv"], qkv, nw=tn["qkv_nw"]),
                 "o": lambda lw: ops.gemv(att, lw["wo"], po, ksplit=tn["o_ks"], nw=4),
                 "gate_up": lambda lw: ops.gemv(h, lw["wgu"], gg, swiglu=True, nw=tn["gu_nw"]),
                 "down": lambda lw: ops.gemv(g, lw["wd"], pd, ksplit=tn["d_ks"], nw=4)}
        wbytes = {"qkv": lambda lw: lw["wqkv"].numel() * 4, "o": lambda lw: lw["wo"].numel() * 4,
                  "gate_up": lambda lw: lw["wgu"].numel() * 4, "down": lambda lw: lw["wd"].numel() * 4}
    res = {}
    side = torch.cuda.Stream(device=dev)
    for name, fn in calls.items():
        with torch.cuda.stream(side):
            for lw in t3.layers:  # warm-up (also makes sure nothing lazy happens under capture)
                fn(lw)
        torch.cuda.synchronize()
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph):
            for lw in t3.layers:
                fn(lw)
        gph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            gph.replay()
        e1.record()
        torch.cuda.synchronize()
        res[name] = dict(ms=e0.elapsed_time(e1), launches=reps * len(t3.layers), per_step=len(t3.layers),
                         bytes=float(reps * sum(wbytes[name](lw) for lw in t3.layers)))
    return res


@torch.inference_mode()
def gemv_row_sweeps(t3, reps=6):
    """The batch-1 GPT-2 decode projections (ops.gemv_row) timed the way gemv_sweeps times the Llama ones: one hipGraph per projection type sweeping the layers'
    (distinct) weights, HIP events around `reps` replays."""
    from chatterbox_amd import ops
    dev, D = t3.dev, t3.D
    f = lambda *s: torch.randn(*s, device=dev)
    x, g = f(D), f(4 * D)
    qkv, out, gg = torch.empty(3 * D, device=dev), torch.zeros(D, device=dev), torch.empty(4 * D, device=dev)
    parts = torch.zeros(t3.H, max(1, int(t3.tune.get("row_splits") or 8)), ops.ATTN_PART_REC, device=dev)
    parts[:, :, 1] = 1.0
    calls = {"c_attn": lambda lw: ops.gemv_row(x, lw["wqkv"], qkv, bias=lw["bqkv"], ln=lw["ln1"]),
             "c_proj": lambda lw: ops.gemv_row(None, lw["wo"], out, bias=lw["bo"], res=out, parts=parts),
             "c_fc": lambda lw: ops.gemv_row(x, lw["wfc"], gg, bias=lw["bfc"], ln=lw["ln2"], act=ops.GELU_TANH),
             "mlp_c_proj": lambda lw: ops.gemv_row(g, lw["wpr"], out, bias=lw["bpr"], res=out)}
    wname = {"c_attn": "wqkv", "c_proj": "wo", "c_fc": "wfc", "mlp_c_proj": "wpr"}
    res = {}
    side = torch.cuda.Stream(device=dev)
    for name, fn in calls.items():
        with torch.cuda.stream(side):
            for lw in t3.layers:
                fn(lw)
        torch.cuda.synchronize()
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph):
            for lw in t3.layers:
                fn(lw)
        gph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            gph.replay()
        e1.record()
        torch.cuda.synchronize()
        res[name] = dict(ms=e0.elapsed_time(e1), launches=reps * len(t3.layers), per_step=len(t3.layers),
                         bytes=float(reps * sum(lw[wname[name]].numel() * 4 for lw in t3.layers)))
    return res


def decode_step_entry(t3, n_layers, gpt2=False):
    """Whole decode step (every kernel of the captured hipGraph: GEMVs + attention + sampler), timed INSIDE the timed region with two
    HIP events around the replay loop of each generate() call.  Algorithmic bytes per step: all streamed weights
    (4 N K per projection + head) + the KV cache read of every row at its current context (2 * ctx * 1024 * 4 B per layer per row).
    gpt2: the Turbo / Nano T3 (c_attn 3 D^2, c_proj D^2, c_fc and mlp c_proj 4 D^2 each; no CFG: one row per utterance)."""
    if not t3.decode_events:
        return None
    torch.cuda.synchronize()
    n_layers = n_layers or t3.L
    w_bytes = 4.0 * (n_layers * 12 * t3.D * t3.D + t3.V * t3.D) if gpt2 else 4.0 * (n_layers * (4 * t3.D * t3.D + 3 * t3.D * t3.F) + t3.V * t3.D)
    ms = steps = kv = 0.0
    for e0, e1, n, s0, rows in t3.decode_events:
        ms += e0.elapsed_time(e1)
        steps += n
        B = rows if gpt2 else rows // 2
        for b in range(B):  # every row of utterance b (both CFG rows of the Llama T3) reads ctx = s0 + t keys at decode step t = 1..n
            kv += (1 if gpt2 else 2) * sum(2.0 * (s0[b] + t) * t3.D * 4 * n_layers for t in range(1, n + 1))
    tot = w_bytes * steps + kv
    gbs = tot / (ms * 1e-3) / 1e9
    return dict(bound="hbm", what="T3 decode step = one hipGraph replay (5 launches per layer + head + sampler), timed with HIP events inside "
                "the timed region", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                frac_of_measured_copy_peak=round(gbs / 6290.0, 4), steps=int(steps), ms_per_step=round(ms / steps, 4),
                weight_bytes_per_step=round(w_bytes, 0), kv_bytes_per_step_mean=round(kv / steps, 0))


def _on_green(t3):
    """Is the decode geometry the timed region ran on the committed allow-list of hardware-verified geometries (chatterbox_amd/decode_green.json)?"""
    from chatterbox_amd import autotune as at
    return at.canon(t3.tune, t3.knobs) in at.green_variants()



def vc60_main(args, dev, rank, world):
    """configs[4] (BASELINE.json): the voice-conversion path of example_vc.py / vc.py:83-104 on long-form audio -- S3 tokenizer on the 16 kHz source,
    S3Gen (encoder + 10-step CFG CFM) with the target voice's 10 s prompt, HiFT -- `--batch` utterances of `--vc-seconds` per step in ONE device
    batch.  60 s = 1500 speech tokens + the 250-token prompt = T 3500 mel frames per CFM row (2 rows per utterance with CFG): attention is
    3.5x heavier per frame than in the headline config.  No T3 on this path.  Prints its own JSON line (NOT the headline metric's config)."""
    import torch.distributed as dist
    from chatterbox_amd import dist as cdist, ops, synth
    from chatterbox_amd.api import ChatterboxVC
    B = args.batch
    t_build = time.perf_counter()
    vc = ChatterboxVC.from_synthetic(dev, seed=0)
    eng, tok, ref = vc.engine, vc.analyzer.tokenizer, vc.ref_dict
    if args.s3gen_precision is not None:
        eng.flow.precision = eng.hift.precision = args.s3gen_precision
    build_s = time.perf_counter() - t_build
    L16 = int(args.vc_seconds * 16000)
    g = torch.Generator().manual_seed(31 + rank)
    # synthetic "speech": band-limited noise with a syllable-rate envelope (the tokenizer's arithmetic does not depend on the content)
    t = torch.arange(L16) / 16000.0
    src = [(0.1 * torch.randn(L16, generator=g) * (0.6 + 0.4 * torch.sin(2 * math.pi * (3.0 + b) * t))).float() for b in range(B)]
    gd = torch.Generator(device=dev).manual_seed(77 + rank)

    def one_step():
        t0 = time.perf_counter()
        toks = [tok(w)[0].view(-1).long().cpu() for w in src]  # S3 tokenizer per utterance (25 tokens / s)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        T = 2 * (int(ref["prompt_token"].shape[1]) + max(int(x.numel()) for x in toks))
        z = torch.randn(len(toks), T, 80, generator=gd, device=dev)
        wavs, _ = eng.vocode(toks, ref, z=z)
        host = [w.cpu() for w in wavs]
        t2 = time.perf_counter()
        tm = dict(tokenizer_s=t1 - t0, **eng.last_timing)
        cdist.gather_waveforms(wavs, dst=0)
        return sum(w.numel() for w in host) / 24000.0, t2 - t0, tm, T

    for _ in range(args.warmup):
        one_step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    audio, lats, stage, T = 0.0, [], {}, 0
    for _ in range(args.steps):
        a, lat, tm, T = one_step()
        audio += a
        lats.append(lat)
        for k, v in tm.items():
            stage[k] = stage.get(k, 0.0) + v
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    stats = torch.tensor([elapsed, audio], dtype=torch.float64, device=dev)
    if world > 1:
        mx, sm = stats.clone(), stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, audio = float(mx[0]), float(sm[1])
    timer = ops.KernelTimer(["gemm_f32", "gemm_split", "flash_attn_f32", "gemm_planes", "flash_attn_planes"])
    ops.TIMER = timer  # the instrumented extra step, outside the timed region (per-kernel HIP events; the Python launch sequencing)
    one_step()
    torch.cuda.synchronize()
    ops.TIMER = None
    if rank != 0:
        return
    roofs = roofline_entries(timer.summary(), elapsed, args.steps, 1, eng.flow.precision, 0, None)
    dom = max(roofs, key=lambda k: roofs[k]["share_of_step"]) if roofs else None
    roof = roofs.pop(dom, None)
    for e in [roof] + list(roofs.values()):
        if e:  # the committed PMC passes are of the headline shape (T = 1000): not this workload's traffic
            e["traffic"], e["traffic_source"] = None, "not collected at this shape (the committed PMC passes are of the headline workload)"
    lats.sort()
    print(json.dumps({
        "metric": "audio-sec/wall-sec (xRT), voice conversion of long-form audio (configs[4]; NOT the headline metric's config)",
        "value": round(audio / elapsed, 3), "unit": "audio-s/wall-s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "stage_ms": {k[:-2]: round(1e3 * v / args.steps, 1) for k, v in stage.items()},
        "dtype": "f32 (S3 tokenizer: exact fp32 MFMA; S3Gen: f16x3 planes = 22 significand bits per operand, fp32 accumulate)" if eng.flow.precision == 16
                 else f"f32 (S3Gen precision mode {eng.flow.precision})",
        "data": "synthetic (seeded random-init weights in the reference checkpoint layout; synthetic 16 kHz source audio, synthetic 10 s target-voice prompt)",
        "p50_latency_ms": round(1e3 * lats[len(lats) // 2], 1),
        "config": {"workload": (f"configs[4]: voice conversion (example_vc.py): S3 tokenizer -> S3Gen 10-step CFG CFM -> HiFT, {args.vc_seconds:.0f} s utterances "
                                f"({T // 2 - int(ref['prompt_token'].shape[1])} tokens + {int(ref['prompt_token'].shape[1])}-token prompt: T = {T} mel frames per CFM row), "
                                f"batch {B}/GPU in one device batch"),
                   "global_batch": B * world, "parallelism": f"dp{world}", "model_build_s": round(build_s, 1),
                   "stage_seams": {"flow": bool(eng.flow.c_seam), "hift": bool(eng.hift.c_seam)}},
        "roofline": roof, "roofline_secondary": list(roofs.values())}), flush=True)


def _co_resident_on_green(t3):
    """Is the decode geometry of the throughput schedule (T3Engine.co_resident) on the committed allow-list of hardware-verified geometries?"""
    from chatterbox_amd import autotune as at
    return at.canon(dict(t3.tune, half_tiles=0, d_ks2=4, d_nw2=8), t3.knobs) in at.green_variants()


def _range_trips():
    from chatterbox_amd import engine
    return int(engine.RANGE_TRIPS)


def log(msg):
    if os.environ.get("CBX_BENCH_VERBOSE"):
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or run `python bench.py "
                 f"--gpus {args.gpus}` without a torch.distributed environment and it spawns the ranks itself)")
    import torch.distributed as dist
    if args.selftest_rendezvous:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
        selftest_rendezvous(args, rank, world)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if world > 1 or args.force_rccl:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(_free_port())
        os.environ.setdefault("RANK", "0"), os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        assert dist.get_world_size() == args.gpus
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from chatterbox_amd import dist as cdist, ops, synth
    from chatterbox_amd.engine import ChatterboxEngine, TurboEngine

    if args.workload == "vc60":
        vc60_main(args, dev, rank, world)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    t_build = time.perf_counter()
    turbo = args.workload != "mtl"
    if turbo:
        dm, nl = (768, 12) if args.workload == "nano" else (1024, 24)
        t3_sd = synth.t3_turbo_state_dict(nl, dm, 0)
        s3_sd = synth.s3gen_state_dict(0, meanflow=True)
        eng = TurboEngine(t3_sd, s3_sd, dev)
        args.no_cpu_baseline = True
    else:
        t3_sd = synth.t3_state_dict(args.t3_layers, 0)
        s3_sd = synth.s3gen_state_dict(0)
        eng = ChatterboxEngine(t3_sd, s3_sd, dev, n_t3_layers=args.t3_layers)
    if args.s3gen_precision is not None:
        eng.flow.precision = eng.hift.precision = args.s3gen_precision
    s3_prec = eng.flow.precision
    build_s = time.perf_counter() - t_build
    log(f"model built in {build_s:.1f}s")
    tune_rep = None  # T3 decode-step geometry picked by measurement on THIS GPU before anything is timed (below, once the workload exists)
    # C1: rank 0 "analysed the voice prompt"; everybody else receives the packed Conditionals over RCCL
    t3c, gen = (synth.t3_cond(prompt_len=375 if turbo else 150), synth.s3gen_ref()) if rank == 0 else (None, None)
    t3c, gen = cdist.broadcast_conditionals(t3c, gen, src=0, device=dev, force=args.force_rccl)

    B, N = args.batch, args.tokens
    texts = [(synth.turbo_text_tokens if turbo else synth.text_tokens)(args.text_tokens, seed=100 * rank + b) for b in range(B)]
    T = 2 * (gen["prompt_token"].shape[1] + N)

    if not turbo and not args.no_autotune and 2 * B <= 16:
        # chatterbox_amd/autotune.py: the candidates run in a child process (a faulting one cannot take the bench down); a geometry whose logits
        # are bit-identical to the built-in one's is adopted as is; the report goes into the JSON line (config.t3_decode_autotune).
        # the fastest candidate overall may sum the down projection in another (valid fp32) order: it is kept only if ALL B x N tokens of the
        # benched workload come out as the built-in geometry samples them (whose utterance 0 the parity block checks against the CPU reference)
        t_tune = time.perf_counter()
        gv = torch.Generator(device=dev).manual_seed(777 + rank)
        kwv = dict(max_new_tokens=N, uniforms=torch.rand(B, N, generator=gv, device=dev), ban_eos=True, ban_from=6561)
        want = [t.tolist() for t in eng.t3.generate(t3c, texts, **kwv)]
        # green_only: nothing off the committed allow-list of hardware-verified geometries (chatterbox_amd/decode_green.json) is even timed
        tune_rep = eng.t3.autotune(B=B, ctx=34 + args.text_tokens + 2 + N // 2, log=log, green_only=True,
                                   validate=lambda: [t.tolist() for t in eng.t3.generate(t3c, texts, **kwv)] == want)
        tune_rep["autotune_s"] = round(time.perf_counter() - t_tune, 1)
        log(f"decode autotune: adopted {tune_rep.get('adopted')} in {tune_rep['autotune_s']} s")

    def one_step(seed):
        g = torch.Generator(device=dev).manual_seed(1234 + 1000 * rank + seed)
        u = torch.rand(B, N, generator=g, device=dev)
        z = torch.randn(B, T, 80, generator=g, device=dev)
        t0 = time.perf_counter()
        if turbo:
            wavs, st = eng.synthesize(texts, t3c, gen, max_gen_len=N - 1, uniforms=u, ban_eos=True, ban_from=6561)
        else:
            wavs, st = eng.synthesize(texts, t3c, gen, max_new_tokens=N, uniforms=u, ban_eos=True, ban_from=6561, z=z,
                                      drop_last_token=True)
        log(f"step seed={seed}: {eng.last_timing}")
        host = [w.cpu() for w in wavs]  # first (and only) audio reaches the host here: the path is non-streaming
        lat = time.perf_counter() - t0
        allw = cdist.gather_waveforms(wavs, dst=0, force=args.force_rccl)  # C2
        if args.force_rccl:
            assert len(allw) == len(host) and all(torch.equal(a, h) for a, h in zip(allw, host)), "C2 over RCCL returned other waveforms"
        return sum(w.numel() for w in host) / 24000.0, lat, dict(eng.last_timing)

    def pipelined_run(n, seed0, precision=None):
        """n batches through the throughput schedule; C2 per batch through the background gatherer (closed before the caller's barrier); -> (audio seconds, per-batch latencies)"""
        jobs = []
        for i in range(n):
            g = torch.Generator(device=dev).manual_seed(1234 + 1000 * rank + seed0 + i)
            jobs.append(dict(text_tokens=texts, t3_conds=t3c, gen_ref=gen, uniforms=torch.rand(B, N, generator=g, device=dev),
                             z=torch.randn(B, T, 80, generator=g, device=dev)))
        gat = cdist.AsyncGatherer(dst=0, force=args.force_rccl)
        try:
            return pipelined_batches(eng, jobs, gat, max_new_tokens=N, ban_eos=True, ban_from=6561, drop_last_token=True)
        finally:
            gat.close()

    def serial_run(n, seed0, stage):
        a, ls = 0.0, []
        for i in range(n):
            ai, lat, tm = one_step(seed0 + i)
            a += ai
            ls.append(lat)
            for k, v in tm.items():
                stage[k] = stage.get(k, 0.0) + v
        return a, ls

    def timed(fn):
        """barrier + synchronize on both sides (the driver's contract)"""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        return time.perf_counter() - t0, r

    schedule = "serial" if (turbo or args.serial or (args.schedule == "serial" and not args.pipelined)) else "pipelined"
    explicit_pipelined = args.pipelined or args.schedule == "pipelined"
    pipelined = schedule == "pipelined"
    fallback = None
    # HIP-event timing of every GEMM / attention launch costs ~25 ms per step (4000 event records) and sends the flow / vocoder through their
    # per-kernel Python sequencing: it runs in ONE EXTRA serial step AFTER the timed region (round 5; rounds 1-4 instrumented the last timed step), so
    # the headline number carries none of it.  The decode-step roofline stays in-run: two HIP events per generate() call around the graph replays.
    timer = ops.KernelTimer(["gemm_f32", "gemm_split", "flash_attn_f32", "gemm_planes", "flash_attn_planes"])
    timed_steps = 1
    ops.TIMER = None
    eng.t3.time_decode, eng.t3.decode_events = True, []
    stage, other = {}, None
    if pipelined:
        try:
            pipelined_run(max(2, args.warmup), -1000)
            eng.t3.decode_events = []
            elapsed, (audio, lats) = timed(lambda: pipelined_run(args.steps, 0))
        except Exception as e:  # the headline must never be lost to the throughput schedule: fall back to the serial one and say so
            if world > 1 or explicit_pipelined:  # (a rank-local fallback would desynchronise the C2 collectives of the other ranks; an explicit request is not second-guessed)
                raise
            fallback = f"{type(e).__name__}: {e}"[:300]
            log(f"pipelined schedule failed ({fallback}): falling back to the serial schedule")
            torch.cuda.synchronize()
            pipelined, schedule = False, "serial"
    pipe_events = list(eng.t3.decode_events)
    eng.t3.decode_events = []
    if pipelined:
        # the serial (latency) schedule on the same box, outside the timed region: its own warm-up (the decode graph of the default geometry is captured
        # again), then min(K, 5) steps -- stage split, in-run decode-step roofline and the serial headline of rounds 1-4 come from here
        one_step(-1)
        eng.t3.decode_events = []
        n_ser = max(1, min(args.steps, 5))
        ser_elapsed, (ser_audio, ser_lats) = timed(lambda: serial_run(n_ser, 0, stage))
        ser_stats = torch.tensor([ser_elapsed, ser_audio], dtype=torch.float64, device=dev)
        if world > 1:
            mx, sm = ser_stats.clone(), ser_stats.clone()
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            ser_elapsed, ser_audio = float(mx[0]), float(sm[1])
        ser_lats.sort()
        other = dict(schedule="serial: the batches strictly one after the other (the latency schedule; the headline of rounds 1-4), measured after the timed region",
                     value=round(ser_audio / ser_elapsed, 3), unit="audio-s/wall-s", steps=n_ser, ms_per_step=round(1e3 * ser_elapsed / n_ser, 2),
                     p50_first_audio_latency_ms=round(1e3 * ser_lats[len(ser_lats) // 2], 1))
        step_s, stage_n = ser_elapsed / n_ser, n_ser  # what kernel shares / stage times are relative to
    else:
        for i in range(args.warmup):
            one_step(-1 - i)
        eng.t3.decode_events = []
        elapsed, (audio, lats) = timed(lambda: serial_run(args.steps, 0, stage))
        step_s, stage_n = elapsed / args.steps, args.steps
    eng.t3.time_decode = False
    ops.TIMER = timer  # the instrumented step (every rank: one_step contains the C2 collective)
    one_step(args.steps)
    torch.cuda.synchronize()
    ops.TIMER = None

    stats = torch.tensor([elapsed, audio], dtype=torch.float64, device=dev)
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, audio = float(mx[0]), float(sm[1])

    # ---- outside the timed region: (a) eager replay of the decode step for the gemv roofline, (b) one step at each of the other
    # S3Gen precisions so that the exact-fp32 figure is reported by the same run
    alt, gemv, dstep, cfg3, stream = {}, None, None, None, None
    run_cfg3 = (args.config3 or world == 8) and not turbo
    if run_cfg3:  # configs[3]: 256 utterances, contiguous shards (dist.shard_range), 32 per GPU at 8 GPUs; every rank takes part
        lo, hi = cdist.shard_range(256, rank, world)
        tx = [synth.text_tokens(args.text_tokens, seed=1000 + i) for i in range(lo, hi)]
        gcfg = torch.Generator(device=dev).manual_seed(4321 + rank)

        def cfg3_step():
            nb = len(tx)
            u3 = torch.rand(nb, N, generator=gcfg, device=dev)
            audio3 = 0.0
            for c0 in range(0, nb, 32):  # device batches of <= 32 utterances (64 decode rows)
                c1 = min(nb, c0 + 32)
                z3 = torch.randn(c1 - c0, T, 80, generator=gcfg, device=dev)
                w3, _ = eng.synthesize(tx[c0:c1], t3c, gen, max_new_tokens=N, uniforms=u3[c0:c1], ban_eos=True, ban_from=6561, z=z3,
                                       drop_last_token=True)
                audio3 += sum(w.numel() for w in w3) / 24000.0
                cdist.gather_waveforms(w3, dst=0)
            return audio3

        cfg3_step()  # warm-up (KV cache / graph for the 64-row shape)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        tc = time.perf_counter()
        a3 = cfg3_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        st3 = torch.tensor([time.perf_counter() - tc, a3], dtype=torch.float64, device=dev)
        if world > 1:
            mx3, sm3 = st3.clone(), st3.clone()
            dist.all_reduce(mx3, op=dist.ReduceOp.MAX)
            dist.all_reduce(sm3, op=dist.ReduceOp.SUM)
            st3 = torch.stack([mx3[0], sm3[1]])
        cfg3 = dict(workload="configs[3]: 256 fixed-length utterances strong-sharded over the ranks (contiguous blocks)",
                    utterances=256, per_gpu=hi - lo, device_batch=min(32, hi - lo), scaling="strong",
                    value=round(float(st3[1]) / float(st3[0]), 2), unit="audio-s/wall-s", wall_s=round(float(st3[0]), 3))
    if rank == 0:
        summ = timer.summary()
        if turbo:
            dstep = decode_step_entry(eng.t3, None, gpt2=True)
            if B == 1 and eng.t3.tune.get("row_path"):
                gemv = gemv_row_sweeps(eng.t3)
        dstep_pipe = bf16x6 = repeat_ms = None
        if not turbo:
            dstep = decode_step_entry(eng.t3, args.t3_layers)  # the serial steps' events (the decode step with the GPU to itself)
            if pipelined and pipe_events:  # ... and the same step beside the co-resident flow kernels of the previous batch (the timed region)
                keep, eng.t3.decode_events = eng.t3.decode_events, pipe_events
                dp = decode_step_entry(eng.t3, args.t3_layers)
                eng.t3.decode_events = keep
                dstep_pipe = dict(ms_per_step=dp["ms_per_step"], frac=dp["frac"], steps=dp["steps"],
                                  note="the decode step inside the timed region: its workgroups share the CUs with the flow + vocoder of the previous batch")
            gemv = gemv_sweeps(eng.t3, 2 * B)
            if pipelined and s3_prec != 6 and not args.no_alt_precisions and world == 1:
                # the THROUGHPUT schedule at strictly fp32-width S3Gen arithmetic (bf16x6: three bf16 planes = all 24 significand bits per operand, fp32
                # exponent range; T3 is exact fp32 in every mode): same schedule, same batches, after the timed region
                eng.flow.precision = eng.hift.precision = 6
                try:
                    pipelined_run(2, -2000)
                    n6 = max(3, min(args.steps, 6))
                    e6, (a6, l6) = timed(lambda: pipelined_run(n6, 500))
                    l6.sort()
                    bf16x6 = dict(value=round(a6 / e6, 3), steps=n6, ms_per_step=round(1e3 * e6 / n6, 2), p50_first_audio_latency_ms=round(1e3 * l6[len(l6) // 2], 1))
                except Exception as e:
                    bf16x6 = dict(value=None, error=f"{type(e).__name__}: {e}"[:200])
                    torch.cuda.synchronize()
                eng.flow.precision = eng.hift.precision = s3_prec
            if not args.no_alt_precisions and world == 1:  # one_step() contains the C2 collective: single-rank runs only
                for pr in ((1, 16, 6, 3) if args.all_precisions else (16, 6, 3)):
                    if pr == s3_prec:
                        continue
                    eng.flow.precision = eng.hift.precision = pr
                    one_step(-100)
                    torch.cuda.synchronize()
                    ta = time.perf_counter()
                    a, _, tm_alt = one_step(-101)
                    torch.cuda.synchronize()
                    alt[f"s3gen_precision_{pr}"] = round(a / (time.perf_counter() - ta), 2)
                    if pr == 6:  # what ONE fp16-range trip of the default mode costs: the flow + vocoder of the batch again, at bf16x6
                        repeat_ms = round(1e3 * (tm_alt.get("flow_s", 0.0) + tm_alt.get("hift_s", 0.0)), 1)
                if not args.no_fast_mode:  # the full opt-in fast mode: S3Gen bf16x3 + T3 decode weights rounded to bf16 (both narrower than fp32)
                    from chatterbox_amd.t3 import T3Engine
                    t3_fp32 = eng.t3
                    eng.t3 = T3Engine(t3_sd, dev, n_layers=args.t3_layers, weight_dtype="bf16")
                    eng.flow.precision = eng.hift.precision = 3
                    one_step(-100)
                    torch.cuda.synchronize()
                    ta = time.perf_counter()
                    a, _, _ = one_step(-101)
                    torch.cuda.synchronize()
                    alt["fast_mode"] = round(a / (time.perf_counter() - ta), 2)
                    eng.t3 = t3_fp32
                eng.flow.precision = eng.hift.precision = s3_prec
        stream = None
        if not turbo and world == 1 and not args.no_streaming:
            # chunked synthesis (engine.synthesize_stream): wall time until the first audio chunk is on the host, and the cost of the
            # whole chunked run, on the benched batch (first chunk = 1 s of audio, then 2 s chunks)
            def stream_run(first_alone):
                fl, tl = [], []
                for rep in range(4):
                    g = torch.Generator(device=dev).manual_seed(99 + rep)
                    us = torch.rand(B, N, generator=g, device=dev)
                    torch.cuda.synchronize()
                    ts = time.perf_counter()
                    first = None
                    for r in eng.synthesize_stream(texts, t3c, gen, first_chunk=12, chunk=100, chunk_growth=1.35, first_alone=first_alone, max_new_tokens=N, uniforms=us,
                                                   ban_eos=True, ban_from=6561):
                        if first is None:
                            first = time.perf_counter() - ts
                    torch.cuda.synchronize()
                    if rep:  # (the first run of a variant captures its graphs)
                        fl.append(first)
                        tl.append(time.perf_counter() - ts)
                fl.sort(), tl.sort()
                return dict(p50_first_audio_latency_ms=round(1e3 * fl[1], 1), p50_total_ms=round(1e3 * tl[1], 1), audio_s_per_wall_s=round(B * (N - 1) / 25.0 / tl[1], 2))

            stream = dict(schedule="three rounds over 15 / 115 / 250 tokens (first chunk 12 tokens = 0.5 s of audio + 3 lookahead, then 100, then the rest), each round = encoder + "
                                   "CFM + vocoder over the tokens so far; T3 keeps decoding on its own stream beside rounds 1 and 2 and -- latency first -- NOT beside round 0 "
                                   "(engine.synthesize_stream, overlap=True, first_alone=True)", **stream_run(True))
            # the same schedule with T3 decoding beside round 0 as well: more throughput, later first audio (same-box A/B: profiles/r06_streaming_schedules_ab.jsonl)
            stream["t3_beside_first_round"] = stream_run(False)
        pipe_extra = None
        if not turbo and world == 1 and not args.no_streaming and not pipelined and fallback is None:
            # the throughput schedule on the same workload, outside the timed region: T3 of batch k + 1 on a high-priority stream beside flow + vocoder
            # of batch k (engine.synthesize_pipelined; same kernels, same results, about twice the per-batch latency) -- 6 batches, fill and drain included
            try:  # (an extra: it must never cost the bench line)
                jobs = [dict(text_tokens=texts, t3_conds=t3c, gen_ref=gen) for _ in range(6)]
                for _ in eng.synthesize_pipelined(jobs[:2], max_new_tokens=N, ban_eos=True, ban_from=6561, drop_last_token=True):
                    pass
                torch.cuda.synchronize()
                tp, ap, lp = time.perf_counter(), 0.0, []
                for host, st_, lat in eng.synthesize_pipelined(jobs, max_new_tokens=N, ban_eos=True, ban_from=6561, drop_last_token=True):
                    ap += sum(w.numel() for w in host) / 24000.0
                    lp.append(lat)
                torch.cuda.synchronize()
                lp.sort()
                pipe_extra = dict(schedule="pipelined: T3(k+1) on a high-priority stream overlaps flow + HiFT(k); 6 batches incl. fill and drain",
                                  audio_s_per_wall_s=round(ap / (time.perf_counter() - tp), 2), p50_batch_latency_ms=round(1e3 * lp[len(lp) // 2], 1))
            except Exception as e:
                pipe_extra = dict(error=f"{type(e).__name__}: {e}"[:200])
                torch.cuda.synchronize()
        roofs = roofline_entries(summ, step_s, 1, timed_steps, s3_prec, N - 1, gemv, "gpt2" if turbo else "llama")  # shares relative to a SERIAL step (kernels with the GPU to themselves)
        dom = args.roofline_kernel
        if dom == "auto":
            dom = max(roofs, key=lambda k: roofs[k]["share_of_step"]) if roofs else None
        roof = roofs.pop(dom, None)
        lats.sort()
        out = {
            "metric": "audio-sec/wall-sec (xRT) + p50 first-audio latency, Multilingual-V3 500M",
            "value": round(audio / elapsed, 3), "unit": "audio-s/wall-s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # T3 / flow / HiFT wall ms per step (host clocks around each stage of the serial schedule; the same numbers as config.stage_ms_per_step)
            "stage_ms": {k[:-2]: round(1e3 * v / stage_n, 1) for k, v in stage.items()},
            "schedule": schedule,
            "dtype": "f32" if s3_prec == 1 else
                     ("f32 (T3: exact fp32 MFMA; S3Gen: every fp32 operand = 2 fp16 planes h + l/2048 = 22 significand bits, 3 fp16 MFMA products "
                      "per fp32 product in two fp32 accumulators: error at or below the exact-fp32 MFMA path; operand range checked on the "
                      "device, bf16x6 repeat otherwise)" if s3_prec == 16 else
                      "f32 (T3: exact fp32 MFMA; S3Gen: every fp32 operand = 3 bf16 planes = all 24 significand bits, 6 MFMA products per fp32 product, "
                      "fp32 accumulate: error at or below the exact-fp32 MFMA path)" if s3_prec == 6 else
                      "f32 operands, bf16x3 FAST MODE for S3Gen (2 bf16 planes, 3 MFMA products: 16 significand bits per operand; narrower than the "
                      "reference's fp32 -- not the headline configuration)"),
            "data": "synthetic (seeded random-init weights in the reference checkpoint layout; synthetic prompts)",
            # latency of the HEADLINE schedule: enqueue of a batch's T3 -> its audio on the host (one-shot synthesis; the pipelined schedule holds two batches
            # in flight, so a batch waits for its predecessor's flow + vocoder).  The latency modes are beside it: other_schedule (serial one-shot) and streaming
            "p50_first_audio_latency_ms": round(1e3 * lats[len(lats) // 2], 1),
            # both schedules and the precision variants as SHORT top-level keys (ADVICE r05 / VERDICT r05 items 2, 3): `value` is value_<schedule>
            "value_pipelined": round(audio / elapsed, 3) if pipelined else (pipe_extra or {}).get("audio_s_per_wall_s"),
            "value_serial": other["value"] if other else round(audio / elapsed, 3),
            "p50_first_audio_latency_ms_pipelined": round(1e3 * lats[len(lats) // 2], 1) if pipelined else (pipe_extra or {}).get("p50_batch_latency_ms"),
            "p50_first_audio_latency_ms_serial": other["p50_first_audio_latency_ms"] if other else round(1e3 * lats[len(lats) // 2], 1),
            "p50_first_audio_latency_ms_streaming": (stream or {}).get("p50_first_audio_latency_ms"),
            # the throughput schedule with S3Gen at strictly fp32-width operands (bf16x6: 24 significand bits, fp32 exponent range) -- THE fp32-width claim; `value` runs
            # S3Gen at f16x3 (22 significand bits per operand, range-checked, error measured at or below the exact-fp32 MFMA's: DESIGN.md section 1)
            "value_bf16x6": (bf16x6 or {}).get("value") if not turbo else None,
            # the fp16 range check of the default S3Gen mode: passes of THIS run (warm-up, timed region, extras) that tripped it and were repeated at bf16x6, and
            # what one such repeat costs per batch (serial flow + vocoder at bf16x6).  Seeded random-init weights: 0 trips; trip rate against activation outliers
            # of 1e2 .. 1e5: tests/test_models_gpu.py::test_f16x3_range_flag_trip_rate_and_repeat_cost_on_outlier_activations
            "f16x3_range_trips": _range_trips(), "bf16x6_repeat_cost_ms": repeat_ms,
            "config": {"workload": (f"configs[2]: Chatterbox-Multilingual-V3 500M architecture (T3 Llama-520M {args.t3_layers}L + S3Gen 10-step CFG CFM + "
                                    f"HiFT), batch {B}/GPU, {args.text_tokens} text tokens, {N} speech tokens = {N / 25:.0f} s audio per utterance, "
                                    f"10 s voice prompt") if not turbo else
                                   (f"configs[{1 if args.workload == 'turbo' else 0}] architecture: Chatterbox-{args.workload} (GPT-2 T3, 2-step meanflow S3Gen, HiFT), "
                                    f"batch {B}/GPU, {N} speech tokens (NOT the headline metric's config)"),
                       "global_batch": B * world, "parallelism": f"dp{world}",
                       "stage_ms_per_step": {k: round(1e3 * v / stage_n, 1) for k, v in stage.items()}, "model_build_s": round(build_s, 1),
                       "schedule": ("pipelined (the throughput schedule, engine.synthesize_pipelined): T3 decode of batch k + 1 on a high-priority HIP stream beside "
                                    "the CFM + vocoder of batch k on a second one; both stages on their CO-RESIDENT kernel forms (one workgroup per CU that leaves half "
                                    "of the register file free: DESIGN 6.000), the T3 launches enqueued by a second host thread; K batches incl. fill and drain "
                                    "inside the timed region; same work and results as the serial schedule (tests/test_models_gpu.py::test_pipelined_equals_serial); "
                                    "stage_ms / roofline / decode_step are of the SERIAL steps measured after the timed region (other_schedule)"
                                    if pipelined else "serial: the K batches strictly one after the other" + (f" (FALLBACK: the pipelined schedule failed: {fallback})" if fallback else "")),
                       # which stages ran through their stage-level C entry point (cbx_t3_prefill + cbx_t3_decode_step / cbx_cfm_solve / cbx_hift_decode)
                       "stage_seams": {"t3": bool(getattr(eng.t3, "c_step", False)), "flow": bool(eng.flow.c_seam), "hift": bool(eng.hift.c_seam)}},
            # the T3 decode geometry the timed region ran (built-in + what the autotuner adopted from the hardware-green allow-list)
            "t3_geometry": ({"adopted": (tune_rep or {}).get("adopted") or {}, "tune": {k: v for k, v in eng.t3.tune.items() if v != type(eng.t3)._TUNE.get(k)},
                             "knobs": dict(eng.t3.knobs), "on_green_list": _on_green(eng.t3)} if not turbo else {"tune": dict(eng.t3.tune), "knobs": dict(eng.t3.knobs)}),
            # ... and the geometry its decode chains ran in the throughput schedule (T3Engine.co_resident: every launch <= 8 waves x <= 128 VGPRs; the tune part is a
            # member of the allow-list, the shallow SwiGLU batches are an op-level identity: tests/test_ops_gpu.py::test_gemv_packed_rms_fused)
            "t3_geometry_throughput_schedule": ({"tune": {"half_tiles": 0, "d_ks2": 4, "d_nw2": 8}, "knobs": {"shallow": 1}, "t3_batches_in_flight": 2,
                                                  "on_green_list": _co_resident_on_green(eng.t3)} if (pipelined and not turbo) else None),
            "multi_gpu_note": ("single-GPU run: no N > 1 scaling curve exists in this repo (the driver owns multi-GPU leases)" if world == 1 else None),
            "roofline": roof,
            "decode_step": dstep,
            "roofline_secondary": list(roofs.values()),
        }
        if tune_rep is not None:
            out["config"]["t3_decode_autotune"] = {
                "adopted": tune_rep.get("adopted") or {}, "error": tune_rep.get("error"), "seconds": tune_rep.get("autotune_s"),
                "best_bit_identical": tune_rep.get("best"), "best_overall": tune_rep.get("best_any"),
                "best_overall_tokens_equal_builtin": tune_rep.get("best_any_validated"),
                "ms_per_token": {"builtin": tune_rep.get("baseline_ms_per_token"), "best_bit_identical": tune_rep.get("ms_per_token"),
                                 "best_overall": tune_rep.get("ms_per_token_any")},
                "rule": "candidates timed in a child process (hipGraph replays of the whole token step, synthetic state, >= 1 % faster, confirmed back "
                        f"to back); a bit-identical one is adopted as is; one that 'reorders' (another fp32 summation order) only if all {B} x {N} "
                        "tokens of the benched batch equal the built-in geometry's; ONLY candidates on the committed allow-list of hardware-green "
                        "geometries (chatterbox_amd/decode_green.json) are timed; identity = logits of single steps over ragged contexts "
                        "{1, 38, 63, 64, 65, 225, 640} + the timed run's final logits",
                "candidates": [{k: r[k] for k in ("variant", "ms_per_token", "identical", "reorders", "valid", "error", "confirm", "skipped") if k in r}
                               for r in tune_rep.get("candidates", [])]}
        if alt:
            labels = {"s3gen_precision_3": "s3gen_bf16x3_fast_mode (narrower than the reference's fp32; bf16-mode tolerances)",
                      "s3gen_precision_1": "s3gen_exact_fp32_mfma", "s3gen_precision_6": "s3gen_bf16x6 (fp32-level, fp32 exponent range)",
                      "s3gen_precision_16": "s3gen_f16x3 (fp32-level)",
                      "fast_mode": "full_fast_mode: s3gen bf16x3 + T3 decode weights rounded to bf16 (NOT fp32 parity: tokens differ from the reference's)"}
            out["audio_s_per_wall_s_at_other_precisions"] = {labels.get(k, k): v for k, v in alt.items()}
        if cfg3:
            out["configs3"] = cfg3
        if stream:
            out["streaming"] = stream
        if pipe_extra:
            out["pipelined_schedule"] = pipe_extra
        if other:
            out["other_schedule"] = other
        if not turbo and bf16x6:
            out["throughput_schedule_bf16x6"] = bf16x6
        if dstep_pipe:
            out["decode_step_in_throughput_schedule"] = dstep_pipe
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only (rank 0's host cores)
            log("cpu baseline ...")
            art, out["cpu_baseline"] = cpu_baseline(t3_sd, s3_sd, args, args.t3_layers)
            if not args.no_parity:
                out["parity"] = gpu_parity(eng, art, out["cpu_baseline"]["kind"])
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
