#!/usr/bin/env python
"""bench.py -- headline metric of BASELINE.json: audio-seconds per wall-second (xRT) + p50 first-audio latency of
generate() = T3 -> S3Gen (10-step CFM, CFG) -> HiFT, Multilingual-V3 500M architecture, batch 8 per GPU.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the whole hot path over one batch of synthetic utterances: 64 text tokens,
250 speech tokens (10 s of audio, EOS banned so the length is fixed), 150-token T3 voice prompt, 250-token / 500-frame
S3Gen prompt, seeded random-init weights in the reference's checkpoint layout (no network => no pretrained weights).
Prints ONE JSON line (rank 0).  `roofline` is the dominant kernel timed with HIP events inside the timed region;
`cpu_baseline` is the oracle (CPU port of the reference path) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, exact fp32
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="utterances per GPU per step")
    ap.add_argument("--tokens", type=int, default=250, help="speech tokens per utterance (25/s)")
    ap.add_argument("--text-tokens", type=int, default=64)
    ap.add_argument("--t3-layers", type=int, default=30)
    ap.add_argument("--workload", default="mtl", choices=["mtl", "turbo", "nano"],
                    help="mtl = configs[2] (the headline metric); turbo / nano = configs[1] / configs[0] architectures (GPT-2 T3, 2-step meanflow)")
    ap.add_argument("--serial", action="store_true",
                    help="run the K steps strictly one after the other (default: T3 of batch k+1 overlaps the CFM/vocoder of batch k "
                         "on a second HIP stream -- same work, same results, a serving loop's steady state)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tokens", type=int, default=25, help="speech tokens of the bounded CPU-baseline sample")
    ap.add_argument("--roofline-kernel", default="gemm_f32", choices=["gemm_f32", "flash_attn_f32", "gemv_f32"],
                    help="kernel class timed with HIP events for the roofline object (default: the dominant one by time)")
    ap.add_argument("--cpu-threads", type=int, default=32)
    return ap.parse_args()


def cpu_baseline(t3_sd, s3_sd, args, n_layers):
    """The oracle (oracle/ref_torch.py, a CPU fp32 port of the reference path validated against the reference itself)
    on ONE utterance with `--cpu-tokens` speech tokens, same prompts; B>1 on the reference is a serial loop of such calls."""
    from chatterbox_amd import synth
    from oracle import ref_torch as O
    n = args.cpu_tokens
    torch.set_num_threads(max(1, min(args.cpu_threads, os.cpu_count() or 1)))  # tiny decode matmuls crawl on 256 threads
    tt = synth.text_tokens(args.text_tokens)
    with torch.inference_mode():
        t0 = time.perf_counter()
        toks = O.t3_inference(t3_sd, n_layers, synth.t3_cond(), torch.stack([tt, tt]), n, synth.rand((n,), seed=7), ban_eos=True)
        t1 = time.perf_counter()
        toks = toks.clamp(max=6560)
        ref = synth.s3gen_ref()
        T = 2 * (250 + n)
        phase = torch.zeros(1, 9, 1)
        wav, _ = O.s3gen_inference(s3_sd, toks[None], torch.tensor([n]), ref, synth.randn((1, 80, T), seed=5), phase,
                                   synth.randn((1, 9, 960 * n), seed=6), 10)
        t2 = time.perf_counter()
    audio_s = n / 25.0
    return dict(value=round(audio_s / (t2 - t0), 4), unit="audio-s/wall-s", cores=torch.get_num_threads(), kind="port",
                sample=f"1 utterance, {args.text_tokens} text tokens, {n} speech tokens ({audio_s:.1f} s audio), 10 s voice prompt, "
                       f"T3 {t1 - t0:.1f} s + S3Gen/HiFT {t2 - t1:.1f} s on {torch.get_num_threads()} threads "
                       f"(reference cannot batch: B>1 = serial loop)")


def log(msg):
    if os.environ.get("CBX_BENCH_VERBOSE"):
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from chatterbox_amd import dist as cdist, ops, synth
    from chatterbox_amd.engine import ChatterboxEngine, TurboEngine

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
    build_s = time.perf_counter() - t_build
    log(f"model built in {build_s:.1f}s")

    # C1: rank 0 "analysed the voice prompt"; everybody else receives the packed Conditionals over RCCL
    t3c, gen = (synth.t3_cond(prompt_len=375 if turbo else 150), synth.s3gen_ref()) if rank == 0 else (None, None)
    t3c, gen = cdist.broadcast_conditionals(t3c, gen, src=0, device=dev)

    B, N = args.batch, args.tokens
    texts = [(synth.turbo_text_tokens if turbo else synth.text_tokens)(args.text_tokens, seed=100 * rank + b) for b in range(B)]
    T = 2 * (gen["prompt_token"].shape[1] + N)

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
        allw = cdist.gather_waveforms(wavs, dst=0)  # C2
        return sum(w.numel() for w in host) / 24000.0, lat, dict(eng.last_timing)

    for i in range(args.warmup):
        one_step(-1 - i)
    pipelined = not args.serial and not turbo
    timer = ops.KernelTimer([args.roofline_kernel])
    ops.TIMER = timer
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    audio, lats, stage = 0.0, [], {}
    if pipelined:
        jobs = [dict(text_tokens=texts, t3_conds=t3c, gen_ref=gen) for _ in range(args.steps)]
        for host, st, lat in eng.synthesize_pipelined(jobs, max_new_tokens=N, ban_eos=True, ban_from=6561, drop_last_token=True):
            audio += sum(w.numel() for w in host) / 24000.0
            lats.append(lat)
            cdist.gather_waveforms(host, dst=0)  # C2
    else:
        for i in range(args.steps):
            a, lat, tm = one_step(i)
            audio += a
            lats.append(lat)
            for k, v in tm.items():
                stage[k] = stage.get(k, 0.0) + v
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ops.TIMER = None

    stats = torch.tensor([elapsed, audio], dtype=torch.float64, device=dev)
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, audio = float(mx[0]), float(sm[1])

    if rank == 0:
        ks = timer.summary().get(args.roofline_kernel)
        roof = None
        if ks and ks["ms"] > 0 and args.roofline_kernel == "gemv_f32":
            gbs = ks["bytes"] / (ks["ms"] * 1e-3) / 1e9
            roof = dict(bound="hbm", kernel="gemv_kernel (decode weight streaming; eager launches only, graph replays are not event-timed)",
                        achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4), traffic=None,
                        launches=ks["launches"], avg_launch_us=round(1e3 * ks["ms"] / ks["launches"], 2),
                        bytes_per_launch=round(ks["bytes"] / ks["launches"], 0))
        elif ks and ks["ms"] > 0:
            tf = ks["flops"] / (ks["ms"] * 1e-3) / 1e12
            kname = {"gemm_f32": "gemm_f32_kernel (implicit-GEMM linear/conv, every launch with M > 32)",
                     "flash_attn_f32": "flash_attn_f32_kernel"}[args.roofline_kernel]
            roof = dict(bound="mfma", kernel=kname, achieved=round(tf, 2), peak=MFMA_F32_PEAK_TFLOPS,
                        unit="TFLOP/s", frac=round(tf / MFMA_F32_PEAK_TFLOPS, 4), traffic=None, launches=ks["launches"],
                        avg_launch_us=round(1e3 * ks["ms"] / ks["launches"], 2),
                        flops_per_launch=round(ks["flops"] / ks["launches"], 0),
                        share_of_step=round(ks["ms"] * 1e-3 / elapsed, 3))
        lats.sort()
        out = {
            "metric": "audio-sec/wall-sec (xRT) + p50 first-audio latency, Multilingual-V3 500M",
            "value": round(audio / elapsed, 3), "unit": "audio-s/wall-s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (seeded random-init weights in the reference checkpoint layout; synthetic prompts)",
            "p50_first_audio_latency_ms": round(1e3 * lats[len(lats) // 2], 1),
            "config": {"workload": (f"configs[2]: Chatterbox-Multilingual-V3 500M architecture (T3 Llama-520M {args.t3_layers}L + S3Gen 10-step CFG CFM + "
                                    f"HiFT), batch {B}/GPU, {args.text_tokens} text tokens, {N} speech tokens = {N / 25:.0f} s audio per utterance, "
                                    f"10 s voice prompt") if not turbo else
                                   (f"configs[{1 if args.workload == 'turbo' else 0}] architecture: Chatterbox-{args.workload} (GPT-2 T3, 2-step meanflow S3Gen, HiFT), "
                                    f"batch {B}/GPU, {N} speech tokens (NOT the headline metric's config)"),
                       "global_batch": B * world, "parallelism": f"dp{world}",
                       "stage_ms_per_step": {k: round(1e3 * v / args.steps, 1) for k, v in stage.items()}, "model_build_s": round(build_s, 1),
                       "schedule": "pipelined: T3(k+1) on a high-priority stream overlaps flow+HiFT(k)" if pipelined else "serial"},
            "roofline": roof,
        }
        if not args.no_cpu_baseline:
            log("cpu baseline ...")
            out["cpu_baseline"] = cpu_baseline(t3_sd, s3_sd, args, args.t3_layers)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
