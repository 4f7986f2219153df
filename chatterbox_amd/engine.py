"""generate() = T3.inference -> S3Gen.flow_inference -> HiFT.inference for a BATCH of utterances on one MI355X.

Mirrors the body of `ChatterboxMultilingualTTS.generate` (reference mtl_tts.py:324-352) after text tokenisation:
speech tokens are stripped of SOS/EOS (`drop_invalid_tokens`, s3tokenizer/__init__.py:16-30), the flow + vocoder run
on the padded batch, and each waveform is cut to its own length (the multilingual path also drops the last
token's 40 ms, mtl_tts.py:348-352).
"""
import os
import time
import warnings

import torch

from . import ops
from .hift import HiFTEngine
from .s3gen import FlowEngine
from .t3 import T3Engine, START_SPEECH, STOP_SPEECH

SPEECH_VOCAB = 6561
SAMPLES_PER_TOKEN = 960  # 24 kHz / 25 tokens per second


RANGE_TRIPS = 0  # S3Gen passes of this process that met an operand outside the fp16 range and were repeated at bf16x6 (bench.py reports it)


def _range_checked(eng, run, check=True):
    """S3Gen's default numerics (precision 16 = f16x3) need operands inside the fp16 range; a launch that meets one outside it raises a
    device flag (ops.enable_range_flag).  Then the result is not meaningful and the work is repeated at bf16x6, which has the fp32
    exponent range.  check=False: the caller looks at ops.range_flag_tripped() itself at its next synchronisation point."""
    out = run()
    if check and 16 in (eng.flow.precision, eng.hift.precision) and ops.range_flag_tripped():
        warnings.warn("an S3Gen operand exceeded the fp16 range: repeating flow matching + vocoder at bf16x6")
        global RANGE_TRIPS
        RANGE_TRIPS += 1
        saved = eng.flow.precision, eng.hift.precision
        eng.flow.precision, eng.hift.precision = (6 if p == 16 else p for p in saved)
        try:
            out = run()
        finally:
            eng.flow.precision, eng.hift.precision = saved
    return out


def drop_invalid_tokens(x):
    """Strip [SOS ... EOS) and ids outside the S3 codebook (reference s3tokenizer/__init__.py:16-30, tts.py:260-262)."""
    x = x.view(-1)
    sos = (x == START_SPEECH).nonzero()
    s = int(sos[0]) + 1 if len(sos) else 0
    eos = (x == STOP_SPEECH).nonzero()
    e = int(eos[0]) if len(eos) else len(x)
    x = x[s:e]
    return x[x < SPEECH_VOCAB]


class ChatterboxEngine:
    def __init__(self, t3_sd, s3gen_sd, device="cuda", n_t3_layers=None, meanflow=False, t3_weights=None):
        self.dev = torch.device(device)
        self.t3_weights = t3_weights  # None / "fp32" (parity path) or "bf16" (opt-in: decode weight images rounded to bf16)
        self.t3 = self._build_t3(t3_sd, n_t3_layers)
        self.flow = FlowEngine(s3gen_sd, self.dev, meanflow=meanflow)
        self.hift = HiFTEngine(s3gen_sd, self.dev)
        self.last_timing = {}

    def _build_t3(self, t3_sd, n_layers):
        """CBX_PACK_CACHE=<dir>: keep the packed device layout of the T3 weights (4 GB: row-major + lane-ordered GEMV images) on disk,
        keyed by a fingerprint of the checkpoint, and map it straight to the device on later starts (formats.py)."""
        cache = os.environ.get("CBX_PACK_CACHE")
        if not cache:
            return T3Engine(t3_sd, self.dev, n_layers=n_layers, weight_dtype=self.t3_weights)
        from . import formats
        fp = formats.fingerprint(t3_sd) + f"-L{n_layers}"
        path = os.path.join(cache, f"t3_{fp}.cbxpack")
        kind = formats.packed_kind("t3-llama-" + os.environ.get("CBX_T3_DECODE", "v2") + "-" + (self.t3_weights or os.environ.get("CBX_T3_WEIGHTS", "fp32"))
                                   + f"-ht{int(bool(T3Engine._TUNE.get('half_tiles')))}")  # the half-tile images are part of the packed set
        t = formats.load_packed(path, fp, kind)
        if t is not None:
            return T3Engine.from_packed(t, self.dev)
        eng = T3Engine(t3_sd, self.dev, n_layers=n_layers, weight_dtype=self.t3_weights)
        os.makedirs(cache, exist_ok=True)
        formats.save_packed(eng.export_packed(), path, fp, kind)
        return eng

    @ops.on_device
    @torch.inference_mode()
    def vocode(self, speech_tokens, gen_ref, z=None, phase=None, noise=None, n_cfm_timesteps=10, drop_last_token=False, sync=True, hift_stream=None):
        """S3Gen.inference for a list of 1-D token tensors (already valid ids).  Returns (list of 1-D wav tensors on
        device, mel (B, 2Nmax, 80) channel-last).  hift_stream (synthesize_pipelined): the vocoder runs on THAT stream behind an event, so the flow
        stream is free for the next batch's encoder + CFM; the returned waveforms belong to it."""
        B = len(speech_tokens)
        ns = [int(t.numel()) for t in speech_tokens]
        Nmax = max(ns)
        tok = torch.zeros(B, Nmax, dtype=torch.long)
        for b, t in enumerate(speech_tokens):
            tok[b, : ns[b]] = t
        lens = torch.tensor(ns, dtype=torch.int32)
        same = all(n == Nmax for n in ns)
        refs = list(gen_ref) if isinstance(gen_ref, (list, tuple)) else [gen_ref] * B  # one voice for all, or one per utterance (s3gen.py:173-229)
        assert len(refs) == B, f"{len(refs)} voices for {B} utterances"
        shorts = [int(r["prompt_feat"].reshape(-1, 80).shape[0]) - 2 * int(r["prompt_token"].numel()) for r in refs]

        def run():
            t0 = time.perf_counter()
            mel = self.flow.inference(tok.to(self.dev), lens.to(self.dev), gen_ref, z=z, n_steps=n_cfm_timesteps)
            if sync:  # per-stage wall times; the pipelined mode never blocks the host between stages
                torch.cuda.synchronize()
            t1 = time.perf_counter()
            # short_b > 0 only when the prompt mel has an odd frame more than 2 * prompt tokens (flow.py:170-195); per voice when the batch mixes voices
            mel_lens = None if same and len(set(shorts)) == 1 else (2 * lens - torch.tensor(shorts, dtype=torch.int32)).to(self.dev)
            if hift_stream is not None:
                ev = torch.cuda.Event()
                ev.record()
                with torch.cuda.stream(hift_stream):
                    hift_stream.wait_event(ev)
                    mel.record_stream(hift_stream)
                    if mel_lens is not None:
                        mel_lens.record_stream(hift_stream)
                    wav, _ = self.hift.inference(mel, phase=phase, noise=noise, lens=mel_lens, fade=True)
            else:
                wav, _ = self.hift.inference(mel, phase=phase, noise=noise, lens=mel_lens, fade=True)
            if sync:
                torch.cuda.synchronize()
            t2 = time.perf_counter()
            self.last_timing.update(flow_s=t1 - t0, hift_s=t2 - t1)
            return wav, mel

        wav, mel = _range_checked(self, run, check=sync)
        out = []
        for b, n in enumerate(ns):
            keep = max(1, n - 1) if drop_last_token else n
            out.append(wav[b, : min(keep * SAMPLES_PER_TOKEN, (2 * n - shorts[b]) * (SAMPLES_PER_TOKEN // 2))])
        return out, mel

    @ops.on_device
    @torch.inference_mode()
    def synthesize(self, text_tokens, t3_conds, gen_ref, *, max_new_tokens=1000, temperature=0.8, top_p=1.0, min_p=0.05,
                   repetition_penalty=1.2, cfg_weight=0.5, uniforms=None, ban_eos=False, ban_from=0, z=None, phase=None,
                   noise=None, n_cfm_timesteps=10, drop_last_token=True):
        """Full hot path for B utterances.  Returns (wavs: list of 1-D device tensors, speech_tokens: list)."""
        self.co_resident(False)  # the serial schedule runs every kernel on its fastest-alone form (a no-op unless synthesize_pipelined ran before)
        t0 = time.perf_counter()
        toks = self.t3.generate(t3_conds, text_tokens, max_new_tokens=max_new_tokens, temperature=temperature, top_p=top_p,
                   min_p=min_p, repetition_penalty=repetition_penalty, cfg_weight=cfg_weight, uniforms=uniforms,
                   ban_eos=ban_eos, ban_from=ban_from)
        torch.cuda.synchronize()
        self.last_timing = dict(t3_s=time.perf_counter() - t0)
        st = [drop_invalid_tokens(t) for t in toks]
        st = [t if t.numel() > 0 else torch.zeros(1, dtype=torch.long) for t in st]
        wavs, _ = self.vocode(st, gen_ref, z=z, phase=phase, noise=noise, n_cfm_timesteps=n_cfm_timesteps,
                              drop_last_token=drop_last_token)
        self.last_timing["total_s"] = time.perf_counter() - t0
        return wavs, st


    def co_resident(self, on):
        """Both stages on (or off) the kernel forms whose workgroups can share a CU with the other stage's (T3Engine.co_resident, FlowEngine.co_resident)."""
        if self.t3 is not None and hasattr(self.t3, "co_resident"):
            self.t3.co_resident(on)
        self.flow.co_resident(on)

    def _pipeline_streams(self, priorities=(-1, 0)):
        """The T3 / flow streams of the overlapped schedules (created once; HIP priorities: measured, no effect either way)."""
        if not hasattr(self, "_s_t3"):
            pt3, pvoc = priorities
            self._s_t3 = torch.cuda.Stream(device=self.dev, priority=pt3)
            self._s_t3x = [self._s_t3] + [torch.cuda.Stream(device=self.dev, priority=pt3) for _ in range(2)]  # one per T3 state in flight
            self._s_voc = torch.cuda.Stream(device=self.dev, priority=pvoc)

    @torch.inference_mode()
    def synthesize_pipelined(self, jobs, co_resident=True, host_threads=True, t3_in_flight=2, stream_priorities=(-1, 0), **kw):
        """Throughput mode for a stream of batches: T3 of batch k+1 runs on a high-priority HIP stream WHILE the flow
        matching + vocoder of batch k run on a second stream.  jobs: list of dicts(text_tokens=[...], t3_conds=..., gen_ref=...); yields
        (wavs, tokens, latency_s) per job in order.  Results are identical to synthesize() called per job.

        What makes the two stages actually overlap (round 5, profiles/r05_overlap_*; none of it changes a result):
          * co_resident: a chain of small dependent kernels keeps its pace beside chip-filling kernels of another stream only if its workgroups FIT
            on the CUs beside theirs (scripts/micro/concur.hip: 1.0x when they fit, 6-7x slower when every launch has to wait for workgroups to retire).
            The CFM runs its plane GEMMs / attention on one-workgroup-per-CU forms (8 waves x <= 120 VGPRs + 96 KiB LDS; 4 waves x 200 VGPRs), the
            decode step on launches of <= 8 waves x <= 128 VGPRs; the decode kernels raise their wave priority (s_setprio 3).
          * host_threads: the decode loop is 250 hipGraph launches of 153 kernel nodes -- ~0.6 ms of HOST time each, invisible in the serial schedule
            (the GPU needs 1.1 ms per token) but 150-180 ms per batch in front of the flow's own ~130 ms of launches when one thread enqueues both
            (scripts/overlap_probe.py: the flow started 180 ms late).  The T3 enqueue runs on a second host thread (graph launches and the ctypes
            calls of the C stage seams release the GIL)."""
        import threading
        torch.cuda.set_device(self.dev)  # a generator cannot hold a device guard across yields: pin the device for the caller
        self._pipeline_streams(stream_priorities)
        t3_kw = {k: kw[k] for k in ("max_new_tokens", "temperature", "top_p", "min_p", "repetition_penalty", "cfg_weight", "ban_eos",
                                    "ban_from") if k in kw}
        self.co_resident(bool(co_resident))
        # ... and the flow + vocoder stream carries the co-resident ATTRIBUTE: its LayerNorm / split-GEMM launches (encoder, vocoder) keep to one or two
        # workgroups per CU as well (cbx_set_stream_coresident; no effect on results)
        from ._lib import check, lib
        # (with TWO T3 batches in flight the flow stream is the critical one and T3 has slack: capping its short LayerNorm launches costs more than the
        # decode chains gain -- 217.2x with the attribute, 220.8x without, same box; it stays on for t3_in_flight = 1)
        attr = bool(co_resident) and t3_in_flight < 2
        check(lib.cbx_set_stream_coresident(self._s_voc.cuda_stream, int(attr)), "cbx_set_stream_coresident")
        torch.cuda.synchronize()

        def enqueue_t3(job, box):
            try:
                torch.cuda.set_device(self.dev)
                with torch.inference_mode(), torch.cuda.stream(self._s_t3):
                    box["handle"] = self.t3.generate(job["t3_conds"], job["text_tokens"], async_mode=True, uniforms=job.get("uniforms"), **t3_kw)
            except BaseException as e:  # re-raised by the consumer thread
                box["error"] = e

        def voc_of(job, st):
            with torch.cuda.stream(self._s_voc):
                def voc():
                    wavs, _ = self.vocode(st, job["gen_ref"], z=job.get("z"), phase=job.get("phase"), noise=job.get("noise"),
                                          n_cfm_timesteps=kw.get("n_cfm_timesteps", 10),
                                          drop_last_token=kw.get("drop_last_token", True), sync=False)
                    return [w.cpu() for w in wavs]  # D2H on the vocoder stream: returns when this batch's audio is on the host
                return _range_checked(self, voc)

        def tokens_of(toks):
            st = [drop_invalid_tokens(t) for t in toks]
            return [t if t.numel() > 0 else torch.zeros(1, dtype=torch.long) for t in st]

        if not host_threads:  # round 4's form: one host thread enqueues T3(k + 1), then flow + vocoder(k)
            try:
                pending = None  # (job, speech tokens, t_start)
                for k in range(len(jobs) + 1):
                    box = {}
                    if k < len(jobs):
                        t_start = time.perf_counter()
                        enqueue_t3(jobs[k], box)
                        if "error" in box:
                            raise box["error"]
                    if pending is not None:
                        job, st, t0 = pending
                        host = voc_of(job, st)
                        yield host, st, time.perf_counter() - t0
                    pending = None
                    if "handle" in box:
                        with torch.cuda.stream(self._s_t3):
                            toks = self.t3.collect(box["handle"])
                        pending = (jobs[k], tokens_of(toks), t_start)
            finally:
                torch.cuda.synchronize()
                self.co_resident(False)
            return

        # ---- two host threads, two T3 states: the worker enqueues T3(k) into state k % 2 as soon as batch k - 2's tokens were collected, so the T3 stream
        #      runs T3(0), T3(1), ... back to back (the Python prologue of a generate() call -- conditioning, embeddings, ~10 ms -- and the per-batch host
        #      round trip are off its critical path); this thread waits for T3(k)'s END EVENT on the vocoder stream, fetches its tokens there (a copy on the
        #      T3 stream would queue behind T3(k + 1)) and enqueues flow + vocoder(k) beside T3(k + 1).
        import queue
        n_t3 = max(1, min(3, int(t3_in_flight)))
        q, stop = queue.Queue(), threading.Event()
        n_slots = max(2, n_t3)
        slot_free = [threading.Semaphore(1) for _ in range(n_slots)]

        def worker():
            try:
                torch.cuda.set_device(self.dev)
                for k, job in enumerate(jobs):
                    slot_free[k % n_slots].acquire()
                    if stop.is_set():
                        return
                    t_start = time.perf_counter()
                    with torch.inference_mode(), torch.cuda.stream(self._s_t3x[k % n_t3]):
                        h = self.t3.generate(job["t3_conds"], job["text_tokens"], async_mode=True, uniforms=job.get("uniforms"), slot=k % n_slots, **t3_kw)
                        ev = torch.cuda.Event()
                        ev.record()
                    q.put((h, ev, t_start))
            except BaseException as e:  # re-raised by the consumer thread
                q.put(e)

        # (the vocoder of batch k on a stream of its own was measured WORSE -- 208.9x against 226.3x, same box, profiles/r05_throughput_schedule_sweep.log: a
        # fourth stream of chip-filling work thrashes like a third decode chain does -- and is gone.)  Consecutive batches report fp16-range trips into
        # alternating flag words (ops.select_range_flag: a launch carries the word that was registered when it was ENQUEUED).
        import collections
        inflight = collections.deque()
        voc_kw = dict(n_cfm_timesteps=kw.get("n_cfm_timesteps", 10), drop_last_token=kw.get("drop_last_token", True))

        def start_voc(job, st, which):
            ops.select_range_flag(self.dev, which)
            with torch.cuda.stream(self._s_voc):
                wavs, _ = self.vocode(st, job["gen_ref"], z=job.get("z"), phase=job.get("phase"), noise=job.get("noise"), sync=False, **voc_kw)
                host = [torch.empty(w.shape, dtype=w.dtype, pin_memory=True).copy_(w, non_blocking=True) for w in wavs]
                ev = torch.cuda.Event()
                ev.record()
            return host, ev

        def finish(job, st, t0, host, ev, which):
            ev.synchronize()
            if 16 in (self.flow.precision, self.hift.precision) and ops.range_flag_tripped(self.dev, which):
                warnings.warn("an S3Gen operand exceeded the fp16 range: repeating flow matching + vocoder of this batch at bf16x6")
                global RANGE_TRIPS
                RANGE_TRIPS += 1
                saved = self.flow.precision, self.hift.precision
                self.flow.precision, self.hift.precision = (6 if p == 16 else p for p in saved)
                try:
                    with torch.cuda.stream(self._s_voc):
                        wavs, _ = self.vocode(st, job["gen_ref"], z=job.get("z"), phase=job.get("phase"), noise=job.get("noise"), sync=False, **voc_kw)
                        host = [w.cpu() for w in wavs]
                finally:
                    self.flow.precision, self.hift.precision = saved
            return host, st, time.perf_counter() - t0

        th = threading.Thread(target=worker, name="cbx-t3-enqueue", daemon=True)
        th.start()
        try:
            for k, job in enumerate(jobs):
                item = q.get()
                if isinstance(item, BaseException):
                    raise item
                h, ev, t0 = item
                with torch.cuda.stream(self._s_voc):
                    self._s_voc.wait_event(ev)
                    toks = self.t3.collect(h)  # blocks this thread until T3(k) is done; T3(k + 1) is already enqueued behind it
                slot_free[k % n_slots].release()
                st = tokens_of(toks)
                host, evv = start_voc(job, st, k % 2)
                inflight.append((job, st, t0, host, evv, k % 2))
                yield finish(*inflight.popleft())
            while inflight:
                yield finish(*inflight.popleft())
        finally:
            stop.set()
            for sem in slot_free:
                sem.release()
            th.join()
            ops.select_range_flag(self.dev, 0)
            torch.cuda.synchronize()
            self.co_resident(False)  # callers of vocode() / flow.inference() / synthesize_stream() get the fastest-alone forms back


def _stream_plan(n_tokens, done, exhausted, lookahead):
    """Per utterance: (final?, mel frames to hold back).  An utterance is final once its EOS was sampled or the step budget is spent;
    until then the encoder's `lookahead` tokens (2 mel frames each) are not vocoded yet."""
    fin = [bool(d) or exhausted for d in done]
    return fin, [0 if f else 2 * lookahead for f in fin]


def stream_token_schedule(n_tokens, first_chunk=25, chunk=50, lookahead=3, chunk_growth=1.0):
    """Tokens available to round 0, 1, 2, ... of synthesize_stream for an utterance of `n_tokens` sampled tokens: first_chunk + lookahead,
    then `chunk` more per round, each chunk `chunk_growth` times the previous one (1.0: constant chunks -- every round re-runs encoder + CFM
    over all tokens so far, so the total work grows with the number of rounds; a growth of 2 keeps the first-audio latency and bounds the
    total at about twice the one-shot synthesis)."""
    out, n, c = [], min(n_tokens, first_chunk + lookahead), float(chunk)
    while True:
        out.append(n)
        if n >= n_tokens:
            return out
        n = min(n_tokens, n + max(1, int(round(c))))
        c *= chunk_growth


def synthesize_stream(self, text_tokens, t3_conds, gen_ref, *, first_chunk=25, chunk=50, chunk_growth=1.0, lookahead=3, fade=480, max_new_tokens=1000,
                      temperature=0.8, top_p=1.0, min_p=0.05, repetition_penalty=1.2, cfg_weight=0.5, uniforms=None, ban_eos=False,
                      ban_from=0, z=None, phase=None, noise=None, n_cfm_timesteps=10, drop_last_token=True, overlap=True, first_alone=True, run_ahead=2):
    """Chunked synthesis: first audio after `first_chunk` tokens instead of after the whole utterance.

    The reference is non-streaming; of its vestigial hooks only HiFT's `cache_source` works (hifigan.py:470-472) -- `finalize=False`
    (flow.py:170-171) raises a shape error there -- so the schedule is this build's own, with its own oracle
    (tests/test_stream_gpu.py restates it on the CPU oracle):
      * round r works on the first n_r tokens, n_0 = first_chunk + lookahead, n_r = n_{r-1} + chunk * chunk_growth^(r-1) (stream_token_schedule);
      * every round runs encoder + CFM over the tokens so far with the same noise realisation, masking the last 2 * lookahead mel frames of
        unfinished utterances (the encoder looks 3 tokens ahead), and HiFT with the previous round's source as `cache_source`
        (phase-continuous excitation);
      * new samples are emitted up to `fade` samples before the end of what the round could vocode; that tail is cross-faded (linear ramp)
        with the next round's re-synthesis of the same samples.
    The last round is a full synthesis: identical mel to synthesize() for the same noise.
    overlap (round 6, the default): the two stages of the SAME utterances run side by side -- a second host thread keeps the T3 decode going on
    its own high-priority stream (at most two rounds ahead of the vocoder; graph replays of the captured step), this thread waits for a round's
    tokens on the flow stream and runs that round's flow + vocoder there, both stages on their co-resident kernel forms (synthesize_pipelined).
    A round still sees exactly its n_r tokens, so every yielded sample is the one the serial form (overlap=False) yields.
    Yields dicts {wavs: [B CPU tensors of NEW samples], final: [B bools], n_tokens: [B]}; concatenating an utterance's pieces gives its waveform."""
    torch.cuda.set_device(self.dev)
    dev, B = self.dev, len(text_tokens)
    P = gen_ref["prompt_token"].shape[-1]
    N = max_new_tokens
    assert gen_ref["prompt_feat"].shape[-2] == 2 * P, "chunked synthesis needs a whole-token prompt (embed_ref output trimmed to 2 frames per token)"
    if z is None:
        z = torch.randn(B, 2 * (P + N), 80, device=dev)
    if phase is None:
        phase = (torch.rand(B, 9, device=dev) * 2 - 1) * 3.141592653589793
        phase[:, 0] = 0
    if noise is None:
        noise = torch.randn(B, 9, SAMPLES_PER_TOKEN * N, device=dev)
    z, phase, noise = z.to(dev), phase.to(dev).reshape(B, 9), noise.to(dev)
    t3_kw = dict(max_new_tokens=N, temperature=temperature, top_p=top_p, min_p=min_p, repetition_penalty=repetition_penalty, cfg_weight=cfg_weight,
                 uniforms=uniforms, ban_eos=ban_eos, ban_from=ban_from, async_mode=True)
    totals = stream_token_schedule(N, first_chunk, chunk, lookahead, chunk_growth)  # tokens decoded when round r starts
    emitted, tails, closed, cache = [0] * B, [None] * B, [False] * B, [None]
    ramp = torch.linspace(0.0, 1.0, fade + 2, device=dev)[1:-1]

    def one_round(toks, done, exhausted):
        """flow + vocoder over the tokens so far on the CURRENT stream -> the dict this generator yields"""
        fin, hold = _stream_plan([t.numel() for t in toks], done, exhausted, lookahead)
        st = [drop_invalid_tokens(t) for t in toks]
        st = [t if t.numel() > 0 else torch.zeros(1, dtype=torch.long) for t in st]
        ns = [int(t.numel()) for t in st]
        Nk = max(ns)
        frames = [max(0, 2 * n - hb) for n, hb in zip(ns, hold)]
        out = [torch.zeros(0)] * B
        if max(frames) > 0:
            tok = torch.zeros(B, Nk, dtype=torch.long)
            for b, t in enumerate(st):
                tok[b, : ns[b]] = t
            fl = torch.tensor(frames, dtype=torch.int32, device=dev)

            def run():
                mel = self.flow.inference(tok.to(dev), torch.tensor(ns, dtype=torch.int32, device=dev), gen_ref, z=z[:, : 2 * (P + Nk)],
                                          n_steps=n_cfm_timesteps, hold_back=hold)
                return self.hift.inference(mel, phase=phase, noise=noise[:, :, : 480 * mel.shape[1]], lens=fl, fade=True,
                                           cache_source=cache[0])
            wav, src = _range_checked(self, run)
            cache[0] = src[:, : 480 * min(frames)].clone() if min(frames) > 0 else None
            for b in range(B):
                if closed[b]:
                    continue
                avail = 480 * frames[b]
                if fin[b]:
                    keep = max(1, ns[b] - 1) if drop_last_token else ns[b]
                    avail = min(avail, keep * SAMPLES_PER_TOKEN)
                end = avail if fin[b] else max(emitted[b], avail - fade)
                new = wav[b, emitted[b]: end].clone()
                if tails[b] is not None and new.numel() > 0:
                    k = min(tails[b].numel(), new.numel())
                    new[:k] = tails[b][:k] * (1.0 - ramp[:k]) + new[:k] * ramp[:k]
                tails[b] = None if fin[b] else wav[b, end: min(avail, end + fade)].clone()
                emitted[b] = end
                closed[b] = fin[b]
                out[b] = new.cpu()
        return dict(wavs=out, final=list(fin), n_tokens=ns)

    if not overlap:  # the serial form of rounds 3-5: T3 waits while a round is synthesised
        h = self.t3.generate(t3_conds, text_tokens, run_steps=totals[0], **t3_kw)
        for r, n_r in enumerate(totals):
            toks, done = self.t3.peek(h)
            exhausted = h["next_i"] >= h["max_new_tokens"]
            yield one_round(toks, done, exhausted)
            if all(closed) or exhausted:
                return
            self.t3.advance(h, totals[r + 1] - n_r)
        return

    # ---- overlapped: T3 on its own stream + host thread, at most `ahead` rounds in front of the vocoder
    import queue
    import threading
    self._pipeline_streams()
    self.co_resident(True)
    torch.cuda.synchronize()
    # T3 run-ahead policy (round 6, measured: profiles/r06_streaming_*): the FIRST round's flow + vocoder decide the first-audio latency, so T3 does not
    # decode beside them (`first_alone`); once the first audio is out T3 may be up to `ahead` rounds in front of the vocoder
    ahead = max(1, int(run_ahead))
    q, stop, credit = queue.Queue(), threading.Event(), threading.Semaphore(1 if first_alone else ahead)

    def worker():
        try:
            torch.cuda.set_device(dev)
            h = None
            with torch.inference_mode(), torch.cuda.stream(self._s_t3):
                for r, n_r in enumerate(totals):
                    credit.acquire()
                    if stop.is_set():
                        return
                    if h is None:
                        h = self.t3.generate(t3_conds, text_tokens, run_steps=n_r, **t3_kw)
                    else:
                        self.t3.advance(h, n_r - totals[r - 1])
                    ev = torch.cuda.Event()
                    ev.record()
                    q.put((h, ev, n_r))
        except BaseException as e:  # re-raised by the consumer
            q.put(e)

    th = threading.Thread(target=worker, name="cbx-t3-stream", daemon=True)
    th.start()
    try:
        for r in range(len(totals)):
            item = q.get()
            if isinstance(item, BaseException):
                raise item
            h, ev, n_r = item
            # the FIRST round decides the first-audio latency and T3 is far ahead of what round 1 needs: its flow runs on the fastest-alone kernel forms (T3's
            # workgroups wait for CU slots meanwhile); from round 1 on both stages share the CUs on the co-resident forms
            self.flow.co_resident(r > 0 or not first_alone)
            with torch.cuda.stream(self._s_voc):  # (not held across the yield: the caller keeps its own current stream)
                self._s_voc.wait_event(ev)
                toks, done = self.t3.peek(h)  # synchronises THIS stream (behind the event); T3 may already be further: a round sees its n_r tokens
                done = [bool(d) and int(t.numel()) <= n_r for d, t in zip(done, toks)]
                toks = [t[:n_r] for t in toks]
                exhausted = n_r >= N
                res = one_round(toks, done, exhausted)
            for _ in range(ahead if (first_alone and r == 0) else 1):
                credit.release()
            yield res
            if all(closed) or exhausted:
                return
    finally:
        stop.set()
        for _ in range(ahead + 1):
            credit.release()
        th.join()
        torch.cuda.synchronize()
        self.co_resident(False)


S3GEN_SIL = 4299  # reference models/s3gen/const.py:2


ChatterboxEngine.synthesize_stream = torch.inference_mode()(synthesize_stream)


class TurboEngine:
    """ChatterboxTurboTTS hot path (reference tts_turbo.py:298-317): GPT-2 T3 (no CFG) -> ids < 6561 + 3 silence tokens ->
    2-step meanflow S3Gen (no CFG) -> HiFT."""

    def __init__(self, t3_sd, s3gen_sd, device="cuda", n_t3_layers=None):
        from .t3_turbo import T3TurboEngine
        self.dev = torch.device(device)
        self.t3 = T3TurboEngine(t3_sd, self.dev, n_layers=n_t3_layers)
        self.flow = FlowEngine(s3gen_sd, self.dev, meanflow=True)
        self.hift = HiFTEngine(s3gen_sd, self.dev)
        self.last_timing = {}

    vocode = ChatterboxEngine.vocode

    @ops.on_device
    @torch.inference_mode()
    def synthesize(self, text_tokens, t3_conds, gen_ref, *, max_gen_len=1000, temperature=0.8, top_k=1000, top_p=0.95,
                   repetition_penalty=1.2, uniforms=None, ban_eos=False, ban_from=0, z=None, phase=None, noise=None):
        t0 = time.perf_counter()
        toks = self.t3.generate(t3_conds, text_tokens, max_gen_len=max_gen_len, temperature=temperature, top_k=top_k, top_p=top_p,
                                repetition_penalty=repetition_penalty, uniforms=uniforms, ban_eos=ban_eos, ban_from=ban_from)
        torch.cuda.synchronize()
        self.last_timing = dict(t3_s=time.perf_counter() - t0)
        sil = torch.full((3,), S3GEN_SIL, dtype=torch.long)
        st = [torch.cat([t[t < SPEECH_VOCAB], sil]) for t in toks]
        wavs, _ = self.vocode(st, gen_ref, z=z, phase=phase, noise=noise, n_cfm_timesteps=2, drop_last_token=False)
        self.last_timing["total_s"] = time.perf_counter() - t0
        return wavs, st
