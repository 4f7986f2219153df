"""Data-parallel utterance sharding over the GPUs of one node (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference has no distributed path; utterances are independent, so the path shards with NO
collective inside it.  Only two exchanges exist, both outside the kernels' critical path:
  C1  broadcast of the packed voice `Conditionals` (~170 KB) from the rank that analysed the prompt,
  C2  gather of the finished waveforms (padded to the longest + int32 lengths) to rank 0 (`dist.gather`: only rank 0 receives).
Over 7 xGMI links x ~153 GB/s a 31 MB/GPU gather costs < 1 ms, so scaling is decided by load balance: shards are
contiguous blocks of the utterance list (length-sorted lists balance best).
"""
import torch
import torch.distributed as dist

_T3_KEYS = ("speaker_emb", "cond_prompt_speech_tokens", "emotion_adv")
_GEN_KEYS = ("prompt_token", "prompt_token_len", "prompt_feat", "embedding")


def shard_range(n_items, rank, world):
    """Contiguous block partition: the first (n % world) ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _flatten(t3_cond, gen_ref):
    tensors = [torch.as_tensor(t3_cond[k]) for k in _T3_KEYS] + [torch.as_tensor(gen_ref[k]) for k in _GEN_KEYS]
    meta = [(tuple(t.shape), t.dtype) for t in tensors]
    flat = torch.cat([t.reshape(-1).to(torch.float64) for t in tensors])  # ids < 2^53: exact in fp64
    return flat, meta


def _unflatten(flat, meta):
    out, off = [], 0
    for shape, dtype in meta:
        n = 1
        for s in shape:
            n *= s
        out.append(flat[off:off + n].to(dtype).reshape(shape))
        off += n
    return (dict(zip(_T3_KEYS, out[:3])), dict(zip(_GEN_KEYS, out[3:]), prompt_feat_len=None))


def broadcast_conditionals(t3_cond, gen_ref, src=0, device=None, force=False):
    """C1: every rank returns (t3_cond, gen_ref) of `src`.  Non-src ranks may pass None.  force: run the collectives even in a group of one
    (what a single GPU can exercise of the RCCL path)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return t3_cond, gen_ref
    rank = dist.get_rank()
    obj = [None]
    if rank == src:
        flat, meta = _flatten(t3_cond, gen_ref)
        obj = [(meta, flat.numel())]
    dist.broadcast_object_list(obj, src=src)  # tiny metadata (shapes/dtypes)
    meta, n = obj[0]
    dev = device or ("cuda" if dist.get_backend() == "nccl" else "cpu")
    buf = flat.to(dev) if rank == src else torch.empty(n, dtype=torch.float64, device=dev)
    dist.broadcast(buf, src=src)
    t3c, gen = _unflatten(buf.cpu(), meta)
    return t3c, gen


def gather_waveforms(wavs, dst=0, force=False):
    """C2: `wavs` is this rank's list of 1-D float tensors.  Returns, on rank `dst`, the list over all ranks in rank
    order (CPU tensors); None elsewhere.  A real gather-to-`dst`: every rank all-gathers two int64 (its utterance count and
    longest waveform) so that all agree on the padded block shape, then ONE `dist.gather` of the padded (n_max, L_max) fp32 block
    and one of the int32 lengths -- only `dst` receives the 31 MB / rank, the other ranks send and are done."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return [w.detach().cpu() for w in wavs]
    world, rank = dist.get_world_size(), dist.get_rank()
    # nccl (= RCCL) collectives need device tensors on THIS rank's GPU whatever the inputs are (host copies, an empty shard);
    # every rank must pick the same kind of device or the collective hangs.  gloo (CPU tests): host tensors.
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    shape = torch.tensor([len(wavs), max([w.numel() for w in wavs], default=0)], dtype=torch.int64, device=dev)
    shapes = [torch.zeros_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape)
    n_max = int(max(s[0] for s in shapes))
    l_max = int(max(s[1] for s in shapes))
    block = torch.zeros(n_max, l_max, dtype=torch.float32, device=dev)
    lens = torch.zeros(n_max, dtype=torch.int32, device=dev)
    for i, w in enumerate(wavs):
        block[i, : w.numel()] = w.to(dev)
        lens[i] = w.numel()
    blocks = [torch.empty_like(block) for _ in range(world)] if rank == dst else None
    all_lens = [torch.empty_like(lens) for _ in range(world)] if rank == dst else None
    dist.gather(block, blocks, dst=dst)
    dist.gather(lens, all_lens, dst=dst)
    if rank != dst:
        return None
    out = []
    for r in range(world):
        for i in range(int(shapes[r][0])):
            out.append(blocks[r][i, : int(all_lens[r][i])].cpu())
    return out


class AsyncGatherer:
    """C2 off the critical path of a stream of batches (engine.synthesize_pipelined): `post(wavs)` hands a finished batch to ONE background
    thread that runs `gather_waveforms` for the batches in the order they were posted (every rank posts the same number of batches in the same
    order, so the collectives line up); the posting thread goes on enqueueing the next batch.  `close()` waits for the gathers still in flight
    and returns, on `dst`, one list of waveforms per posted batch (None elsewhere) -- call it before any other collective of the process group
    (the barrier that ends a timed region).  Without an initialised process group (one GPU) a post is a host copy and no thread is started."""

    def __init__(self, dst=0, force=False):
        import queue
        import threading
        self.dst, self.force, self.results, self.error = dst, force, [], None
        self.active = dist.is_initialized() and (dist.get_world_size() > 1 or force)
        self._q = queue.Queue() if self.active else None
        self._thread = None
        if self.active:
            dev = torch.cuda.current_device() if dist.get_backend() == "nccl" else None

            def run():
                try:
                    if dev is not None:
                        torch.cuda.set_device(dev)
                    while True:
                        item = self._q.get()
                        if item is None:
                            return
                        self.results.append(gather_waveforms(item, dst=self.dst, force=self.force))
                except BaseException as e:  # re-raised by close()
                    self.error = e
            self._thread = threading.Thread(target=run, name="cbx-c2-gather", daemon=True)
            self._thread.start()

    def post(self, wavs):
        if self.active:
            self._q.put(list(wavs))
        else:
            self.results.append([w.detach().cpu() for w in wavs])

    def close(self):
        if self._thread is not None:
            self._q.put(None)
            self._thread.join()
            self._thread = None
            if self.error is not None:
                raise self.error
        return self.results
