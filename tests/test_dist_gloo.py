"""CPU (-m "not gpu"): the N>1 path with world_size 2 over gloo -- C1 broadcast of Conditionals, C2 gather of waveforms."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chatterbox_amd import dist as cdist, synth
    t3c, gen = (synth.t3_cond(), synth.s3gen_ref(n_prompt_tokens=20)) if rank == 0 else (None, None)
    t3c, gen = cdist.broadcast_conditionals(t3c, gen, src=0)
    ref_t3, ref_gen = synth.t3_cond(), synth.s3gen_ref(n_prompt_tokens=20)
    ok = all(torch.equal(t3c[k], ref_t3[k]) and t3c[k].dtype == ref_t3[k].dtype for k in ref_t3)
    ok &= all(torch.equal(gen[k], ref_gen[k]) for k in ref_gen if ref_gen[k] is not None) and gen["prompt_feat_len"] is None
    # ragged shards: rank 0 owns 3 utterances, rank 1 owns 2, all of different lengths
    lo, hi = cdist.shard_range(5, rank, world)
    wavs = [torch.full((100 * (i + 1),), float(i)) for i in range(lo, hi)]
    allw = cdist.gather_waveforms(wavs, dst=0)
    if rank == 0:
        ok &= len(allw) == 5 and all(w.numel() == 100 * (i + 1) and bool((w == i).all()) for i, w in enumerate(allw))
    else:
        ok &= allw is None
    # an EMPTY shard on one rank (1 utterance over 2 ranks) must neither hang nor mis-shape the collective
    lo, hi = cdist.shard_range(1, rank, world)
    one = cdist.gather_waveforms([torch.arange(7, dtype=torch.float32) for _ in range(lo, hi)], dst=0)
    if rank == 0:
        ok &= len(one) == 1 and one[0].tolist() == list(range(7))
    # strong sharding of configs[3]: 256 utterances over 8 ranks = 32 each, contiguous, complete
    cover = [cdist.shard_range(256, r, 8) for r in range(8)]
    ok &= all(b - a == 32 for a, b in cover) and cover[0][0] == 0 and cover[-1][1] == 256 and all(cover[i][1] == cover[i + 1][0] for i in range(7))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == {0: True, 1: True}
