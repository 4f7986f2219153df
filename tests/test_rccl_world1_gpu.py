"""What ONE GPU can exercise of the multi-GPU path: the RCCL process group, C1 (broadcast of the Conditionals) and C2
(gather of the waveforms) on device tensors in a group of one -- so that the first multi-GPU lease does not discover API errors.  The
scaling curve itself cannot be measured here: bench.py says so in its JSON line (`multi_gpu_note`)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_dist_collectives_over_rccl_in_a_group_of_one(dev):
    """dist.broadcast_conditionals / gather_waveforms with force=True under backend "nccl" (= RCCL), world size 1, in a child process (a process
    group is process state): the Conditionals come back bit-identical, the waveforms in order with their own lengths, an empty shard works."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from chatterbox_amd import dist as cdist, synth
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
t3c, gen = synth.t3_cond(), synth.s3gen_ref()
a, b = cdist.broadcast_conditionals(t3c, gen, src=0, device=torch.device("cuda", 0), force=True)
for k in cdist._T3_KEYS:
    assert torch.equal(torch.as_tensor(a[k]).cpu(), torch.as_tensor(t3c[k]).cpu()), k
for k in cdist._GEN_KEYS:
    assert torch.equal(torch.as_tensor(b[k]).cpu(), torch.as_tensor(gen[k]).cpu()), k
wavs = [torch.randn(n, device="cuda") for n in (2400, 960, 4801)]
out = cdist.gather_waveforms(wavs, dst=0, force=True)
assert len(out) == 3 and all(torch.equal(o, w.cpu()) for o, w in zip(out, wavs))
assert cdist.gather_waveforms([], dst=0, force=True) == []
dist.barrier()
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
''' % ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_bench_under_torchrun_with_one_rank_goes_through_rccl(dev):
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 --force-rccl` on a 2-layer model: the launcher path the driver uses
    for N > 1, the RCCL group, C1 before and C2 inside the timed region (checked against the local waveforms), one JSON line that says no N > 1
    curve exists."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--force-rccl", "--t3-layers", "2", "--tokens", "24", "--batch", "2",
           "--no-cpu-baseline", "--no-alt-precisions", "--no-streaming", "--no-autotune"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and "no N > 1 scaling curve" in line["multi_gpu_note"]
    assert line["t3_geometry"]["adopted"] == {} and line["t3_geometry"]["knobs"]["da_pipe"] == 7 and line["t3_geometry"]["on_green_list"]
