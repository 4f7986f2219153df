"""CPU (-m "not gpu"): `python bench.py --gpus 2` must launch TWO ranks by itself (no wrapper), have them rendezvous, fire C1 and C2, and
print ONE JSON line with n_gpus = 2; a rank count that disagrees with --gpus must be an error, not a silent single-GPU run
(VERDICT r02 missing #1).  Runs bench.py's --selftest-rendezvous mode: gloo + host tensors, no kernels, no measurement."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, cwd=ROOT, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_gpus_2_spawns_two_ranks_and_reports_n_gpus_2():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "0", "--batch", "3", "--tokens", "20", "--selftest-rendezvous"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 6
    st = out["selftest"]
    assert st["ranks"] == 2 and st["c1_ok"] is True and st["c2_waveforms_gathered"] == 2 * 6  # 2 steps x (3 utterances x 2 ranks)
    assert out["value"] is None  # the self-test measures nothing
    # the throughput schedule's C2: batches posted to dist.AsyncGatherer by bench.pipelined_batches while a second host thread produces them
    assert st["pipelined_c2_ok"] is True and st["pipelined_batches"] == 2


def test_pipelined_c2_order_with_three_ranks_and_many_batches():
    """the collective order of the PIPELINED path (bench.pipelined_batches: a generator fed by a second host thread, C2 posted per batch
    to the background gatherer, closed before the barrier) over gloo, 3 ranks x 7 batches whose ranks finish their batches at different times."""
    r = _run(["--gpus", "3", "--steps", "7", "--warmup", "0", "--batch", "2", "--tokens", "6", "--selftest-rendezvous"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["selftest"]["ranks"] == 3 and out["selftest"]["pipelined_c2_ok"] is True and out["selftest"]["pipelined_batches"] == 7


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--selftest-rendezvous"], env_extra=dict(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_single_rank_selftest_needs_no_process_group():
    r = _run(["--gpus", "1", "--steps", "1", "--selftest-rendezvous", "--batch", "2", "--tokens", "10"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["selftest"]["c2_waveforms_gathered"] == 2 and out["selftest"]["pipelined_c2_ok"] is True
