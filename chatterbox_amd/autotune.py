"""Decode-step autotuner of the Llama T3 engine: picks the launch geometry of the token step on the GPU it runs on.

The token step of `T3.inference` (reference models/t3/t3.py:338-386) is 5 launches per layer whose geometry is a free choice that does not
change the arithmetic: output columns per workgroup of the q/k/v and o / down GEMVs (`T3Engine.tune`: qkv_tc, od_tc), and the K / V stream
of the decode attention (`cbx_set_decode_attn_pipeline`, `cbx_set_decode_attn_unroll`).  Which one is fastest depends on the box (the pool's
MI355X boxes differ by 3-10 %), the batch (rows = 2 B) and the context, so it is measured, the way MIOpen's find step picks a convolution:

    report = T3Engine.autotune(B=8)          # times every candidate on a synthetic decode state in a CHILD process, adopts the winner

Rules:
  * only candidates on the committed ALLOW-LIST (`decode_green.json`: geometries whose hardware tests -- tests/test_zz_abi_v9_gpu.py
    `test_green_variant_*`, run on an MI355X -- are green) are timed at all when the caller asks for `green_only` (bench.py does);
  * a candidate is ADOPTED only if its logits are bit-identical to the current geometry's (same products, same summation order) on a PROBE of
    single token steps over ragged contexts {1, 38, 63, 64, 65, 225, 640} (fewer rows than one attention step, exactly one, the ragged tail of the
    second register set, the bench context, a split-grid context) AND after the timed run, and it is faster by `min_gain`; candidates that sum in another (equally valid fp32) order -- the down projection without
    split-K partial images, another wave count -- are timed and reported (`"reorders": true`) but not adopted unless `allow_reorder=True`;
  * the measurement runs in a child process on seeded synthetic weights of the engine's shape (time does not depend on weight values): a
    candidate that faults or hangs takes the child down, not the serving process, and the engine keeps its current geometry;
  * timing is HIP events around hipGraph replays of the whole token step (every GEMV, the attention, the head and the sampler), i.e. exactly
    what `generate()` replays per token.

`python -m chatterbox_amd.autotune --layers 30 --batch 8 --ctx 224` prints the report as one JSON line (the child's protocol).
"""
import json
import os
import subprocess
import sys
import time

# Geometry candidates, each a DIFF on top of the geometry the engine currently runs ({} = keep it).  GEMV side (keys of T3Engine.tune):
TILE_VARIANTS = (dict(),
                 dict(qkv_ks=0, head_ct=0),                  # the one-tile q/k/v + head kernels of rounds 2-3 (every workgroup reads all of x + the partial images)
                 dict(head_ct=0),
                 dict(half_tiles=0, d_ks2=4, d_nw2=8),       # down projection on 16-column tiles, 4 split-K partial images (x : W = 1 : 1 per workgroup)
                 dict(od_tc=4))                              # o / down projections on 4-column tiles (256 workgroups)
# launch knobs of the decode attention (cbx_decode_attn_t.pipeline / unroll), tried on top of the best tile geometry
ATTN_VARIANTS = (dict(da_pipe=0), dict(da_pipe=1), dict(da_pipe=3), dict(da_pipe=5))
# cbx_gemv_t.flags of every GEMV launch, tried on top of the best geometry so far
EPI_VARIANTS = (dict(pre_epi=0),)
# the engines' default launch knobs (cbx_decode_attn_t.pipeline / unroll, cbx_gemv_t.flags): round 4's hardware A/B (profiles/r04_decode_geometry_ab.log)
LIB_KNOBS = dict(da_pipe=7, da_u=4, deep=0, pre_epi=1)
# FROZEN reference geometry the allow-list is written against (the round-3 defaults): an entry of decode_green.json is the set of keys in which a
# full geometry (tune + knobs) differs from THIS, so entries keep their meaning when the engines' defaults move
BASE_TUNE = dict(qkv_nw=8, o_ks=4, gu_nw=8, d_ks=8, head_nw=4, o_nw2=8, d_ks2=2, d_nw2=16, half_tiles=1, qkv_tc=0, od_tc=0, prefill_prec=0, qkv_ks=0, qkv_ct=3, head_ct=0)
BASE_KNOBS = dict(da_pipe=0, da_u=4, deep=0, pre_epi=0)
PROBE_CTXS = (1, 38, 63, 64, 65, 225, 640)
GREEN_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "decode_green.json")


def canon(tune, knobs=None):
    """Canonical form of a FULL geometry (T3Engine.tune [+ knobs]; missing keys = the frozen base's): the sorted tuple of the keys in which it
    differs from BASE_TUNE / BASE_KNOBS -- the key of the allow-list."""
    full = dict(BASE_TUNE, **BASE_KNOBS)
    full.update(tune)
    full.update(knobs or {})
    if not full.get("qkv_ks"):
        full["qkv_ct"] = BASE_TUNE["qkv_ct"]  # (column tiles of a form that is off)
    base = dict(BASE_TUNE, **BASE_KNOBS)
    return tuple(sorted((k, int(x)) for k, x in full.items() if k in base and int(x) != base[k]))


def green_variants():
    """The committed allow-list: full geometries (canon form) that passed their hardware tests on an MI355X (written by scripts/green_variants.py from
    a GPU run, re-checked by tests/test_zz_abi_v9_gpu.py::test_green_variant_*).  The frozen base geometry is always on it (three rounds of goldens)."""
    try:
        with open(GREEN_FILE) as f:
            rows = json.load(f)["green"]
    except (OSError, ValueError, KeyError):
        rows = []
    return {()} | {canon(v) for v in rows}


def composed_candidates(tune0, knobs0):
    """Every geometry tune_decode can reach from (tune0, knobs0): tile diff x attention diff x epilogue diff, as full (tune, knobs) pairs."""
    out, seen = [], set()
    for t in TILE_VARIANTS:
        for a in (dict(),) + ATTN_VARIANTS:
            for e in (dict(),) + EPI_VARIANTS:
                tv, kv = split_variant(dict(t, **a, **e))
                full = (dict(tune0, **tv), dict(knobs0, **kv))
                c = canon(*full)
                if c not in seen:
                    seen.add(c)
                    out.append(full)
    return out


def env_knobs():
    """The engines' launch knobs as the environment asks for them (CBX_DA_PIPE, CBX_DA_U, CBX_GEMV_DEEP, CBX_GEMV_PRE_EPI: A/B scripts)."""
    e = os.environ.get
    i = lambda name, dflt: int(e(name)) if e(name) not in (None, "") else dflt
    return dict(da_pipe=i("CBX_DA_PIPE", LIB_KNOBS["da_pipe"]) & 7, da_u=i("CBX_DA_U", 0) or LIB_KNOBS["da_u"], deep=int(bool(i("CBX_GEMV_DEEP", LIB_KNOBS["deep"]))),
                pre_epi=int(bool(i("CBX_GEMV_PRE_EPI", LIB_KNOBS["pre_epi"]))))


def split_variant(v):
    """(tune keys, library knobs) of a candidate."""
    return {k: x for k, x in v.items() if k not in LIB_KNOBS}, {k: x for k, x in v.items() if k in LIB_KNOBS}


def tune_decode(eng, B=8, ctx=224, steps=24, reps=2, min_gain=0.01, allow_reorder=False, use_graph=True, tiles=TILE_VARIANTS, attn=ATTN_VARIANTS,
                epi=EPI_VARIANTS, log=None, green_only=False, probe_ctxs=PROBE_CTXS):
    """Time every candidate on `eng` (in this process) and return the report; `eng` is left on the geometry it came with.
    report["best"]: the fastest candidate whose logits are bit-identical to the current geometry's ({} = keep it); report["best_any"]: the
    fastest candidate overall, reordering ones included (== best unless a reordering candidate is faster still by min_gain) -- for callers
    that validate it on their own workload (T3Engine.autotune(validate=...)).  green_only: candidates off the allow-list are not run."""
    import torch
    base_tune, base_knobs = dict(eng.tune), dict(eng.knobs)
    rows, seen = [], {}
    green = green_variants() if green_only else None

    def full(v):
        t, k = split_variant(v)
        return dict(base_tune, **t), dict(base_knobs, **k)

    def run(v):
        eng.apply_variant(*full(v))
        probe = eng.probe_decode(B=B, ctxs=probe_ctxs)
        ms, lg = eng.measure_decode(B=B, ctx=ctx, steps=steps, reps=reps, use_graph=use_graph)
        return ms, torch.cat([probe.flatten(), lg.flatten(), eng.last_measure["final_logits"].flatten()])

    if green is not None and canon(base_tune, base_knobs) not in green:
        raise RuntimeError(f"the engine's current decode geometry {canon(base_tune, base_knobs)} is not on the hardware-green allow-list (decode_green.json)")
    ms0, ref = run({})
    scale = max(1.0, float(ref.abs().max()))
    rows.append(dict(variant={}, ms_per_token=round(ms0, 5), identical=True, reorders=False, max_abs_diff=0.0, valid=True))
    if log:
        log(f"autotune: current geometry {ms0:.4f} ms / token")

    def consider(v):
        key = tuple(sorted(v.items()))
        if not v or key in seen:
            return
        seen[key] = True
        if green is not None and canon(*full(v)) not in green:
            rows.append(dict(variant=v, skipped="not on the hardware-green allow-list (decode_green.json)"))
            return
        try:
            ms, lg = run(v)
        except Exception as e:  # a candidate this build / shape does not support: reported, never adopted
            rows.append(dict(variant=v, error=f"{type(e).__name__}: {e}"[:200]))
            return
        same = bool(torch.equal(lg, ref))
        diff = float((lg - ref).abs().max())
        ok_num = same or diff <= 2e-4 * scale  # another fp32 summation order of the same products
        rows.append(dict(variant=v, ms_per_token=round(ms, 5), identical=same, reorders=not same, max_abs_diff=diff, valid=ok_num))
        if log:
            log(f"autotune: {v} {ms:.4f} ms / token, identical={same} (max |d logits| {diff:.2e})")

    def pick(identical_only):
        ok = [r for r in rows if r.get("valid") and (r["identical"] or not identical_only) and r["ms_per_token"] < ms0 * (1.0 - min_gain)]
        return dict(min(ok, key=lambda r: r["ms_per_token"])["variant"]) if ok else {}

    for v in tiles[1:]:
        consider(v)
    for base in (pick(True), pick(False)):  # the attention knobs on top of the best identical tile geometry and of the best one overall
        for a in attn:
            consider(dict(base, **a))
    for base in (pick(True), pick(False)):  # the GEMV epilogue prefetch on top of whatever leads now
        for a in epi:
            consider(dict(base, **a))

    def confirm(v):  # back to back against the current geometry (`reps` more rounds each): the pool's boxes drift by a few per cent over seconds
        if not v:
            return {}, ms0
        a0 = min(run({})[0] for _ in range(reps))
        a1 = min(run(v)[0] for _ in range(reps))
        rows.append(dict(confirm=dict(variant=v, current=round(a0, 5), candidate=round(a1, 5))))
        return (v, a1) if a1 < a0 * (1.0 - min_gain) else ({}, min(a0, ms0))

    best, best_ms = confirm(pick(True))
    cand = pick(False)
    best_any, any_ms = (best, best_ms) if cand == best or tuple(sorted(cand.items())) == tuple(sorted(best.items())) else confirm(cand)
    if not best_any or not any_ms < best_ms * (1.0 - min_gain):
        best_any, any_ms = best, best_ms
    if allow_reorder:
        best, best_ms = best_any, any_ms
    eng.apply_variant(base_tune, base_knobs)
    eng.release_state(slot=7)  # the probe / measurement states (KV caches of 2B rows x 704 positions x L layers) must not stay pinned on a serving engine
    return dict(best=best, ms_per_token=round(best_ms, 5), best_any=best_any, ms_per_token_any=round(any_ms, 5), baseline_ms_per_token=round(ms0, 5),
                B=B, ctx=ctx, steps=steps, layers=eng.L, graph=bool(use_graph), allow_reorder=bool(allow_reorder), green_only=bool(green_only),
                probe_ctxs=list(probe_ctxs), candidates=rows)


def tune_in_child(layers, B, ctx, steps, reps, min_gain, allow_reorder, device_index, base_tune, base_knobs, timeout=180.0, log=None, green_only=False):
    """Run `python -m chatterbox_amd.autotune` and parse its report; any failure (non-zero exit, timeout, no JSON) returns {"error": ...}."""
    cmd = [sys.executable, "-m", "chatterbox_amd.autotune", "--layers", str(layers), "--batch", str(B), "--ctx", str(ctx), "--steps", str(steps),
           "--reps", str(reps), "--min-gain", str(min_gain), "--device", str(device_index), "--tune", json.dumps(base_tune), "--knobs",
           json.dumps(base_knobs)] + (["--allow-reorder"] if allow_reorder else []) + (["--green-only"] if green_only else [])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CBX_T3_TUNE", "CBX_DA_PIPE", "CBX_DA_U", "CBX_GEMV_DEEP",
              "CBX_GEMV_PRE_EPI"):
        env.pop(k, None)  # the child is a plain single-device process whose starting geometry arrives on the command line
    for k in [k for k in env if k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS")) or (k == "LD_PRELOAD" and "rocprof" in env[k])]:
        env.pop(k)  # under rocprofv3 the candidates' kernels (same names, other geometries) must not enter the parent's kernel statistics
    t0 = time.perf_counter()
    try:
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=root)
    except OSError as e:
        return dict(error=f"spawn failed: {e}")
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        p.kill()
        try:
            p.communicate(timeout=10)
        except Exception:
            pass
        return dict(error=f"timeout after {timeout:.0f} s", wall_s=round(time.perf_counter() - t0, 1))
    rep = None
    for line in reversed(out.strip().splitlines()):
        if line.startswith("{"):
            try:
                rep = json.loads(line)
                break
            except ValueError:
                continue
    if p.returncode != 0 or rep is None:
        return dict(error=f"child exit {p.returncode}", stderr_tail=err[-400:], wall_s=round(time.perf_counter() - t0, 1))
    rep["wall_s"] = round(time.perf_counter() - t0, 1)
    if log:
        log(f"autotune child: {rep['wall_s']} s, best {rep['best']}")
    return rep


def main(argv=None, device=None):
    """`device` is for tests (the SIMT emulator drives this on the CPU); the command line always measures on the GPU."""
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--layers", type=int, default=30)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--ctx", type=int, default=224)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--min-gain", type=float, default=0.01)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--allow-reorder", action="store_true")
    ap.add_argument("--green-only", action="store_true", help="time only candidates on the hardware-green allow-list (decode_green.json)")
    ap.add_argument("--tune", default="{}", help="JSON: T3Engine.tune overrides of the starting geometry")
    ap.add_argument("--knobs", default="{}", help="JSON: library knobs (da_pipe, da_u, deep) of the starting geometry")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args(argv)
    import torch

    from . import synth
    from .t3 import T3Engine
    if device is None:
        assert torch.cuda.is_available(), "the autotuner measures on the GPU"
        torch.cuda.set_device(a.device)
        device = torch.device("cuda", a.device)
    eng = T3Engine(synth.t3_state_dict(a.layers, 0), device, n_layers=a.layers)
    eng.apply_variant(dict(eng.tune, **json.loads(a.tune)), dict(LIB_KNOBS, **json.loads(a.knobs)))
    log = (lambda m: print(m, file=sys.stderr, flush=True)) if a.verbose else None
    rep = tune_decode(eng, B=a.batch, ctx=a.ctx, steps=a.steps, reps=a.reps, min_gain=a.min_gain, allow_reorder=a.allow_reorder, log=log,
                      use_graph=device.type == "cuda", green_only=a.green_only)
    print(json.dumps(rep), flush=True)


if __name__ == "__main__":
    main()
