"""The throughput schedule with CO-RESIDENT kernels (run on the GPU box, unprofiled): T3 decode alone / flow + vocoder alone / both at once, for
  A  today's kernels;
  B  flow plane GEMMs on the one-workgroup-per-CU 8-wave form (tile 17: 96 KiB LDS, <= 120 VGPRs -> half of every SIMD's registers stay free);
  C  T3 on its co-residency geometry (down projection on 512-thread workgroups, gate | up on the <= 128-VGPR shallow form);
  D  B + C;
  E  D + the 4-wave plane attention (CBX_OV_ATTN=<version>, when the library has it).
profiles/r05_overlap_coresidency_micro.jsonl says a dependent chain keeps its pace beside chip-filling kernels iff its workgroups FIT beside theirs."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chatterbox_amd import _lib, synth
from chatterbox_amd.autotune import LIB_KNOBS
from chatterbox_amd.engine import ChatterboxEngine, drop_invalid_tokens

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
L = int(os.environ.get("CBX_AB_LAYERS", "30"))
eng = ChatterboxEngine(synth.t3_state_dict(L, 0), synth.s3gen_state_dict(0), dev, n_t3_layers=L)
B, N = 8, 250
t3c, gen = synth.t3_cond(prompt_len=150), synth.s3gen_ref()
texts = [synth.text_tokens(64, seed=b) for b in range(B)]
g = torch.Generator(device=dev).manual_seed(1)
u = torch.rand(B, N, generator=g, device=dev)
T = 2 * (gen["prompt_token"].shape[1] + N)
z = torch.randn(B, T, 80, generator=g, device=dev)
kw = dict(max_new_tokens=N, uniforms=u, ban_eos=True, ban_from=6561)
toks0 = [t.tolist() for t in eng.t3.generate(t3c, texts, **kw)]
st = [drop_invalid_tokens(torch.tensor(t)) for t in toks0]
wav0, mel0 = eng.vocode(st, gen, z=z, drop_last_token=True)
mel0 = mel0.clone()
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(device=dev, priority=-1), torch.cuda.Stream(device=dev)
base_tune, base_knobs = dict(eng.t3.tune), dict(eng.t3.knobs)
polite_tune = dict(base_tune, half_tiles=0, d_ks2=4, d_nw2=8)
polite_knobs = dict(base_knobs, shallow=1)
rows = []


def run(what):
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if what in ("t3", "both"):
            with torch.cuda.stream(sa):
                eng.t3.generate(t3c, texts, async_mode=True, **kw)
        if what in ("voc", "both"):
            with torch.cuda.stream(sb):
                eng.vocode(st, gen, z=z, drop_last_token=True, sync=False)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return round(1e3 * best, 1)


def config(label, flow_tile, t3_polite, attn=None):
    _lib.lib.cbx_set_planes_tile(flow_tile)
    if attn is not None:
        _lib.lib.cbx_set_attn_planes_version(attn)
    eng.t3.apply_variant(polite_tune if t3_polite else base_tune, polite_knobs if t3_polite else base_knobs)
    # same results first (tokens of the geometry; mel of the tile / attention form), then the three timings
    same_tok = [t.tolist() for t in eng.t3.generate(t3c, texts, **kw)] == toks0
    _, mel = eng.vocode(st, gen, z=z, drop_last_token=True)
    dmel = float((mel - mel0).abs().max())
    r = dict(config=label, tokens_equal=same_tok, mel_max_abs_diff=dmel, t3_ms=run("t3"), voc_ms=run("voc"), both_ms=run("both"))
    r["xRT_steady_state"] = round(B * (N - 1) / 25.0 / (r["both_ms"] * 1e-3), 1)
    rows.append(r)
    print(json.dumps(r), flush=True)


config("A today", 0, False)
config("B flow GEMMs on tile 17 (1 workgroup per CU, 8 waves)", 17, False)
config("C T3 co-residency geometry (down on 512-thread workgroups, shallow gate|up)", 0, True)
config("D = B + C", 17, True)
if os.environ.get("CBX_OV_ATTN"):
    config("E = D + plane attention version " + os.environ["CBX_OV_ATTN"], 17, True, int(os.environ["CBX_OV_ATTN"]))
    config("F = E with the flow on the default GEMM tiles", 0, True, int(os.environ["CBX_OV_ATTN"]))
    config("G = plane attention version " + os.environ["CBX_OV_ATTN"] + " only (T3 on its default geometry, default GEMM tiles)", 0, False, int(os.environ["CBX_OV_ATTN"]))
_lib.lib.cbx_set_planes_tile(0)
_lib.lib.cbx_set_attn_planes_version(0)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
