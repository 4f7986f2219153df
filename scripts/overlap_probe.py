"""Where does the time go when T3 decode and flow + vocoder run at once?  (GPU box, unprofiled)  Per configuration: HIP events at the start / end of each
stream's work, the order the host enqueued the two, and T3 beside the flow stage alone / the vocoder alone."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chatterbox_amd import _lib, synth
from chatterbox_amd.engine import ChatterboxEngine, drop_invalid_tokens

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
L = 30
eng = ChatterboxEngine(synth.t3_state_dict(L, 0), synth.s3gen_state_dict(0), dev, n_t3_layers=L)
B, N = 8, 250
t3c, gen = synth.t3_cond(prompt_len=150), synth.s3gen_ref()
texts = [synth.text_tokens(64, seed=b) for b in range(B)]
g = torch.Generator(device=dev).manual_seed(1)
u = torch.rand(B, N, generator=g, device=dev)
T = 2 * (gen["prompt_token"].shape[1] + N)
z = torch.randn(B, T, 80, generator=g, device=dev)
kw = dict(max_new_tokens=N, uniforms=u, ban_eos=True, ban_from=6561)
toks0 = eng.t3.generate(t3c, texts, **kw)
st = [drop_invalid_tokens(t) for t in toks0]
eng.vocode(st, gen, z=z, drop_last_token=True)
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(device=dev, priority=-1), torch.cuda.Stream(device=dev)
tok = torch.stack([t[:N - 1] if t.numel() >= N - 1 else torch.nn.functional.pad(t, (0, N - 1 - t.numel())) for t in st]).to(dev)
lens = torch.full((B,), tok.shape[1], dtype=torch.int32, device=dev)
mel_cache = {}


def ev():
    return torch.cuda.Event(enable_timing=True)


def both(order, flow_part="vocode"):
    torch.cuda.synchronize()
    e = {k: ev() for k in ("t0", "t3_s", "t3_e", "v_s", "v_e")}
    e["t0"].record()
    h0 = time.perf_counter()
    hw = {}

    def t3():
        with torch.cuda.stream(sa):
            e["t3_s"].record()
            eng.t3.generate(t3c, texts, async_mode=True, **kw)
            e["t3_e"].record()
        hw["t3_enqueued_ms"] = round(1e3 * (time.perf_counter() - h0), 1)

    def voc():
        with torch.cuda.stream(sb):
            e["v_s"].record()
            if flow_part == "vocode":
                eng.vocode(st, gen, z=z, drop_last_token=True, sync=False)
            elif flow_part == "flow":
                mel_cache["mel"] = eng.flow.inference(tok, lens, gen, z=z, n_steps=10)
            elif flow_part == "hift":
                for _ in range(8):
                    eng.hift.inference(mel_cache["mel"], lens=None, fade=True)
            e["v_e"].record()
        hw["voc_enqueued_ms"] = round(1e3 * (time.perf_counter() - h0), 1)

    if order == "threads":  # the T3 enqueue (250 graph launches of 153 nodes: ~0.6 ms of HOST time each) on a second host thread
        import threading
        th = threading.Thread(target=t3)
        th.start()
        voc()
        th.join()
    else:
        for f in ((t3, voc) if order == "t3 first" else (voc, t3)):
            f()
    torch.cuda.synchronize()
    wall = 1e3 * (time.perf_counter() - h0)
    rel = lambda k: round(e["t0"].elapsed_time(e[k]), 1)
    return dict(order=order, flow_part=flow_part, wall_ms=round(wall, 1), t3_start=rel("t3_s"), t3_end=rel("t3_e"), voc_start=rel("v_s"), voc_end=rel("v_e"), **hw)


def report(label):
    for order in ("t3 first", "voc first", "threads"):
        r = both(order)
        r = both(order)
        print(json.dumps(dict(config=label, **r)), flush=True)


report("default kernels")
_lib.lib.cbx_set_attn_planes_version(5)
report("attention v5 only")
_lib.lib.cbx_set_attn_planes_version(5)
_lib.lib.cbx_set_planes_tile(17)
eng.t3.apply_variant(dict(eng.t3.tune, half_tiles=0, d_ks2=4, d_nw2=8), dict(eng.t3.knobs, shallow=1))
eng.t3.generate(t3c, texts, **kw)
report("co-residency kernels (attention v5, GEMM tile 17, T3 geometry)")
_lib.lib.cbx_set_attn_planes_version(0)
_lib.lib.cbx_set_planes_tile(0)
