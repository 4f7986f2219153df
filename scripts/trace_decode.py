"""Launch timeline of the Llama T3 decode step (B = 8, 30 layers): see scripts/trace_lib.py for what is measured.  scripts/trace_decode.sh run"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from trace_lib import *  # noqa: F401,F403  (torch, np, json, synth, lib, buf, cnt, CAP, collect, analyse, out_dir, dev, L, B, N)
from chatterbox_amd.t3 import T3Engine

eng = T3Engine(synth.t3_state_dict(L, 0), dev, n_layers=L)
texts = [synth.text_tokens(64, seed=b) for b in range(B)]
u = synth.rand((B, N), seed=1).to(dev)
kw = dict(max_new_tokens=N, uniforms=u, ban_eos=True, ban_from=6561)
variants = [("default", {}, {})]
if os.environ.get("CBX_TRACE_TUNE"):  # e.g. CBX_TRACE_TUNE="qkv_ks=4,qkv_ct=3,head_ct=2,da_pipe=7"
    from chatterbox_amd.autotune import split_variant
    tv, kv = split_variant({k: int(x) for k, x in (kv.split("=") for kv in os.environ["CBX_TRACE_TUNE"].split(","))})
    variants = [(os.environ["CBX_TRACE_TUNE"], tv, kv)]
if os.environ.get("CBX_TRACE_VARIANTS", "1") == "1":
    variants += [("da_pipe=7,pre_epi=1", {}, dict(da_pipe=7, pre_epi=1)), ("qkv_tc=12,od_tc=4,d_ks2=1,d_nw2=8", dict(qkv_tc=12, od_tc=4, d_ks2=1, d_nw2=8), {})]
result = {}
with open(os.path.join(out_dir, "trace_decode.txt"), "w") as f:
    for label, tune, knobs in variants:
        kn = dict(eng.knobs, da_pipe=0, pre_epi=0, deep=0, da_u=4)
        kn.update(knobs)
        eng.apply_variant(dict(T3Engine._TUNE, **tune), kn)
        for graph in (True, False):
            h = eng.generate(synth.t3_cond(), texts, async_mode=True, run_steps=N - 8, use_graph=graph, **kw)  # context ~ 100 + 150
            torch.cuda.synchronize()
            assert lib.cbx_trace_set(buf.data_ptr(), cnt.data_ptr(), CAP) == 0
            cnt.zero_()
            eng.advance(h, 4)
            rec = collect()
            assert lib.cbx_trace_set(None, None, 0) == 0
            np.save(os.path.join(out_dir, f"trace_{label.replace(',', '_').replace('=', '')}_{'graph' if graph else 'eager'}.npy"), rec)
            result[f"{label} | {'hipGraph' if graph else 'eager'}"] = analyse(rec, f"{label}, B = {B}, {L} layers, {'hipGraph replays' if graph else 'eager launches'}, 4 token steps", f)
            f.flush()
    json.dump(result, open(os.path.join(out_dir, "trace_decode.json"), "w"), indent=1)
print(open(os.path.join(out_dir, "trace_decode.txt")).read())
