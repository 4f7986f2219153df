"""Launch timeline of the GPT-2 T3 decode step of Turbo (batch 1, 24 layers, d = 1024) / Nano (CBX_TRACE_NANO=1: 12 layers, d = 768): eager token steps
on the CBX_TRACE side build.  CBX_LIB_PATH=chatterbox_amd/build/libcbx_hip_trace.so python scripts/trace_decode_turbo.py [out_dir]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from trace_lib import *  # noqa: F401,F403
from chatterbox_amd.t3_turbo import T3TurboEngine

nano = os.environ.get("CBX_TRACE_NANO") == "1"
d, nl = (768, 12) if nano else (1024, 24)
eng = T3TurboEngine(synth.t3_turbo_state_dict(nl, d, 0), dev, n_layers=nl)
for kv in filter(None, os.environ.get("CBX_TRACE_TUNE", "").split(",")):
    k, v = kv.split("=")
    (eng.knobs if k in eng.knobs else eng.tune)[k] = int(v)
texts = [synth.turbo_text_tokens(64, seed=0)]
n = 120
u = synth.rand((1, n + 1), seed=1).to(dev)
eng.generate(synth.t3_cond(prompt_len=375), texts, max_gen_len=n, uniforms=u, ban_eos=True, ban_from=6561, temperature=0.8, top_k=1000, top_p=0.95, repetition_penalty=1.2)
st = next(iter(eng._state.values()))
torch.cuda.synchronize()
with open(os.path.join(out_dir, "trace_decode_turbo.txt"), "a") as f:
    for graph in (True, False):
        assert lib.cbx_trace_set(buf.data_ptr(), cnt.data_ptr(), CAP) == 0
        cnt.zero_()
        for _ in range(4):
            if graph and st["graph"] is not None:
                st["graph"].replay()
            else:
                eng._decode_step(st)
        rec = collect()
        assert lib.cbx_trace_set(None, None, 0) == 0
        analyse(rec, f"{'Nano' if nano else 'Turbo'} batch 1, {nl} layers, tune {eng.tune} knobs {eng.knobs}, {'hipGraph replays' if graph else 'eager launches'}, 4 token steps", f)
print(open(os.path.join(out_dir, "trace_decode_turbo.txt")).read())
