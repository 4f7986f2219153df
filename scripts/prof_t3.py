"""Time T3 decode: eager launches vs hipGraph replay (run on the GPU box)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chatterbox_amd import synth
from chatterbox_amd.t3 import T3Engine

L = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N = 64
dev = torch.device("cuda:0")
eng = T3Engine(synth.t3_state_dict(L, 0), dev)
texts = [synth.text_tokens(64, seed=b) for b in range(B)]
u = synth.rand((B, N), seed=1)
for use_graph in (False, True, True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    toks = eng.generate(synth.t3_cond(), texts, max_new_tokens=N, uniforms=u, ban_eos=True, ban_from=6561, use_graph=use_graph)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"L={L} B={B} graph={use_graph}: total {t1-t0:.3f}s -> {(t1-t0)/N*1e3:.2f} ms/step (incl. prefill)")
st = list(eng._state.values())[0]
g = st["graph"]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): g.replay()
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"pure graph replay: {(t1-t0)/50*1e3:.3f} ms/step")
t0 = time.perf_counter()
for _ in range(20): eng._decode_step(st)
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"eager decode step: {(t1-t0)/20*1e3:.3f} ms/step")
