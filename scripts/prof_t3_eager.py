"""A few eager (non-graph) T3 decode steps at the bench shape -- the target of the rocprofv3 --pmc passes for the gemv kernels
(rocprofv3 --pmc does not survive hipGraph replays)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chatterbox_amd import synth
from chatterbox_amd.t3 import T3Engine
dev = torch.device("cuda:0")
B, N = 8, int(os.environ.get("CBX_STEPS", "12"))
eng = T3Engine(synth.t3_state_dict(30, 0), dev)
texts = [synth.text_tokens(64, seed=b) for b in range(B)]
toks = eng.generate(synth.t3_cond(prompt_len=150), texts, max_new_tokens=N, uniforms=synth.rand((B, N), seed=1).to(dev), ban_eos=True,
                    ban_from=6561, use_graph=False)
torch.cuda.synchronize()
print("ok", [int(t.numel()) for t in toks])
