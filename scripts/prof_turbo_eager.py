"""A few eager (non-graph) T3-Turbo decode steps at batch 1 -- the target of the rocprofv3 --pmc passes for the row kernels (gemv_row_kernel,
decode_attn_parts_kernel): rocprofv3 --pmc does not survive hipGraph replays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chatterbox_amd import synth
from chatterbox_amd.t3_turbo import T3TurboEngine
dev = torch.device("cuda:0")
N = int(os.environ.get("CBX_STEPS", "8"))
eng = T3TurboEngine(synth.t3_turbo_state_dict(24, 1024, 0), dev)
toks = eng.generate(synth.t3_cond(prompt_len=375), [synth.turbo_text_tokens(64)], max_gen_len=N, uniforms=synth.rand((1, N + 1), seed=1).to(dev), ban_eos=True,
                    ban_from=6561, use_graph=False)
torch.cuda.synchronize()
print("ok", [int(t.numel()) for t in toks])
