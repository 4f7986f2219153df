import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatterbox_amd import synth
from chatterbox_amd.t3_turbo import T3TurboEngine
dev = torch.device("cuda:0")
for L, d, name in ((24, 1024, "turbo"), (12, 768, "nano")):
    eng = T3TurboEngine(synth.t3_turbo_state_dict(L, d, 0), dev)
    cond, tt = synth.t3_cond(prompt_len=375), synth.turbo_text_tokens(64)
    u = torch.rand(1, 300)
    for n in (1, 33, 249):
        ts = []
        for i in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.generate(cond, [tt], max_gen_len=n, uniforms=u[:, :n + 1], ban_eos=True, temperature=0.8, top_k=1000, top_p=0.95, repetition_penalty=1.2)
            torch.cuda.synchronize()
            if i >= 2: ts.append(1e3 * (time.perf_counter() - t0))
        print(name, "max_gen_len", n, f"{sorted(ts)[len(ts)//2]:.2f} ms", flush=True)
