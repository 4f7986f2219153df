import sys, time, torch, traceback
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatterbox_amd import synth
import chatterbox_amd.t3_turbo as TT
dev = torch.device("cuda:0")
eng = TT.T3TurboEngine(synth.t3_turbo_state_dict(24, 1024, 0), dev)
cond, tt = synth.t3_cond(prompt_len=375), synth.turbo_text_tokens(64)
u = torch.rand(1, 300)
kw = dict(max_gen_len=1, uniforms=u[:, :2], ban_eos=True, temperature=0.8, top_k=1000, top_p=0.95, repetition_penalty=1.2)
for _ in range(3): eng.generate(cond, [tt], **kw)
torch.cuda.synchronize()
orig = torch.tensor
log = []
def timed(*a, **k):
    t0 = time.perf_counter(); r = orig(*a, **k); dt = 1e3 * (time.perf_counter() - t0)
    log.append((round(dt, 3), traceback.extract_stack(limit=2)[0].lineno))
    return r
torch.tensor = timed
opc = eng._prefill_c
def pc(*a):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); opc(*a); e1.record(); eng._ev = (e0, e1)
eng._prefill_c = pc
t0 = time.perf_counter(); eng.generate(cond, [tt], **kw); torch.cuda.synchronize(); print("generate", 1e3 * (time.perf_counter() - t0))
print("prefill GPU ms", eng._ev[0].elapsed_time(eng._ev[1]))
print("torch.tensor calls (ms, line):", log)
