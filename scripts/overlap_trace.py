"""Why does the throughput schedule (T3 of batch k + 1 beside flow + vocoder of batch k) barely overlap?  (run on the GPU box under
`rocprofv3 --kernel-trace`; scripts/overlap_analyse.py reads the trace)

Phases, each preceded by a 120 ms host sleep on an idle GPU (the analysis splits the trace at idle gaps >= 60 ms) and announced on stdout in order:
for every configuration  [T3 alone, flow + vocoder alone, both at once].  Configurations: plain streams (T3 high priority); plain streams with the
plane GEMMs one tile per workgroup (short-lived workgroups); CU-masked streams 128 / 128 (`block` mask layout).  Shapes: the bench's (B = 8, 30 layers,
T = 1000), CBX_OV_TOKENS tokens per utterance for T3 (default 100: the trace stays small; the flow always runs 250 tokens)."""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chatterbox_amd import _lib, synth
from chatterbox_amd.engine import ChatterboxEngine, drop_invalid_tokens


def hip():
    path = next(line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line)
    lib = ctypes.CDLL(path)
    lib.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    lib.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
    return lib


def masked_stream(dev, bits, n=256):
    words = [0] * ((n + 31) // 32)
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    arr = (ctypes.c_uint32 * len(words))(*words)
    s = ctypes.c_void_p()
    rc = hip().hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    assert rc == 0 and s.value, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


dev = torch.device("cuda:0")
torch.cuda.set_device(0)
L = int(os.environ.get("CBX_AB_LAYERS", "30"))
NT = int(os.environ.get("CBX_OV_TOKENS", "100"))
eng = ChatterboxEngine(synth.t3_state_dict(L, 0), synth.s3gen_state_dict(0), dev, n_t3_layers=L)
B, N = 8, 250
t3c, gen = synth.t3_cond(prompt_len=150), synth.s3gen_ref()
texts = [synth.text_tokens(64, seed=b) for b in range(B)]
g = torch.Generator(device=dev).manual_seed(1)
u = torch.rand(B, N, generator=g, device=dev)
T = 2 * (gen["prompt_token"].shape[1] + N)
z = torch.randn(B, T, 80, generator=g, device=dev)
kw = dict(max_new_tokens=N, uniforms=u, ban_eos=True, ban_from=6561)
toks = eng.t3.generate(t3c, texts, **kw)
st = [drop_invalid_tokens(t) for t in toks]
eng.vocode(st, gen, z=z, drop_last_token=True)
kw_t = dict(max_new_tokens=NT, uniforms=u[:, :NT].contiguous(), ban_eos=True, ban_from=6561)
eng.t3.generate(t3c, texts, **kw_t)
torch.cuda.synchronize()
phases = []


def run(sa, sb, what, label):
    torch.cuda.synchronize()
    time.sleep(0.12)
    t0 = time.perf_counter()
    if what in ("t3", "both"):
        with torch.cuda.stream(sa):
            eng.t3.generate(t3c, texts, async_mode=True, **kw_t)
    if what in ("voc", "both"):
        with torch.cuda.stream(sb):
            eng.vocode(st, gen, z=z, drop_last_token=True, sync=False)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0)
    phases.append(dict(config=label, what=what, wall_ms=round(ms, 1)))
    print(json.dumps(phases[-1]), flush=True)


def trio(sa, sb, label):
    for what in ("t3", "voc", "both"):
        run(sa, sb, what, label)


plain = (torch.cuda.Stream(device=dev, priority=-1), torch.cuda.Stream(device=dev))
trio(*plain, "plain streams, T3 high priority")
_lib.lib.cbx_set_planes_persist(0)
trio(*plain, "plain streams, plane GEMMs one tile per workgroup")
_lib.lib.cbx_set_planes_persist(1)
a = list(range(128))
sa, sb = masked_stream(dev, a), masked_stream(dev, [i for i in range(256) if i not in set(a)])
trio(sa, sb, "CU masks: T3 bits 0-127, flow + vocoder bits 128-255")
# the same partition the other way round (which stream owns which half must not matter if the masks are honoured)
trio(sb, sa, "CU masks: T3 bits 128-255, flow + vocoder bits 0-127")
if len(sys.argv) > 1:
    json.dump(phases, open(sys.argv[1], "w"), indent=1)
