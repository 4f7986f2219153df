"""Does partitioning the CUs between the T3 decode and the flow + vocoder make the pipelined schedule overlap?  (run on the GPU box)
MEASURED (round 4, profiles/r04_cu_mask_pipelined_rejected.log): NO -- plain streams: T3 303 ms, flow + HiFT 210 ms, both at once 476 ms; with the CUs split
128 / 128 (hipExtStreamCreateWithCUMask) T3 alone 391 ms, flow + HiFT alone 300 ms, both at once 561 ms: worse than plain streams.  Kept as the script
that measured it; nothing of it is in the package.
For each partition (CUs of the T3 stream, mask layout): T3 alone on its stream, flow + HiFT alone on theirs, both at once -- at the bench shape
(B = 8, 250 tokens, 30 layers).  Plain streams (today's synthesize_pipelined) as the baseline."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chatterbox_amd import synth
from chatterbox_amd.engine import ChatterboxEngine


class streams:  # (hipExtStreamCreateWithCUMask wrapped for torch: experiment-only, see the result below)
    pass


import ctypes

import os

import torch

_HIP = None


def _s__s__hip():
    """The libamdhip64 that torch already loaded (a second copy of the runtime would not know torch's streams): found through /proc/self/maps."""
    global _HIP
    if _HIP is None:
        path = None
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        if path is None:
            raise RuntimeError("libamdhip64 is not loaded in this process (import torch with ROCm first)")
        lib = ctypes.CDLL(path)
        lib.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
        lib.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
        _HIP = lib
    return _HIP


def _s_cu_mask_words(bits, n_cus=256):
    """32-bit mask words (bit i = CU i enabled) for an iterable of CU indices."""
    words = [0] * ((n_cus + 31) // 32)
    for b in bits:
        if 0 <= b < n_cus:
            words[b // 32] |= 1 << (b % 32)
    return words


def _s_masked_stream(device, bits, n_cus=None):
    """A torch stream on `device` whose kernels may only run on the CUs listed in `bits`.  The stream lives for the rest of the process."""
    dev = torch.device(device)
    idx = torch.cuda.current_device() if dev.index is None else dev.index
    n = n_cus or torch.cuda.get_device_properties(idx).multi_processor_count
    words = _s_cu_mask_words(bits, n)
    assert any(words), "empty CU mask"
    arr = (ctypes.c_uint32 * len(words))(*words)
    s = ctypes.c_void_p()
    with torch.cuda.device(idx):
        rc = _s__hip().hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    if rc != 0 or not s.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed (rc = {rc})")
    return torch.cuda.ExternalStream(s.value, device=torch.device("cuda", idx))


def _s_partition(device, n_first, layout="interleave8", n_cus=None):
    """Two disjoint masked streams: (n_first CUs, the rest).  layout "block": the first n_first mask bits; "interleave8": n_first / 8 of every
    group of n_cus / 8 consecutive bits (if mask bits are XCD-major this gives both streams CUs on every XCD; if they are interleaved over the
    XCDs, "block" does).  Which one is right for a kernel mix is a measurement (scripts/cu_mask_ab.py)."""
    dev = torch.device(device)
    idx = torch.cuda.current_device() if dev.index is None else dev.index
    n = n_cus or torch.cuda.get_device_properties(idx).multi_processor_count
    if layout == "block":
        a = list(range(n_first))
    elif layout == "interleave8":
        g, k = n // 8, n_first // 8
        a = [x * g + j for x in range(8) for j in range(k)]
    elif layout == "stride":  # every (n / n_first)-th bit
        step = n / float(n_first)
        a = sorted({int(i * step) for i in range(n_first)})
    else:
        raise ValueError(layout)
    sa = set(a)
    b = [i for i in range(n) if i not in sa]
    return _s_masked_stream(dev, a, n), _s_masked_stream(dev, b, n)

streams.partition = staticmethod(_s_partition)

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
toks = eng.t3.generate(t3c, texts, **kw)  # warm-up + the tokens the vocoder works on
from chatterbox_amd.engine import drop_invalid_tokens
st = [drop_invalid_tokens(t) for t in toks]
eng.vocode(st, gen, z=z, drop_last_token=True)
torch.cuda.synchronize()
nc = torch.cuda.get_device_properties(0).multi_processor_count
print("CUs:", nc, flush=True)


def run(sa, sb, what):
    """what: 't3', 'voc' or 'both' -> wall seconds (best of 2)"""
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h = None
        if what in ("t3", "both"):
            with torch.cuda.stream(sa):
                h = eng.t3.generate(t3c, texts, async_mode=True, **kw)
        if what in ("voc", "both"):
            with torch.cuda.stream(sb):
                eng.vocode(st, gen, z=z, drop_last_token=True, sync=False)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


rows = []
plain = (torch.cuda.Stream(device=dev, priority=-1), torch.cuda.Stream(device=dev))
r = dict(config="plain streams (T3 high priority)", t3_ms=round(1e3 * run(*plain, "t3"), 1), voc_ms=round(1e3 * run(*plain, "voc"), 1), both_ms=round(1e3 * run(*plain, "both"), 1))
rows.append(r)
print(r, flush=True)
for n_t3, layout in ((128, "block"), (128, "interleave8"), (128, "stride"), (96, "interleave8"), (96, "block"), (64, "interleave8"), (160, "interleave8"), (112, "interleave8")):
    try:
        sa, sb = streams.partition(dev, n_t3, layout, nc)
        r = dict(config=f"T3 on {n_t3} CUs, rest {nc - n_t3}, mask layout {layout}", t3_ms=round(1e3 * run(sa, sb, "t3"), 1), voc_ms=round(1e3 * run(sa, sb, "voc"), 1),
                 both_ms=round(1e3 * run(sa, sb, "both"), 1))
    except Exception as e:
        r = dict(config=f"{n_t3} / {layout}", error=f"{type(e).__name__}: {e}"[:200])
    rows.append(r)
    print(r, flush=True)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
