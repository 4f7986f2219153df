"""Host cost of one launch through the Python wrappers (ctypes struct fill + call): tiny kernels issued back to back, wall time per launch
with the queue never empty (the GPU side of these kernels is ~2-3 us)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chatterbox_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(64, 256, device=dev)
w, b = torch.ones(256, device=dev), torch.zeros(256, device=dev)
yP = ops.Planes(64, 256, dev)
hP, wP, oP = ops.split_planes(x), ops.split_planes(torch.randn(256, 256, device=dev)), ops.Planes(64, 256, dev)
for name, fn in (("layernorm_planes", lambda: ops.layernorm_planes(x, w, b, yP)), ("linear_planes", lambda: ops.linear_planes(hP, wP, outp=oP)),
                 ("torch add_", lambda: x.add_(1.0))):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3000):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:18s} host {1e6 * (t1 - t0) / 3000:6.2f} us per launch, with drain {1e6 * (t2 - t0) / 3000:6.2f} us", flush=True)
