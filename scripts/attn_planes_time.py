"""flash_attn_planes at the CFM bench shape (16 rows x T = 1000, 8 heads), versions 2 and 4: us per launch (hipGraph-free event timing, 50 launches).
A/B of a kernel edit on one box: run once per library (CBX_LIB_PATH)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chatterbox_amd import ops

dev = torch.device("cuda:0")
ROWS, T = 16, 1000
M, Tp = ROWS * T, (T + 7) // 8 * 8
qkP, attP, vtP = ops.Planes(M, 1024, dev), ops.Planes(M, 512, dev), ops.Planes(ROWS * 512, Tp, dev, zero=True)
ops.split_planes(torch.randn(M, 1024, device=dev), qkP)
ops.split_planes(torch.randn(ROWS * 512, Tp, device=dev), vtP)
lens = torch.full((ROWS,), T, dtype=torch.int32, device=dev)
for ver in (2, 4, 2, 4):
    ops.lib.cbx_set_attn_planes_version(ver)
    f = lambda: ops.flash_attn_planes(qkP.cols(0, 512), qkP.cols(512, 512), vtP, attP, Z=ROWS, H=8, T=T, vt_sb=512 * vtP.ld, scale=0.125, key_lens=lens)
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        f()
    e1.record()
    torch.cuda.synchronize()
    print(f"{os.environ.get('CBX_LIB_PATH', 'libcbx_hip.so').split('/')[-1]} flash_attn_planes v{ver}: {e0.elapsed_time(e1) * 10:.1f} us / launch", flush=True)
