"""Versions of the plane attention (cbx_flash_attn_planes_v) at the CFM shape, INTERLEAVED rounds (run on the GPU box).

    CBX_ROWS=16 CBX_T=1000 CBX_ATTN_VERSIONS=4,6,2 CBX_REPS=20
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CBX_DF_MICRO_LIB"] = "1"
import importlib.util  # noqa: E402

import torch  # noqa: E402

from chatterbox_amd import ops  # noqa: E402

spec = importlib.util.spec_from_file_location("df_micro_lib", os.path.join(os.path.dirname(__file__), "df_micro.py"))
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)
dev, ROWS, T, M = m.dev, m.ROWS, m.T, m.M
vers = [int(v) for v in os.environ.get("CBX_ATTN_VERSIONS", "4,6,2").split(",")]
Tp = (T + 7) // 8 * 8
qkP, attP, vtP = ops.Planes(M, 1024, dev), ops.Planes(M, 512, dev), ops.Planes(ROWS * 512, Tp, dev, zero=True)
ops.split_planes(torch.randn(M, 1024, device=dev), qkP)
ops.split_planes(torch.randn(ROWS * 512, Tp, device=dev), vtP)
lens = torch.full((ROWS,), T, dtype=torch.int32, device=dev)
fl = 4.0 * ROWS * 8 * T * T * 64


def mk(v):
    return lambda: ops.flash_attn_planes(qkP.cols(0, 512), qkP.cols(512, 512), vtP, attP, Z=ROWS, H=8, T=T, vt_sb=512 * vtP.ld, scale=0.125, key_lens=lens, version=v)


for order in (vers, vers[::-1]):
    us = m.interleaved([mk(v) for v in order], rounds=7)
    print(f"rows {ROWS} T {T} | " + " ".join(f"v{v}: {u:6.1f} us ({fl * 3 / u / 1e6 / 2500:.3f})" for v, u in zip(order, us)), flush=True)
