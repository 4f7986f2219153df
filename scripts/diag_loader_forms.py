"""Where the time of the loader-wave forms of the plane GEMM goes (-DCBX_DIAG side library: scripts/diag_planes.sh builds it; run with CBX_LIB_PATH set to it):
cbx_gemm_pl_t.reserved0 = 1 no DMA after the prologue | 2 no ds_read / MFMA | 4 no epilogue stores | 8 loader waves at s_setprio 3 | 16 no epilogue at all; interleaved rounds."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from chatterbox_amd import ops  # noqa: E402

sys.argv = sys.argv[:1]
os.environ.setdefault("CBX_PL_TILES", "32")
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("df_micro_lib", os.path.join(os.path.dirname(__file__), "df_micro.py"))
os.environ["CBX_DF_MICRO_LIB"] = "1"
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)
DIAGS = tuple(int(d) for d in os.environ.get('CBX_DIAGS', '0,1,2,4,3,5,6,7,8').split(','))
print("# us per launch with reserved0 = " + " | ".join(str(d) for d in DIAGS), flush=True)
for tile in tuple(int(t) for t in os.environ.get('CBX_DIAG_TILES', '32,35').split(',')):
    for name, N, K, fn in m.cases:
        def mk(d):
            def f():
                ops.GEMM_DIAG = d
                fn(tile)
            return f
        us = m.interleaved([mk(d) for d in DIAGS], rounds=5)
        print(f"t{tile} {name:30s} | " + " ".join(f"{u:6.1f}" for u in us), flush=True)
ops.GEMM_DIAG = 0
