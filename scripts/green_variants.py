"""Which T3 decode geometries are green ON THIS HARDWARE?  (run on the GPU box; writes <out>/green_variants.json)

Every candidate the autotuner can compose (tile geometry x attention knobs x epilogue prefetch) goes through the body of
tests/test_zz_abi_v9_gpu.py::test_green_variant_is_bit_identical_and_samples_the_reference_tokens -- logits over the ragged probe contexts
{1 .. 640} at B = 8 and B = 1 bit-identical to the built-in geometry's (2e-4 of the logit scale for the geometries that sum the down projection
in another order), golden tokens of the reference through the hipGraph path.  The passing ones are what chatterbox_amd/decode_green.json may list."""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import test_zz_abi_v9_gpu as T
from chatterbox_amd import autotune as at

out_dir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r04/first"
os.makedirs(out_dir, exist_ok=True)
dev = torch.device("cuda:0")
TILES = (dict(), dict(qkv_tc=12), dict(od_tc=4), dict(qkv_tc=12, od_tc=4), dict(od_tc=4, d_ks2=1, d_nw2=8), dict(qkv_tc=12, od_tc=4, d_ks2=1, d_nw2=8))
ATTN = (dict(),) + at.ATTN_VARIANTS
EPI = (dict(), dict(pre_epi=1))
singles = [dict(t) for t in at.TILE_VARIANTS[1:]] + [dict(a) for a in at.ATTN_VARIANTS] + [dict(e) for e in at.EPI_VARIANTS]
combos = [dict(t, **a, **e) for t in TILES for a in ATTN for e in EPI]
seen, cands = set(), []
for v in singles + combos:
    k = at.canon(v)
    if k and k not in seen:
        seen.add(k)
        cands.append(dict(k))
budget = float(os.environ.get("CBX_GREEN_BUDGET_S", "240"))
t0 = time.perf_counter()
green, red, skipped = [], [], []
for v in cands:
    if time.perf_counter() - t0 > budget:
        skipped.append(v)
        continue
    try:
        T.test_green_variant_is_bit_identical_and_samples_the_reference_tokens(dev, v)
        green.append(v)
    except AssertionError as e:
        red.append(dict(variant=v, why=str(e)[:300]))
    except Exception:
        red.append(dict(variant=v, why=traceback.format_exc()[-300:]))
    print(f"{'GREEN' if green and green[-1] is v else 'red  '} {v}", flush=True)
doc = dict(green=green, red=red, not_run=skipped, seconds=round(time.perf_counter() - t0, 1), device=torch.cuda.get_device_name(0))
json.dump(doc, open(os.path.join(out_dir, "green_variants.json"), "w"), indent=1)
print(f"{len(green)} green, {len(red)} red, {len(skipped)} not run in {doc['seconds']} s")
