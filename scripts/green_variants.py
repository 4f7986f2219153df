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
from chatterbox_amd.t3 import T3Engine
# every geometry the autotuner can compose from the engines' default and from the frozen round-3 base, plus what is already listed
cands, seen = [], set()
for t, k in at.composed_candidates(T3Engine._TUNE, at.LIB_KNOBS) + at.composed_candidates(at.BASE_TUNE, at.BASE_KNOBS) + [(dict(c), {}) for c in sorted(at.green_variants()) if os.environ.get("CBX_GREEN_RECHECK") == "1"]:
    c = at.canon(t, k)
    if c not in seen:
        seen.add(c)
        cands.append(dict(c))
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
