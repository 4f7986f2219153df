"""green_variants.json (scripts/green_variants.py, a GPU run) -> chatterbox_amd/decode_green.json (the committed allow-list).
    python scripts/write_green.py gpurun_out/r04/fifth/green_variants.json "round 4, fifth GPU call" """
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = json.load(open(sys.argv[1]))
label = sys.argv[2] if len(sys.argv) > 2 else "a GPU run"
doc = {"what": "T3 decode geometries whose hardware tests passed on an MI355X, each as the keys in which the FULL geometry (T3Engine.tune + launch knobs) differs from the "
               "frozen round-3 base (autotune.BASE_TUNE / BASE_KNOBS): the ONLY geometries bench.py may run or adopt (autotune.green_variants).  Written by "
               "scripts/green_variants.py + scripts/write_green.py from a GPU run, re-checked by tests/test_zz_abi_v9_gpu.py::test_green_variant_is_bit_identical_and_samples_the_reference_tokens.",
       "source": f"{label}: {len(g['green'])} green, {len(g['red'])} red, {len(g.get('not_run', []))} not run on {g.get('device')}",
       "green": g["green"]}
json.dump(doc, open(os.path.join(ROOT, "chatterbox_amd", "decode_green.json"), "w"), indent=0)
print(doc["source"])
