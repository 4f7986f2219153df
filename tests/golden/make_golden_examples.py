"""The acceptance surface of SURVEY.md section 2: the reference's example scripts, recorded as TEST INPUTS (authoring container only).

    python tests/golden/make_golden_examples.py     ->  tests/golden/example_scripts.json

`/root/reference` does not exist on the GPU box, so the text of example_tts.py / example_tts_turbo.py / example_tts_nano.py / example_vc.py
is stored (with its SHA-256) the way golden vectors are: produced from the reference by this committed script, consumed by
tests/test_examples_gpu.py, which executes each script UNMODIFIED against the `chatterbox` alias package.  A CPU test re-checks the
fixture against the reference wherever the reference is present.
"""
import hashlib
import json
import os

REF = "/root/reference"
NAMES = ["example_tts.py", "example_tts_turbo.py", "example_tts_nano.py", "example_vc.py"]
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "example_scripts.json")


def collect():
    out = {}
    for n in NAMES:
        text = open(os.path.join(REF, n), encoding="utf-8").read()
        out[n] = dict(sha256=hashlib.sha256(text.encode("utf-8")).hexdigest(), text=text)
    return out


if __name__ == "__main__":
    json.dump(collect(), open(OUT, "w"), indent=1, ensure_ascii=False)
    print("wrote", OUT)
