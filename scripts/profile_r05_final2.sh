#!/bin/bash
# Round 5, last call: the whole -m gpu suite (no -x) on the round's LAST commit (after the test fix that followed the first final call)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/final2
mkdir -p $O
cd $R
timeout 1100 python -m pytest tests -q -m gpu -rfEs -p no:cacheprovider --junitxml=$O/pytest_gpu.xml > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
python - <<'PY' > $O/gpu_tests_per_test.txt
import xml.etree.ElementTree as ET, os
t = ET.parse(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r05/final2/pytest_gpu.xml")).getroot()
for c in t.iter("testcase"):
    st = "FAILED" if c.find("failure") is not None or c.find("error") is not None else "skipped" if c.find("skipped") is not None else "passed"
    print(f"{st:8s} {float(c.get('time', 0)):8.2f}s  {c.get('classname')}::{c.get('name')}")
PY
grep -c "^passed" $O/gpu_tests_per_test.txt; grep -v "^passed" $O/gpu_tests_per_test.txt | head
