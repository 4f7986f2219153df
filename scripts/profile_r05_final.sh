#!/bin/bash
# Round 5, FINAL call (on the round's last commit): the whole -m gpu suite (no -x, per-test results), smoke(), the driver's bench command
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/final
mkdir -p $O
cd $R
timeout 1100 python -m pytest tests -q -m gpu -rfEs -p no:cacheprovider --junitxml=$O/pytest_gpu.xml > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
python - <<'PY' > $O/gpu_tests_per_test.txt
import xml.etree.ElementTree as ET, os
t = ET.parse(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r05/final/pytest_gpu.xml")).getroot()
for c in t.iter("testcase"):
    st = "FAILED" if c.find("failure") is not None or c.find("error") is not None else "skipped" if c.find("skipped") is not None else "passed"
    print(f"{st:8s} {float(c.get('time', 0)):8.2f}s  {c.get('classname')}::{c.get('name')}")
PY
grep -c passed $O/gpu_tests_per_test.txt; grep -v "^passed" $O/gpu_tests_per_test.txt | head
timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -2
CBX_BENCH_VERBOSE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench.err
tail -2 $O/bench.err | cut -c1-200
python -c "
import json; d=json.load(open('$O/bench_steps20_warmup5.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'], 'schedule', d['schedule'], 'p50 lat', d['p50_first_audio_latency_ms'])
print('other', d.get('other_schedule')); print('stage_ms', d['stage_ms']); print('decode', d['decode_step']['ms_per_step'], d['decode_step']['frac'], d.get('decode_step_in_throughput_schedule'))
print('t3_geometry', d['t3_geometry']); print('roofline', d['roofline']['kernel'][:40], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline'].get('traffic'))
for r in d['roofline_secondary']: print('   ', r['kernel'][:40], r['frac'], r['avg_launch_us'], r['share_of_step'], r.get('traffic'))
print('parity', d.get('parity')); print('cpu', d.get('cpu_baseline')); print('alt', d.get('audio_s_per_wall_s_at_other_precisions')); print('streaming', d.get('streaming'))
"
