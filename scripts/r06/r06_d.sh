#!/bin/bash
# round 6, call d: the whole hardware suite on the DPP / permlane xor-lane primitive (cbx_xor_lane replaces every ds_bpermute butterfly: bit-identity
# against the goldens is the check), then the three bench lines
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_d
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
timeout 600 python bench.py --steps 8 --warmup 2 > $O/bench_mtl.json 2> $O/bench_mtl.err
python - <<PY
import json
d=json.loads(open("$O/bench_mtl.json").read().strip().splitlines()[-1])
print("mtl", d["value"], d["stage_ms"], d["decode_step"]["ms_per_step"], d.get("value_serial"), d.get("value_bf16x6"), d.get("p50_first_audio_latency_ms_pipelined"), d.get("p50_first_audio_latency_ms_serial"), d.get("streaming"))
PY
for wl in turbo nano; do
  CBX_TURBO_TUNE="row_splits=16,row_chunks=2" timeout 300 python bench.py --workload $wl --batch 1 --steps 6 --warmup 2 --no-cpu-baseline --no-streaming > $O/bench_${wl}_b1.json 2> $O/bench_${wl}_b1.err
  python - <<PY
import json
d=json.loads(open("$O/bench_${wl}_b1.json").read().strip().splitlines()[-1])
print("$wl", d["value"], d["stage_ms"], d["decode_step"]["ms_per_step"], d["decode_step"]["frac"])
PY
done
