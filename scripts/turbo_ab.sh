mkdir -p gpurun_out/r03b
timeout 600 python -m pytest tests -x -q -m gpu -k "turbo or nano" > gpurun_out/r03b/t_turbo.log 2>&1; tail -3 gpurun_out/r03b/t_turbo.log
for tune in "" "d_ks=4,d_nw=8,o_nw=8,half_tiles=0" "d_ks=2,d_nw=8,o_nw=8,half_tiles=1" "d_ks=4,d_nw=8,o_nw=8,half_tiles=1" "d_ks=2,d_nw=16,o_nw=16,half_tiles=1" "d_ks=2,d_nw=16,o_nw=8,half_tiles=0"; do
  CBX_TURBO_TUNE="$tune" python bench.py --workload turbo --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-streaming --no-alt-precisions --no-fast-mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('turbo [$tune]', d['value'], d['ms_per_step'], d.get('decode_step',{}).get('ms_per_step'))"
done
