mkdir -p gpurun_out/r03b
run() { # env, flags, tag
  env $1 python bench.py $2 --steps 3 --warmup 1 --no-cpu-baseline --no-streaming --no-alt-precisions --no-fast-mode --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$3 [$1]', d['value'], d['ms_per_step'], d.get('decode_step',{}).get('ms_per_step'))"
}
for e in "X=1" "CBX_DA_U=4" "CBX_DA_NO_SPLIT=1" "CBX_DA_U=4 CBX_DA_NO_SPLIT=1" "CBX_DA_SPLIT_MIN=384"; do run "$e" "--batch 1" mtl_b1; done
for e in "X=1" "CBX_DA_U=4 CBX_DA_NO_SPLIT=1" "CBX_DA_SPLIT_MIN=384"; do run "$e" "--workload nano --batch 1" nano_b1; done
for e in "X=1" "CBX_DA_U=4 CBX_DA_NO_SPLIT=1" "CBX_DA_SPLIT_MIN=384"; do run "$e" "--workload turbo --batch 1" turbo_b1; done
