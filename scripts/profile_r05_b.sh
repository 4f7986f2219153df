#!/bin/bash
# Round 5, call B: (1) configs[4] timed for the first time (bench.py --workload vc60 at B = 1 and B = 4) + rocprofv3 kernel stats at that shape;
# (2) the overlap diagnosis VERDICT r04 item 6 asked for: rocprofv3 kernel trace of T3 alone / flow + vocoder alone / both at once on plain streams,
# with short-lived GEMM workgroups, and on CU-masked streams (scripts/overlap_trace.py, overlap_analyse.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 150 python $R/bench.py --workload vc60 --batch 1 --steps 3 --warmup 1 > $O/bench_vc60_b1.json 2> $O/bench_vc60_b1.err
tail -c 600 $O/bench_vc60_b1.json; tail -3 $O/bench_vc60_b1.err
timeout 150 python $R/bench.py --workload vc60 --batch 4 --steps 2 --warmup 1 > $O/bench_vc60_b4.json 2> $O/bench_vc60_b4.err
python -c "import json; d=json.load(open('$O/bench_vc60_b4.json')); print('B=4', d['value'], d['ms_per_step'], d['stage_ms'])"; tail -3 $O/bench_vc60_b4.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vc -o vc -- python $R/bench.py --workload vc60 --batch 1 --steps 2 --warmup 1 > $O/bench_vc60_under_rocprof.json 2> /tmp/rocprof_vc.err
cp $(find /tmp/prof_vc -name "*kernel_stats.csv" | head -1) $O/vc60_b1_kernel_stats.csv 2>/dev/null || tail -5 /tmp/rocprof_vc.err
head -12 $O/vc60_b1_kernel_stats.csv | cut -c1-200
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ov -o ov -- python $R/scripts/overlap_trace.py $O/overlap_phases.json > $O/overlap_phases.log 2> /tmp/rocprof_ov.err
cat $O/overlap_phases.log
TR=$(find /tmp/prof_ov -name "*kernel_trace.csv" | head -1)
ls -la $TR
python $R/scripts/overlap_analyse.py $TR $O/overlap_phases.json > $O/overlap_analysis.txt 2> $O/overlap_analysis.err
tail -5 $O/overlap_analysis.err
head -c 3000 $O/overlap_analysis.txt
gzip -c $TR | head -c 30000000 > $O/overlap_kernel_trace.csv.gz
