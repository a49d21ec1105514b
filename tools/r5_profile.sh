#!/bin/bash
# Round 5: the rocprofv3 evidence of the library that ships - kernel trace + stats, four PMC passes (tools/profile.sh), the launch sequence of one step, the default bench line
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r5_profile; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
bash tools/profile.sh r05 > $OUT/profile.log 2>&1
cp gpurun_out/prof_r05/summary.txt $OUT/summary_trace_and_pmc.txt
python tools/step_sequence.py gpurun_out/prof_r05/trace > $OUT/step_launch_sequence.txt 2>&1; tail -22 $OUT/step_launch_sequence.txt
f=$(find gpurun_out/prof_r05/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_bench_synb.csv
cd $REPO && timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1800 $OUT/bench_default.json
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --steps 300 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('[300 steps]', d['value'], d['ms_per_step'], d['roofline']['frac'], {k: v['avg_ms'] for k, v in d['kernels'].items()})"; done | tee $OUT/bench_300steps.txt
