#!/bin/bash
# Round 6, lease 32: how often does the driver's 20-step window catch a host stall, with Python's cycle collector on (LFS_BENCH_GC=1: the bench as it was) and off inside the
# timed region (the default now)? 12 runs each, alternating; the step's per-attempt log is on the line (config.*), every kernel duration too.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease32; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
for r in $(seq 1 ${RUNS:-12}); do for g in off on; do
  if [ $g = on ]; then export LFS_BENCH_GC=1; else unset LFS_BENCH_GC; fi
  timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-ops-route 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('[gc $g]', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done 2>&1 | tee $OUT/gc_on_off.txt
unset LFS_BENCH_GC
