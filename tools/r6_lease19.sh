#!/bin/bash
# Round 6, lease 19: the library that ships (fused tail with its coefficient rows kept in registers, two-launch count + scan): whole GPU suite, the driver's command x3,
# A/B against the three-pass step, the round's rocprofv3 evidence (trace + 4 PMC passes of the driver's command, headline path only)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease19; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc $?: $(tail -1 $OUT/tests.log)"
grep -n "^FAILED\|^ERROR" $OUT/tests.log | head -20
for r in 1 2 3; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_default_$r.json
  python -c "
import json; d = json.load(open('$OUT/bench_default_$r.json')); print('driver command:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['config'].get('step_form'), d.get('ops_route', {}).get('ms_per_step'))"
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-fused-tail --no-cpu-baseline --no-ops-route 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('three-pass step :', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['config'].get('step_form'))"
done 2>&1 | tee $OUT/driver_command.txt
bash tools/profile.sh r06_lease19 > $OUT/profile.log 2>&1; head -30 gpurun_out/prof_r06_lease19/summary.txt | cut -c1-180
