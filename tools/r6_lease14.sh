#!/bin/bash
# Round 6, lease 14 (13 again, on the shared contraction-free finish_geometry, with the whole GPU suite): the fused tail (lfs_gut_train_step_ex / gut_tail_kernel: SH backward + six Adam updates + next view's SH colours in one launch) - tests, alternating A/B
# against the three-pass step on one box, the driver's command, a kernel timeline
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease14; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider -x > $OUT/tests.log 2>&1; echo "tests rc $?: $(tail -1 $OUT/tests.log)"
grep -n "FAILED\|Error\|assert" $OUT/tests.log | head -20
run() { # name, bench flags
  local name=$1 flags=$2
  timeout 300 python bench.py --no-cpu-baseline --no-ops-route --steps 300 --warmup 20 $flags 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('[$name]', d['value'], d['ms_per_step'], d['config'].get('step_form'), {k: v['avg_ms'] for k, v in d['kernels'].items()})"
}
for r in 1 2 3; do
  run three_pass --no-fused-tail
  run fused_tail ""
done 2>&1 | tee $OUT/ab.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_default.json
python -c "
import json; d = json.load(open('$OUT/bench_default.json')); print('driver command:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('step_form'), d['cpu_baseline'].get('parity_vs_oracle', {}).get('grad_rel_l2'), d.get('ops_route'))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/trace_fused_tail -o t -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-ops-route > $REPO/$OUT/trace.log 2>&1
cd $REPO; python tools/step_timeline.py $OUT/trace_fused_tail | tee $OUT/timeline.txt
