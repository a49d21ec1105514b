#!/bin/bash
# Round 6, lease 11: the pipelined step with PERSISTENT side-stream kernels (a fixed number of wavefronts per CU instead of a problem-sized single-wave grid that starves the
# main stream's workgroups of wave slots - lease 10's traces): tests, then wavefronts per CU x start point of the SH Adam pass, alternating against the one-stream step
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease11; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_000_canary.py tests/test_gpu_gut_step.py -q -m gpu -p no:cacheprovider -x > $OUT/tests.log 2>&1; echo "tests rc $?: $(tail -1 $OUT/tests.log)"
grep -n "FAILED\|Error\|assert" $OUT/tests.log | head -20
run() { # name, bench flags, env...
  local name=$1 flags=$2; shift 2
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-ops-route --steps 200 --warmup 20 $flags 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('[$name]', d['value'], d['ms_per_step'], {k: v['avg_ms'] for k, v in d['kernels'].items()})"
}
for r in 1 2; do
  run serial --no-pipeline X=1
  for w in 4 8 12 16; do
    run pipe_w$w "" LFS_PIPE_WAVES=$w
    run pipe_after_finish_w$w "" LFS_PIPE_WAVES=$w LFS_PIPE_START=finish
  done
done 2>&1 | tee $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in w8 after_finish_w8 after_finish_w16; do
  E="LFS_PIPE_WAVES=8"
  [ $v = after_finish_w8 ] && E="LFS_PIPE_WAVES=8 LFS_PIPE_START=finish"
  [ $v = after_finish_w16 ] && E="LFS_PIPE_WAVES=16 LFS_PIPE_START=finish"
  env $E rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/trace_$v -o t -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-ops-route > $REPO/$OUT/trace_$v.log 2>&1
done
cd $REPO; python tools/step_timeline.py gpurun_out/r6_lease11/trace_w8 gpurun_out/r6_lease11/trace_after_finish_w8 gpurun_out/r6_lease11/trace_after_finish_w16 | tee $OUT/timelines.txt
