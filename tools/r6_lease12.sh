#!/bin/bash
# Round 6, lease 12: register footprint of the side stream's Adam kernel (91 VGPRs at depth 4, 63 at depth 2, 64 + spills under a forced budget) x wavefronts per CU
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease12; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gut_step.py -q -m gpu -p no:cacheprovider -x -k pipelined > $OUT/tests.log 2>&1; echo "tests rc $?: $(tail -1 $OUT/tests.log)"
run() { # name, lib variant, bench flags, env...
  local name=$1 lib=$2 flags=$3; shift 3
  local L="X=1"; [ $lib != default ] && L="LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$lib.so"
  env $L "$@" timeout 300 python bench.py --no-cpu-baseline --no-ops-route --steps 200 --warmup 20 $flags 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('[$name]', d['value'], d['ms_per_step'], {k: v['avg_ms'] for k, v in d['kernels'].items()})"
}
for r in 1 2; do
  run serial default --no-pipeline X=1
  for w in 8 12; do run depth4_w$w default "" LFS_PIPE_WAVES=$w; done
  for w in 8 12 16 24; do run depth2_w$w depth2 "" LFS_PIPE_WAVES=$w; done
  for w in 8 12; do run adam64_w$w adam64 "" LFS_PIPE_WAVES=$w; done
done 2>&1 | tee $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in depth4_w8 depth2_w12 depth2_w16; do
  E="LFS_PIPE_WAVES=8"
  [ $v = depth2_w12 ] && E="LFS_PIPE_WAVES=12 LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_depth2.so"
  [ $v = depth2_w16 ] && E="LFS_PIPE_WAVES=16 LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_depth2.so"
  env $E rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/trace_$v -o t -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-ops-route > $REPO/$OUT/trace_$v.log 2>&1
done
cd $REPO; python tools/step_timeline.py $OUT/trace_depth4_w8 $OUT/trace_depth2_w12 $OUT/trace_depth2_w16 | tee $OUT/timelines.txt
