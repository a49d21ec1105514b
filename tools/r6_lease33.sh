#!/bin/bash
# Round 6, lease 33: the dominant kernel timed through its dispatch packet (hipExtLaunchKernelGGL with start / stop events, LFS_PROF_EXT_LAUNCH) instead of two
# hipEventRecord around its launch: the driver's command, alternating with r6evrec (the same source, -DLFS_PROF_EXT_LAUNCH=0); a trace of both for the gaps; step tests
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease33; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
timeout 900 python -m pytest tests/test_gpu_000_canary.py tests/test_gpu_gut_step.py tests/test_gpu_raster.py -q -m gpu -p no:cacheprovider -x > $OUT/tests.log 2>&1; echo "tests rc $?: $(tail -1 $OUT/tests.log)"
for r in $(seq 1 ${RUNS:-6}); do for v in default r6evrec; do
  if [ $v = default ]; then unset LFS_GSPLAT_LIB; else export LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$v.so; fi
  timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-ops-route 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('[$v]', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline'].get('launches_timed'), {k: v['avg_ms'] for k, v in list(d['kernels'].items())[:3]})"
done; done 2>&1 | tee $OUT/ab.txt
unset LFS_GSPLAT_LIB
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/trace_default -o trace -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ops-route > $REPO/$OUT/trace_default.log 2>&1
cd $REPO; python tools/step_timeline.py $OUT/trace_default | tail -14
