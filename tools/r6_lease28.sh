#!/bin/bash
# Round 6, lease 28: q.x / q.y of the evaluation as one packed chain on interleaved record columns (LFS_REC_PKQ, variant r6pkq) against the default: bit identity of
# deterministic training steps, rasterizer tests on the variant, A/B
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease28; mkdir -p $OUT
V=${VARIANT:-r6pkq}
LFS_DEBUG_FLAGS=16 timeout 300 python tools/step_bits.py 2>&1 | tail -3 | tee $OUT/bits_default.txt
LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$V.so LFS_DEBUG_FLAGS=16 timeout 300 python tools/step_bits.py 2>&1 | tail -3 | tee $OUT/bits_$V.txt
LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$V.so timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_aniso.py tests/test_gpu_refk_golden.py tests/test_gpu_gut_step.py \
  -q -m gpu -p no:cacheprovider -x > $OUT/tests_$V.log 2>&1; echo "tests on $V rc $?: $(tail -1 $OUT/tests_$V.log)"
ab() {  # ab <rounds> <variants...>
  local rounds=$1; shift
  for r in $(seq 1 $rounds); do for v in "$@"; do
    if [ $v = default ]; then unset LFS_GSPLAT_LIB; else export LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$v.so; fi
    timeout 200 python bench.py --no-cpu-baseline --steps 300 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('[$v]', d['value'], d['ms_per_step'], d['roofline']['frac'], {k: v['avg_ms'] for k, v in d['kernels'].items()})"
  done; done; unset LFS_GSPLAT_LIB
}
ab ${AB_ROUNDS:-3} default $V 2>&1 | tee $OUT/ab.txt
