#!/bin/bash
# Round 6, lease 36: LFS_REC_PKQ again, this time with every record arriving as ONE s_load_dwordx16 (LFS_REC_LOAD16: lease 28's build had half of them in five pieces), and
# LFS_REC_LOAD16 alone; rasterizer tests on both variants, A/B against the default
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease36; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
for V in ${AB_VARIANTS:-r6pkq r6load16}; do
LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$V.so timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_aniso.py tests/test_gpu_refk_golden.py tests/test_gpu_gut_step.py \
  -q -m gpu -p no:cacheprovider -x > $OUT/tests_$V.log 2>&1; echo "tests on $V rc $?: $(tail -1 $OUT/tests_$V.log)"
done
ab() {  # ab <rounds> <variants...>
  local rounds=$1; shift
  for r in $(seq 1 $rounds); do for v in "$@"; do
    if [ $v = default ]; then unset LFS_GSPLAT_LIB; else export LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$v.so; fi
    timeout 200 python bench.py --no-cpu-baseline --no-ops-route --steps 300 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('[$v]', d['value'], d['ms_per_step'], d['roofline']['frac'], {k: v['avg_ms'] for k, v in d['kernels'].items()})"
  done; done; unset LFS_GSPLAT_LIB
}
ab ${AB_ROUNDS:-4} default ${AB_VARIANTS:-r6pkq r6load16} 2>&1 | tee $OUT/ab.txt
