#!/bin/bash
# Round 6, lease 30: pinhole cameras rasterized with un-normalised directions (u, v, 1) (LFS_RAY_Z1, ray mode 2): rasterizer / step / headline tests on the new default,
# then the A/B against r6noz1 (the same source with -DLFS_RAY_Z1=0)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease30; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
timeout 1500 python -m pytest tests/test_gpu_000_canary.py tests/test_gpu_raster.py tests/test_gpu_aniso.py tests/test_gpu_refk_golden.py tests/test_gpu_gut_step.py tests/test_gpu_headline_parity.py \
  -q -m gpu -p no:cacheprovider -x > $OUT/tests.log 2>&1; echo "tests rc $?: $(tail -1 $OUT/tests.log)"
grep -n "^FAILED\|^ERROR" $OUT/tests.log | head -30
ab() {  # ab <rounds> <variants...>
  local rounds=$1; shift
  for r in $(seq 1 $rounds); do for v in "$@"; do
    if [ $v = default ]; then unset LFS_GSPLAT_LIB; else export LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$v.so; fi
    timeout 200 python bench.py --no-cpu-baseline --steps 300 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('[$v]', d['value'], d['ms_per_step'], d['roofline']['frac'], {k: v['avg_ms'] for k, v in d['kernels'].items()})"
  done; done; unset LFS_GSPLAT_LIB
}
ab ${AB_ROUNDS:-3} default ${AB_VARIANTS:-r6noz1} 2>&1 | tee $OUT/ab.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_default.json
python -c "
import json; d = json.load(open('$OUT/bench_default.json')); print('driver command:', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('ops_route', {}).get('ms_per_step'))"
