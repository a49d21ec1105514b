#!/bin/bash
# Round 6, lease 27: M0 written once per kernel (LFS_RED_M0_ONCE) and the totals leaving through a buffer atomic whose dead lanes are out of range (LFS_RED_BUF_ATOMIC):
# rasterizer / step / headline tests on the new default, then the A/B: default | r6m0only (buffer atomic off) | r6m0save (the library of commit bc0542f)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease27; mkdir -p $OUT
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
ab ${AB_ROUNDS:-3} default ${AB_VARIANTS:-r6m0only r6m0save} 2>&1 | tee $OUT/ab.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_default.json
python -c "
import json; d = json.load(open('$OUT/bench_default.json')); print('driver command:', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('ops_route', {}).get('ms_per_step'))"
