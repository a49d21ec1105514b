#!/bin/bash
# Round 6, lease 4: lane-mask selects (LFS_SEL_E64) in raster_fwd / raster_bwd: parity on the variant, then same-box A/B
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease4; mkdir -p $OUT
LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_sel64.so timeout 600 python -m pytest tests/test_gpu_raster.py tests/test_gpu_refk_golden.py tests/test_gpu_gut_step.py -q -m gpu -p no:cacheprovider -x > $OUT/tests_sel64.log 2>&1
echo "sel64 tests rc $?: $(tail -1 $OUT/tests_sel64.log)"
bash tools/ab_lib.sh sel64 3 2>&1 | tee $OUT/ab_sel64.txt
