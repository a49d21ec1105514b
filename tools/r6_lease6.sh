#!/bin/bash
# Round 6, lease 6: raster_finish_adam_kernel with every load in one round trip (now the default) against the round-5 form (finish4trips), and sh_fwd with
# unconditional coefficient loads (shuncond): the step tests on the new default first, then the alternating A/B
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease6; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_000_canary.py tests/test_gpu_gut_step.py tests/test_gpu_fused.py tests/test_gpu_pipeline.py -q -m gpu -p no:cacheprovider -x > $OUT/tests_default.log 2>&1
echo "default tests rc $?: $(tail -1 $OUT/tests_default.log)"
LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_shuncond.so timeout 600 python -m pytest tests/test_gpu_projection_sh.py tests/test_gpu_gut_step.py tests/test_gpu_fused.py -q -m gpu -p no:cacheprovider -x > $OUT/tests_shuncond.log 2>&1
echo "shuncond tests rc $?: $(tail -1 $OUT/tests_shuncond.log)"
bash tools/ab_multi.sh 3 finish4trips shuncond 2>&1 | tee $OUT/ab.txt
