#!/bin/bash
# First lease of the next round: decide whether -DLFS_BWD_REORTH=1 (K8's gradients for flat Gaussians) becomes the default.
#   gpurun --timeout 1500 -- 'bash tools/r5_reorth_ab.sh'
# 1. the new edge-size tests of tests/test_gpu_zz_edge_sizes.py on hardware (they have only run on the emulated library so far)
# 2. same-box A/B of the bench, default build against the variant (tools/ab_lib.sh): the cost of the seven extra instructions per evaluation of raster_bwd
# 3. the GPU parity suite on the VARIANT library (LFS_GSPLAT_LIB), the PSNR test included: its deterministic trajectories are re-drawn by the change
# 4. the PSNR comparison on all stored seeds, both builds, for the mean gap with its interval
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r5_reorth; mkdir -p $OUT
python tools/build_variant.py reorth raster.hip -DLFS_BWD_REORTH=1 > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 300 python -m pytest tests/test_gpu_zz_edge_sizes.py -q -m gpu -p no:cacheprovider > $OUT/edge_sizes.log 2>&1; echo "edge sizes rc $?: $(tail -1 $OUT/edge_sizes.log)"
bash tools/ab_lib.sh reorth 3 2>&1 | tee $OUT/ab.txt
LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_reorth.so timeout 900 python -m pytest tests/ -q -m gpu -p no:cacheprovider > $OUT/suite_variant.log 2>&1
echo "suite on the variant rc $?: $(tail -1 $OUT/suite_variant.log)"; grep -n "FAILED\|mean gap" $OUT/suite_variant.log | head -20
for v in default reorth; do
  if [ $v = default ]; then unset LFS_GSPLAT_LIB; else export LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_reorth.so; fi
  LFS_PSNR_SEEDS=26 timeout 600 python -m pytest "tests/test_gpu_convergence.py::test_psnr_after_7k_iterations_mean_gap_to_the_oracle_within_0p05_db" -q -s -m gpu -p no:cacheprovider 2>&1 | grep "seed \|PSNR after\|passed\|failed" > $OUT/psnr_$v.txt
  echo "[$v] $(grep 'PSNR after' $OUT/psnr_$v.txt)"
done
