#!/bin/bash
# PSNR after 7k iterations, HIP vs the oracle's stored final models (5 seeds).   gpurun --timeout 1500 -- 'bash tools/r2_conv.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/conv; mkdir -p $OUT; cd $REPO
timeout 1200 python tests/convergence_l1ssim.py --hip --seeds 0 1 2 3 4 > $OUT/convergence_l1ssim_hip.log 2> $OUT/err.txt || tail -20 $OUT/err.txt
tail -1 $OUT/convergence_l1ssim_hip.log > $OUT/convergence_l1ssim_hip.json
cat $OUT/convergence_l1ssim_hip.log | cut -c1-600
