#!/bin/bash
# Round 6, lease 25: is the -0.067 dB mean gap of the 26-seed isotropic task (deterministic mode, lease 24) a bias of the new backward or the draw of a chaotic task?
# Float-atomic runs re-draw every trajectory: 26 seeds x 3 runs on the new default library and on r6base (the round-6 form before the rotated records), paired by seed.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease25; mkdir -p $OUT
for v in default r6base; do
  if [ $v = default ]; then unset LFS_GSPLAT_LIB; else export LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$v.so; fi
  timeout 1500 python tests/convergence_l1ssim.py --hip --loss mse --seeds $(seq 0 25) --atomic-runs 3 --det-runs 1 --oracle-json profiles/r04/convergence_mse_oracle.json > $OUT/psnr_isotropic_$v.log 2>&1
  tail -1 $OUT/psnr_isotropic_$v.log > $OUT/psnr_isotropic_$v.json
  python -c "
import json; a=json.load(open('$OUT/psnr_isotropic_$v.json')); print('$v', json.dumps(a['summary']))"
done
