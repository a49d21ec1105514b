#!/bin/bash
# Round 5, eighth lease: the MCMC + 3DGUT decline - generalisation gap or unstable optimisation? held-out PSNR, PSNR on 12 training views and the share of Gaussians whose thin
# axis projects below half a pixel, every 2 500 iterations of a 30 000-iteration run; the same with the EWA rasterizer beside it
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r5_lease8; mkdir -p $OUT
D=/tmp/syn_colmap
python tools/make_synthetic_colmap.py $D --views 96 --width 1296 --height 840 --gaussians 300000 --points 60000 > $OUT/make.log 2>&1 || { tail -20 $OUT/make.log; exit 1; }
for r in gut fastgs; do
  flag=""; [ $r = gut ] && flag="--gut"
  timeout 900 python tools/train_colmap.py -d $D $flag --strategy mcmc -i 30000 --eval --eval-every 2500 -o /tmp/scale_out_y > $OUT/mcmc_${r}_30k_curves.json 2> $OUT/mcmc_${r}_30k_curves.err
  echo "[$r] rc $?"; grep iteration $OUT/mcmc_${r}_30k_curves.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['iteration'], 'held-out', d['psnr'], 'train views', d['psnr_train_views'], 'thin axis < 0.5 px', d['thin_axis_below_half_pixel'], 'flat >= 10', d['frac_aspect_ge_10'], 'N', d['gaussians'])"
done
