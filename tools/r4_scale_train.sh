#!/bin/bash
# tools/r4_scale_train.sh: BASELINE configs[2] stand-in (no dataset in the image): a COLMAP directory rendered from a known 300 k-Gaussian scene (96 views, 1296x840, the
# size of Mip-NeRF360 'garden' at images_4), trained for 30 000 iterations with L1 + D-SSIM, every 8th view held out -> PSNR / SSIM. Two runs: ADC densification with the
# reference's default (EWA) rasterizer - the reference's ADC needs its densification_info, which only that rasterizer's backward produces - and MCMC with the 3DGUT one.
set -u
OUT=gpurun_out/scale_train; mkdir -p $OUT
D=/tmp/syn_colmap
python tools/make_synthetic_colmap.py $D --views 96 --width 1296 --height 840 --gaussians 300000 --points 60000 > $OUT/make.log 2>&1 || { tail -20 $OUT/make.log; exit 1; }
for run in ${RUNS:-default: mcmc:--gut}; do
  strat=${run%%:*}; flag=${run#*:}
  timeout ${TRAIN_TIMEOUT:-420} python tools/train_colmap.py -d $D $flag --strategy $strat -i ${ITERS:-30000} --eval -o /tmp/scale_out_$strat > $OUT/train_$strat.json 2> $OUT/train_$strat.err
  echo "rc $? $(tail -1 $OUT/train_$strat.json)"; tail -2 $OUT/train_$strat.err
done
