#!/bin/bash
# Round 5, sixth lease: the driver's round-end sequence verbatim on the final tree; L1 + D-SSIM PSNR (5 stored oracle seeds); MCMC + 3DGUT to 30 000 iterations on the shipped library
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r5_lease6; mkdir -p $OUT
bash tools/r3_driver_verbatim.sh r5a; cp -r gpurun_out/verbatim_r5a $OUT/
timeout 600 python tests/convergence_l1ssim.py --hip --loss l1_ssim --seeds 0 1 2 3 4 --atomic-runs 2 --det-runs 1 --oracle-json profiles/r02/convergence_l1ssim_oracle.json > $OUT/psnr_l1ssim.log 2>&1
tail -1 $OUT/psnr_l1ssim.log > $OUT/psnr_l1ssim.json; python -c "import json; print('[l1ssim]', json.dumps(json.load(open('$OUT/psnr_l1ssim.json'))['summary']))"
D=/tmp/syn_colmap
python tools/make_synthetic_colmap.py $D --views 96 --width 1296 --height 840 --gaussians 300000 --points 60000 > $OUT/make.log 2>&1 || { tail -20 $OUT/make.log; exit 1; }
timeout 600 python tools/train_colmap.py -d $D --gut --strategy mcmc -i 30000 --eval --eval-every 1000 -o /tmp/scale_out_30k > $OUT/train_mcmc_gut_30k.json 2> $OUT/train_mcmc_gut_30k.err
echo "rc $? $(tail -1 $OUT/train_mcmc_gut_30k.json | cut -c1-500)"; grep iteration $OUT/train_mcmc_gut_30k.err | python -c "
import sys, json
print(' '.join(f\"{json.loads(l)['iteration']}:{json.loads(l)['psnr']}\" for l in sys.stdin))"
