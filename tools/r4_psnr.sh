#!/bin/bash
# tools/r4_psnr.sh <tag>: PSNR after 7000 iterations, HIP (deterministic + float-atomic runs) against the stored oracle models, task seeds 0 .. 14 (SEEDS=...), for the BENCHMARKED
# step (clamped MSE through the C++ step driver) and for the reference's loss (L1 + 0.2 D-SSIM) -> gpurun_out/psnr_<tag>/
set -u
TAG=${1:-a}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/psnr_$TAG; mkdir -p $OUT
cd $REPO
timeout ${PSNR_TIMEOUT:-900} python tests/convergence_l1ssim.py --hip --loss mse --seeds ${SEEDS:-0 1 2 3 4 5 6 7 8 9 10 11 12 13 14} --atomic-runs ${ATOMIC_RUNS:-2} --det-runs ${DET_RUNS:-2} --oracle-json profiles/r04/convergence_mse_oracle.json > $OUT/convergence_mse_hip.log 2>&1
tail -1 $OUT/convergence_mse_hip.log > $OUT/convergence_mse_hip.json
python -c "
import json; a=json.load(open('$OUT/convergence_mse_hip.json')); print('MSE', json.dumps(a['summary']))
for k,v in a['seeds'].items(): print(k, v)"
