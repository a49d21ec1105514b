#!/bin/bash
# Round 5, second lease: (1) the whole -m gpu suite on the default library, (2) the differential fuzzer against the GPU library, (3) same-box A/B of the sh_fwd rewrite
# (split prefetch + SGPR row bases: 100 -> 64 VGPRs) against the round-4 form of sh.hip, (4) PSNR after 7000 iterations on all 26 stored seeds for the default (reorth) and the
# noreorth library, (5) the same on the flat-disk task where oracle trajectories exist, (6) MCMC + 3DGUT at scale on both libraries
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r5_lease2; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
LFS_NOISE_LOG=$REPO/$OUT/noise.jsonl timeout 900 python -m pytest tests/ -q -m gpu -p no:cacheprovider > $OUT/suite.log 2>&1; echo "suite rc $?: $(tail -1 $OUT/suite.log)"; grep -n "FAILED\|mean gap\|Gaussians with a radius" $OUT/suite.log | cut -c1-200 | head -20
timeout 420 python tools/fuzz_emulated.py --gpu --oracle --flat 0.4 --cases 2000 --seconds 380 --seed 31 > $OUT/fuzz_gpu.txt 2>&1; echo "fuzz rc $?"; tail -36 $OUT/fuzz_gpu.txt | cut -c1-200
bash tools/ab_lib.sh shfwd_r4 3 2>&1 | tee $OUT/ab_shfwd.txt
for v in default noreorth; do
  if [ $v = default ]; then unset LFS_GSPLAT_LIB; else export LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$v.so; fi
  timeout 900 python tests/convergence_l1ssim.py --hip --loss mse --seeds $(seq 0 25) --atomic-runs 2 --det-runs 1 --oracle-json profiles/r04/convergence_mse_oracle.json > $OUT/psnr_$v.log 2>&1
  tail -1 $OUT/psnr_$v.log > $OUT/psnr_$v.json; python -c "import json; print('[$v]', json.dumps(json.load(open('$OUT/psnr_$v.json'))['summary']))"
  if [ -f profiles/r05/convergence_mse_flat50_oracle.json ]; then
    S=$(python -c "import json; print(' '.join(sorted(json.load(open('profiles/r05/convergence_mse_flat50_oracle.json'))['seeds'], key=int)))")
    timeout 600 python tests/convergence_l1ssim.py --hip --loss mse --flat 50 --seeds $S --atomic-runs 2 --det-runs 1 --oracle-json profiles/r05/convergence_mse_flat50_oracle.json > $OUT/psnr_flat_$v.log 2>&1
    tail -1 $OUT/psnr_flat_$v.log > $OUT/psnr_flat_$v.json; python -c "import json; print('[$v flat]', json.dumps(json.load(open('$OUT/psnr_flat_$v.json'))['summary']))"
  fi
done
unset LFS_GSPLAT_LIB
D=/tmp/syn_colmap
python tools/make_synthetic_colmap.py $D --views 96 --width 1296 --height 840 --gaussians 300000 --points 60000 > $OUT/make.log 2>&1 || { tail -20 $OUT/make.log; exit 1; }
for v in default noreorth; do
  if [ $v = default ]; then unset LFS_GSPLAT_LIB; else export LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$v.so; fi
  timeout 420 python tools/train_colmap.py -d $D --gut --strategy mcmc -i 12500 --eval --eval-every 500 -o /tmp/scale_out_$v > $OUT/train_mcmc_gut_$v.json 2> $OUT/train_mcmc_gut_$v.err
  echo "[$v] rc $? $(tail -1 $OUT/train_mcmc_gut_$v.json | cut -c1-600)"; tail -2 $OUT/train_mcmc_gut_$v.err | cut -c1-300
done
