#!/bin/bash
# Round 6, lease 35: the last SEVENTH of the training run at BASELINE configs[1]'s own size - iterations 6000 -> 7000 of the 1 M-Gaussian / 1920x1080 / SH 3 / flat-disk task from a
# HIP-trained state, HIP (deterministic accumulation) against the oracle on the box's host cores (about 3.3 s per oracle step: ~ 55 minutes); PSNR of both at iteration 7000
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease35; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
T1M="--n 1000000 --width 1920 --height 1080 --views 8 --sh-degree 3 --scale 0.012 --flat 30 --steps 1000 --checkpoints 1 8 50 100 200 400 600 800 1000"
timeout ${LIMIT:-4400} python tests/trajectory_check.py $T1M --pretrain 6000 --out $OUT/trajectory_1M_6000_to_7000.json > $OUT/trajectory_1M_6000_to_7000.txt 2>&1
echo "rc $?"; tail -2 $OUT/trajectory_1M_6000_to_7000.txt | cut -c1-500
