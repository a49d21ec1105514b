#!/bin/bash
# Round 6, lease 29: the training-result criterion at BASELINE configs[1]'s own size on the round-6 library - a 200-step trajectory segment (iterations 6800 -> 7000) of the
# 1 M-Gaussian / 1920x1080 / SH 3 / flat-disk task from a HIP-trained state, HIP (deterministic accumulation) against the oracle on the box's host cores; PSNR of both after it
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease29; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
T1M="--n 1000000 --width 1920 --height 1080 --views 8 --sh-degree 3 --scale 0.012 --flat 30 --steps ${STEPS:-200} --checkpoints 1 8 24 50 100 150 200"
timeout ${LIMIT:-1500} python tests/trajectory_check.py $T1M --pretrain 6800 --out $OUT/trajectory_1M_6800_to_7000.json > $OUT/trajectory_1M_6800_to_7000.txt 2>&1
echo "rc $?"; tail -2 $OUT/trajectory_1M_6800_to_7000.txt | cut -c1-500
