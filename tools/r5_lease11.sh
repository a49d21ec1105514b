#!/bin/bash
# Round 5, lease 11: the training-result criterion towards BASELINE's size.
#  (1) HIP side of the 100 000-Gaussian / 960x540 / SH 3 / flat-disk task, 7 000 iterations, deterministic (twice) + 3 float-atomic runs per seed; the oracle side
#      (tests/convergence_l1ssim.py --oracle, ~2.5 h per seed on the build container's 8 cores) is compared offline: profiles/r05/convergence_mse_100k_oracle.json
#  (2) trajectory segments at configs[1]'s size - 1 M Gaussians, 1920x1080, SH 3 - HIP (deterministic) against the oracle on the box's host cores: 24 steps from the
#      common start and 24 steps from a HIP-trained state (6 900 iterations)
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/lease11; mkdir -p $O
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" > $O/library.txt 2>&1
T100K="--loss mse --flat 30 --n 100000 --width 960 --height 540 --views 8 --sh-degree 3 --scale 0.025"
timeout 400 python tests/convergence_l1ssim.py --hip $T100K --seeds 0 1 2 3 --oracle-json profiles/r05/convergence_mse_100k_oracle.json > $O/psnr_100k_hip.txt 2> $O/psnr_100k_hip.err
T1M="--n 1000000 --width 1920 --height 1080 --views 8 --sh-degree 3 --scale 0.012 --flat 30 --steps 24 --checkpoints 1 2 4 8 16 24"
timeout 330 python tests/trajectory_check.py $T1M --out $O/trajectory_1M_from_start.json > $O/trajectory_1M_from_start.txt 2>&1
timeout 360 python tests/trajectory_check.py $T1M --pretrain 6900 --out $O/trajectory_1M_from_6900.json > $O/trajectory_1M_from_6900.txt 2>&1
tail -3 $O/psnr_100k_hip.txt | cut -c1-600; tail -2 $O/trajectory_1M_from_start.txt | cut -c1-400; tail -2 $O/trajectory_1M_from_6900.txt | cut -c1-400
