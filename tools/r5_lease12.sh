#!/bin/bash
# Round 5, lease 12: closing evidence on the library that ships.
#  (1) the differential fuzzer against the GPU library, 3 000 fresh cases (40 % flat disks), every stage against the oracle
#  (2) two more trajectory segments at configs[1]'s size (1 M Gaussians, 1920x1080, SH 3): from iteration 1000 (shN enters Adam, the step becomes one C++ call) and
#      from iteration 3000 (the bench's steady state), 24 steps each, HIP (deterministic) against the oracle
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/lease12; mkdir -p $O
timeout 400 python tools/fuzz_emulated.py --gpu --oracle --flat 0.4 --cases 3000 --seconds 360 --seed 59 > $O/fuzz_gpu_seed59_3000cases.txt 2>&1; echo "fuzz rc $?"
T1M="--n 1000000 --width 1920 --height 1080 --views 8 --sh-degree 3 --scale 0.012 --flat 30 --steps 24 --checkpoints 1 2 4 8 16 24"
timeout 330 python tests/trajectory_check.py $T1M --pretrain 1000 --out $O/trajectory_1M_from_1000.json > $O/trajectory_1M_from_1000.txt 2>&1
timeout 330 python tests/trajectory_check.py $T1M --pretrain 3000 --out $O/trajectory_1M_from_3000.json > $O/trajectory_1M_from_3000.txt 2>&1
head -4 $O/fuzz_gpu_seed59_3000cases.txt | cut -c1-300; tail -1 $O/trajectory_1M_from_1000.txt | cut -c1-300; tail -1 $O/trajectory_1M_from_3000.txt | cut -c1-300
