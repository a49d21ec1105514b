#!/bin/bash
# Round 5, lease 13: seed 59 of the GPU fuzz again, with --keep-going (case 2782 failed its radii bar in lease 12: the state file had gone to the box's /tmp) - every failed
# case's state under gpurun_out/, the offending rows printed
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/lease13; mkdir -p $O
timeout 420 python tools/fuzz_emulated.py --gpu --oracle --flat 0.4 --cases 3000 --seconds 380 --seed 59 --keep-going --state-dir $O/states > $O/fuzz_gpu_seed59_keep_going.txt 2>&1; echo "fuzz rc $?"
grep -v "^  [a-z_ ]*:.*comparisons" $O/fuzz_gpu_seed59_keep_going.txt | cut -c1-1500 | tail -30
