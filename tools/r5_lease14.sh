#!/bin/bash
# Round 5, lease 14: seed 59 of the GPU fuzz with the final fuzzer (a radius beyond ten image sizes is a degenerate unscented-transform row, held to 1 % instead of +-1 px)
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/lease14; mkdir -p $O
timeout 420 python tools/fuzz_emulated.py --gpu --oracle --flat 0.4 --cases 3000 --seconds 380 --seed 59 --keep-going --state-dir $O/states > $O/fuzz_gpu_seed59_3000cases.txt 2>&1; echo "fuzz rc $?"
grep -v "^  [a-z_ ]*:.*comparisons" $O/fuzz_gpu_seed59_3000cases.txt | cut -c1-800 | tail -12; grep "radii beyond ten\|degenerate UT" $O/fuzz_gpu_seed59_3000cases.txt
