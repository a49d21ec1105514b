#!/bin/bash
# Round 5, lease 15: the two cases of seed 59 that met the integer quantisation of the radius bar replayed with the final fuzzer, then the whole seed again
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/lease15; mkdir -p $O
ok=1
for f in profiles/r05/lease13/states/fuzz_emulated_seed59_case2782.json profiles/r05/lease14/states/fuzz_emulated_seed59_case1855.json; do
  timeout 120 python tools/fuzz_emulated.py --gpu --oracle --flat 0.4 --replay $f > $O/replay_$(basename $f .json).txt 2>&1 || ok=0
  tail -n 3 $O/replay_$(basename $f .json).txt | cut -c1-400
done
if [ $ok = 1 ]; then
  timeout 300 python tools/fuzz_emulated.py --gpu --oracle --flat 0.4 --cases 3000 --seconds 280 --seed 59 --keep-going --state-dir $O/states > $O/fuzz_gpu_seed59_3000cases.txt 2>&1; echo "fuzz rc $?"
  grep -v "^  [a-z_ ]*:.*comparisons" $O/fuzz_gpu_seed59_3000cases.txt | cut -c1-800 | tail -8; grep "radii beyond ten\|degenerate UT" $O/fuzz_gpu_seed59_3000cases.txt
fi
