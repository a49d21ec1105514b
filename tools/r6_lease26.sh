#!/bin/bash
# Round 6, lease 26: the differential fuzzer against the GPU library with the rotated records / symmetric accumulators (every stage of the operator chain against the oracle,
# 40 % flat-disk cases, all camera models and shutters): one seed that was clean on the round-5 library (31), 3000 cases
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease26; mkdir -p $OUT/states
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
timeout 400 python tools/fuzz_emulated.py --gpu --oracle --flat 0.4 --cases 3000 --seconds 330 --seed 31 --keep-going --state-dir $OUT/states > $OUT/fuzz_gpu_seed31_3000cases.txt 2>&1; echo "fuzz rc $?"
tail -30 $OUT/fuzz_gpu_seed31_3000cases.txt | cut -c1-300
