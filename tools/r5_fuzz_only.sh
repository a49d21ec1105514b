#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r5_fuzz; mkdir -p $OUT
timeout ${FUZZ_S:-600} python tools/fuzz_emulated.py --gpu --oracle --flat 0.4 --cases ${FUZZ_CASES:-2500} --seconds ${FUZZ_SEC:-560} --seed ${FUZZ_SEED:-31} > $OUT/fuzz_gpu.txt 2>&1; echo "fuzz rc $?"; tail -48 $OUT/fuzz_gpu.txt | cut -c1-400
