#!/bin/bash
# Round 6, lease 7b: the streaming kernels alone at 1 M and 4 M Gaussians (Infinity-Cache residency), and alternating as in the step
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease7; mkdir -p $OUT
timeout 200 python tools/bench_stream_kernels.py 2>&1 | tail -1 | tee $OUT/stream_kernels_alone_1M.json
timeout 200 python tools/bench_stream_kernels.py --n 4000000 --reps 20 2>&1 | tail -1 | tee $OUT/stream_kernels_alone_4M.json
