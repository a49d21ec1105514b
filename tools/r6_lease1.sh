#!/bin/bash
# Round 6, first GPU lease: the two microbenchmarks that decide this round's kernel work, and a baseline of the round-5 library on the same box.
#   gpurun --timeout 900 -- 'bash tools/r6_lease1.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease1; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 tools/pk_pair_rate.hip -o /tmp/pk_pair_rate 2>/dev/null && timeout 120 /tmp/pk_pair_rate | tee $OUT/pk_pair_rate.json
hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 tools/pk_rate.hip -o /tmp/pk_rate 2>/dev/null && timeout 60 /tmp/pk_rate | tee $OUT/pk_rate.json
hipcc -O3 --offload-arch=gfx950 tools/hbm_stream.hip -o /tmp/hbm_stream 2>/dev/null && timeout 300 /tmp/hbm_stream | tee $OUT/hbm_stream_ceiling.json | tail -3
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tee $OUT/bench_default.json | cut -c1-600
timeout 200 python bench.py --no-cpu-baseline --steps 300 --warmup 20 2>/dev/null | tee $OUT/bench_300.json | cut -c1-300
