#!/bin/bash
# Round 5, fourth lease: A/B of the three front-end / forward changes, the suite on the library that ships, the BASELINE-sized trajectory comparison, the stream structure of the
# data-parallel step under RCCL at world size 1
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r5_lease4; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
for v in fwdzero_off sort2launch shfwd_r4; do bash tools/ab_lib.sh $v 3 2>&1 | tee $OUT/ab_$v.txt; done
LFS_NOISE_LOG=$REPO/$OUT/noise.jsonl timeout 900 python -m pytest tests/ -q -m gpu -p no:cacheprovider > $OUT/suite.log 2>&1; echo "suite rc $?: $(tail -1 $OUT/suite.log)"; grep -n "FAILED\|mean gap\|Gaussians with a radius" $OUT/suite.log | cut -c1-200 | head -20
T="--n 100000 --width 960 --height 540 --views 8 --sh-degree 3 --scale 0.025 --flat 30 --checkpoints 1 2 5 10 50 100 --steps 100"
timeout 600 python tests/trajectory_check.py $T --out $OUT/trajectory_100k_from_start.json > $OUT/trajectory_100k_from_start.log 2>&1; echo "trajectory (start) rc $?"; tail -4 $OUT/trajectory_100k_from_start.log | cut -c1-400
timeout 900 python tests/trajectory_check.py $T --pretrain 6900 --out $OUT/trajectory_100k_from_6900.json > $OUT/trajectory_100k_from_6900.log 2>&1; echo "trajectory (6900) rc $?"; tail -5 $OUT/trajectory_100k_from_6900.log | cut -c1-400
export LFS_DIST_FORCE_COLLECTIVES=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29641
cd /tmp && export TMPDIR=/tmp
for lay in replicated factored; do
  rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/dp_trace_$lay -o trace -- python $REPO/bench.py --gpus 1 --steps 6 --warmup 3 --no-cpu-baseline --no-profile --$lay > $REPO/$OUT/dp_bench_$lay.log 2>&1
  python $REPO/tools/overlap_report.py $REPO/$OUT/dp_trace_$lay | tee $REPO/$OUT/dp_overlap_$lay.txt
done
