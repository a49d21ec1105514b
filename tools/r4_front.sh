#!/bin/bash
# tools/r4_front.sh <tag>: the step with the records packed by the projection kernel (round 4) against the round-3 order (LFS_DEBUG_FLAGS=64: separate pack kernel) on ONE box:
# the step-level parity tests, three alternating bench pairs, one kernel trace of each -> gpurun_out/front_<tag>/
set -u
TAG=${1:-a}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/front_$TAG; mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests/test_gpu_000_canary.py tests/test_gpu_gut_step.py tests/test_gpu_fused.py tests/test_gpu_pipeline.py tests/test_gpu_headline_parity.py tests/test_gpu_torch_ops.py tests/test_gpu_dp2.py -x -q -m gpu -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc $?: $(tail -1 $OUT/pytest.log)" | tee $OUT/summary.txt
[ "$(grep -c passed $OUT/pytest.log)" = "0" ] && tail -60 $OUT/pytest.log
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
for i in 1 2 3; do
  $B > $OUT/bench_packed_$i.json 2>> $OUT/bench.err
  LFS_DEBUG_FLAGS=64 $B > $OUT/bench_round3_order_$i.json 2>> $OUT/bench.err
  python - <<PY | tee -a $OUT/summary.txt
import json
a=json.load(open("$OUT/bench_packed_$i.json")); b=json.load(open("$OUT/bench_round3_order_$i.json"))
print("pair $i: packed by projection", a["ms_per_step"], "ms", a["value"], "img/s | round-3 order", b["ms_per_step"], "ms", b["value"], "img/s")
PY
done
cd /tmp && export TMPDIR=/tmp
P="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_packed -o trace -- $P > $OUT/trace_packed.log 2>&1
LFS_DEBUG_FLAGS=64 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_round3 -o trace -- $P > $OUT/trace_round3.log 2>&1
cd $REPO
for k in packed round3; do
  f=$(find $OUT/trace_$k -name "*kernel_stats.csv" | head -1)
  echo "== $k" >> $OUT/summary.txt; head -22 "$f" | cut -d, -f1-4 | cut -c1-120 >> $OUT/summary.txt
done
tail -50 $OUT/summary.txt
