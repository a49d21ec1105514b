#!/bin/bash
# tools/r4_dist.sh <tag>: the data-parallel layouts on ONE GPU - (1) the two/three-rank gloo tests with the real kernels (tests/test_gpu_dp2.py), (2) what ONE rank of an
# N-GPU job executes, with every collective issued through RCCL at world size 1 (LFS_DIST_FORCE_COLLECTIVES=1): replicated flat all-reduce, factored SH exchange,
# SH-sharded. No collective runs between two GPUs here - these are code-path executions and per-rank compute timings, not scaling numbers.
set -u
TAG=${1:-a}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/dist_$TAG; mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_dp2.py tests/test_gpu_rccl_world1.py -x -q -m gpu -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc $?: $(tail -1 $OUT/pytest.log)" | tee $OUT/summary.txt
grep -q passed $OUT/pytest.log || tail -60 $OUT/pytest.log
export LFS_DIST_FORCE_COLLECTIVES=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29631
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
for lay in replicated factored sh-sharded; do
  timeout 600 $B --$lay > $OUT/bench_world1_forced_$lay.json 2>> $OUT/bench.err
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    a=json.load(open("$OUT/bench_world1_forced_$lay.json"))
    print("$lay:", a["config"]["parallelism"], a["ms_per_step"], "ms/step", a["value"], "img/s | collectives per step:", json.dumps(a["collectives"]["per_step"]))
except Exception as e:
    print("$lay: failed", e)
PY
done
tail -5 $OUT/bench.err
