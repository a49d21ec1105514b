#!/bin/bash
# tools/r4_dist2.sh <tag>: (1) the whole N > 1 flow of bench.py with TWO gloo ranks sharing one GPU (smoke: collectives staged through the host - exercises the headline
# trial step, the side lines and the config-4 line, not a performance number); (2) one rank's cost of 8 views per step under the two exchanges of the replicated
# layout, RCCL at world size 1 (LFS_DIST_FORCE_COLLECTIVES=1): the multi-view SH backward of the factored exchange over 8 views against 8 per-view SH backwards.
set -u
TAG=${1:-a}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/dist2_$TAG; mkdir -p $OUT
cd $REPO
LFS_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --steps 4 --warmup 3 > $OUT/bench_gloo2_smoke.json 2> $OUT/bench_gloo2_smoke.err
echo "gloo x2 smoke rc $?" | tee $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import json
txt=open("$OUT/bench_gloo2_smoke.json").read()
try:
    a=json.loads(txt[txt.rindex('{"metric"'):])
    print("headline:", a["config"]["parallelism"], a["ms_per_step"], "ms/step")
    for k in ("replicated_other_exchange","sh_sharded","config4"):
        if k in a: print(k, ":", json.dumps(a[k])[:400])
except Exception as e:
    print("parse failed", e); print(txt[-1500:])
PY
grep -v "^\[W\|amdgpu.ids\|Gloo" $OUT/bench_gloo2_smoke.err | tail -15
export LFS_DIST_FORCE_COLLECTIVES=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29641
for lay in replicated factored; do
  timeout 600 python bench.py --gpus 1 --steps 10 --warmup 4 --no-cpu-baseline --views-per-rank 8 --$lay > $OUT/bench_world1_forced_8views_$lay.json 2>> $OUT/bench.err
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    txt=open("$OUT/bench_world1_forced_8views_$lay.json").read()
    a=json.loads(txt[txt.rindex('{"metric"'):])
    k=a.get("kernels",{})
    print("8 views/step, $lay:", a["config"]["parallelism"], a["ms_per_step"], "ms/step", a["value"], "img/s | sh_bwd", {n:v for n,v in k.items() if n.startswith("sh_")}, "| collectives:", json.dumps(a["collectives"]["per_step"]))
except Exception as e:
    print("$lay: failed", e)
PY
done
