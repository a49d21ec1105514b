#!/bin/bash
# tools/r4_dist8.sh: the whole `bench.py --gpus 8` flow with EIGHT gloo ranks sharing one GPU (smoke: collectives staged through the host, the ranks time-share the
# device - exercises exactly the control flow the driver's 8-GPU run takes: exchange selection at world x views = 8, the trial step, the side lines, the configs[3] line
# at 64 views per step; NOT a performance number).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/dist8; mkdir -p $OUT
cd $REPO
S=$(date +%s)
LFS_DIST_BACKEND=gloo timeout ${T:-600} python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29713 bench.py --gpus 8 --steps 3 --warmup 2 > $OUT/bench_gloo8_smoke.json 2> $OUT/bench_gloo8_smoke.err
echo "gloo x8 smoke rc $? in $(( $(date +%s) - S )) s" | tee $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import json
txt=open("$OUT/bench_gloo8_smoke.json").read()
try:
    a=json.loads(txt[txt.rindex('{"metric"'):])
    print("headline:", a["n_gpus"], a["config"]["parallelism"], a["ms_per_step"], "ms/step", a["scaling"])
    for k in ("replicated_other_exchange","sh_sharded","config4"):
        if k in a: print(k, ":", json.dumps(a[k])[:500])
    print("collectives:", json.dumps(a.get("collectives", {}).get("per_step"))[:400])
except Exception as e:
    print("parse failed", e); print(txt[-1500:])
PY
grep -v "^\[W\|amdgpu.ids\|Gloo\|^W0\|^\*\*\*\*\|OMP_NUM_THREADS" $OUT/bench_gloo8_smoke.err | tail -15
