#!/bin/bash
# The first box with >= 2 GPUs runs THIS (round-5 review, "Next round" 7): no collective of this repository has ever moved a byte between two devices - the code
# paths exist, the gloo tests hold their logic, RCCL has only run at world size 1. Everything that can only be learned on such a box, in one go, so that nothing is
# discovered late:
#   1. the RCCL variants of tests/test_gpu_dp2.py (they self-skip below 2 GPUs) and tests/test_gpu_rccl_world1.py
#   2. bench.py at --gpus 1 / 2 / 4 / 8 (as many as the box has), the driver's own launch line, one rank per GPU over RCCL
#   3. a SCALE-shaped JSON: per-N value, ms/step, the layout that was the headline, the side lines (other exchange, SH-sharded, BASELINE configs[3]) and the
#      collectives' device time per step next to DESIGN.md 7's estimates
#   bash tools/first_multigpu.sh [outdir]        (from the repo root; ~6 minutes on 8 GPUs)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=${1:-gpurun_out/first_multigpu}; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0     # the host driver supports dmabuf IPC only: without it RCCL fails with hipIpcGetMemHandle: invalid argument
NG=$(python -c "import torch; print(torch.cuda.device_count())")
echo "GPUs visible: $NG" | tee $OUT/summary.txt
if [ "$NG" -lt 2 ]; then echo "fewer than 2 GPUs: nothing to learn here (the gloo / world-1 tests cover this box)" | tee -a $OUT/summary.txt; exit 0; fi
timeout 900 python -m pytest tests/test_gpu_dp2.py tests/test_gpu_rccl_world1.py -q -m gpu -p no:cacheprovider -rs > $OUT/rccl_tests.log 2>&1
echo "RCCL tests rc $?: $(tail -1 $OUT/rccl_tests.log)" | tee -a $OUT/summary.txt
PORT=29731
for N in 1 2 4 8; do
  [ $N -le $NG ] || continue
  if [ $N = 1 ]; then
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n$N.json 2> $OUT/bench_n$N.err
  else
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $N --steps 20 --warmup 5 > $OUT/bench_n$N.json 2> $OUT/bench_n$N.err
  fi
  echo "bench --gpus $N rc $?" | tee -a $OUT/summary.txt
  PORT=$((PORT + 1))
done
python - "$OUT" <<'PY' | tee -a $OUT/summary.txt
import json, os, sys
out = sys.argv[1]
rows, base = [], None
for n in (1, 2, 4, 8):
    p = os.path.join(out, f"bench_n{n}.json")
    if not os.path.exists(p):
        continue
    txt = open(p).read()
    try:
        a = json.loads(txt[txt.rindex('{"metric"'):])
    except Exception as e:
        rows.append({"n_gpus": n, "failed": f"no bench line ({e})"}); continue
    if n == 1:
        base = a["value"]
    row = {"n_gpus": n, "value": a["value"], "ms_per_step": a["ms_per_step"], "parallelism": a["config"]["parallelism"], "scaling": a["scaling"],
           "speedup_vs_1": round(a["value"] / base, 3) if base else None, "collectives_per_step": a.get("collectives", {}).get("per_step"),
           "other_exchange": a.get("replicated_other_exchange"), "sh_sharded": a.get("sh_sharded"), "config4": a.get("config4")}
    rows.append(row)
scale = {"what": "first multi-GPU run of this repository (tools/first_multigpu.sh)", "rows": rows,
         "design_estimates_at_8": {"factored_1_view_per_rank": 5.9, "flat_1_view_per_rank": 4.2, "configs3_8_views_per_rank": 7.0, "source": "DESIGN.md 7"}}
json.dump(scale, open(os.path.join(out, "SCALE_first_multigpu.json"), "w"), indent=1)
for r in rows:
    print({k: r[k] for k in r if k in ("n_gpus", "value", "ms_per_step", "parallelism", "speedup_vs_1", "failed")})
PY
