#!/bin/bash
# round-2 evidence: bench lines (SYN-B with the CPU baseline, configs 4 / 5, fastgs) + rocprofv3 trace and PMC passes.   gpurun --timeout 2400 -- 'bash tools/r2_final.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/final; mkdir -p $OUT; cd $REPO
timeout 400 python bench.py > $OUT/bench_synb_final.json 2> $OUT/bench_synb.err || tail -5 $OUT/bench_synb.err
timeout 300 python bench.py --rasterizer fastgs --no-cpu-baseline > $OUT/bench_synb_fastgs.json 2> $OUT/bench_fastgs.err || tail -5 $OUT/bench_fastgs.err
bash tools/r2_configs.sh > $OUT/configs.log 2>&1; tail -8 $OUT/configs.log
bash tools/profile.sh r02 > $OUT/profile.log 2>&1; tail -60 $OUT/profile.log
python - <<'PY'
import json, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "final")
for f in ("bench_synb_final.json", "bench_synb_fastgs.json"):
    try:
        d = json.loads(open(os.path.join(out, f)).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"], (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "failed", e)
PY
