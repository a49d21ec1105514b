#!/bin/bash
# Round 6, lease 8: the whole -m gpu suite on the current library (with the round-6 tests: staging slot, float-atomic PSNR, pinned radius counts), the bench line with
# ops_route, the BASELINE side lines configs[3] / configs[4] on the shipped library, and a kernel trace of the drop-in route (what libtorch's glue is made of)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease8; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
S=$(date +%s)
timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider -x --durations=15 > $OUT/suite.log 2>&1; echo "suite rc $? in $(( $(date +%s) - S )) s: $(tail -1 $OUT/suite.log)"
grep -n "FAILED\|Error\|mean gap\|float atomics, task\|staging slot" $OUT/suite.log | cut -c1-300 | head -30
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_default.json
timeout 400 python bench.py --workload syn-c --views-per-rank 8 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_config4_syn_c_8views.json
timeout 400 python bench.py --workload syn-d --strategy mcmc --bilateral-grid --loss l1_ssim --steps 20 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_config5_syn_d_mcmc_bilateral.json
python - <<'PY'
import json
for f in ("bench_default", "bench_config4_syn_c_8views", "bench_config5_syn_d_mcmc_bilateral"):
    try:
        d = json.loads(open(f"gpurun_out/r6_lease8/{f}.json").read())
        print(f, d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in list(d["kernels"].items())[:8]})
        if d.get("ops_route"): print("  ops_route", {k: d["ops_route"].get(k) for k in ("ms_per_step", "backend_kernel_ms", "libtorch_glue_ms")})
    except Exception as e:
        print(f, "failed", e)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/ops_trace -o t -- python $REPO/bench.py --path ops --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $REPO/$OUT/ops_trace.log 2>&1
cd $REPO; python - <<'PY'
import csv, glob, re
from collections import defaultdict
f = glob.glob("gpurun_out/r6_lease8/ops_trace/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:32]:
        print(f"{re.sub(r'[(].*', '', r['Name'])[:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
