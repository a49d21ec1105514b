#!/bin/bash
# Which clock did the kernels run at?  GRBM_GUI_ACTIVE (cycles the GPU was busy, at the ACTUAL clock) per launch / the launch's duration from the same trace.
#   gpurun --timeout 600 -- 'bash tools/clock_pass.sh'   -> gpurun_out/clock_pass/summary.txt
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/clock_pass; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CYCLES SQ_CYCLES --output-format csv -d $OUT/pmc -o pmc -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile > $OUT/bench.log 2>&1
python - <<PY > $OUT/summary.txt 2>&1
import csv, glob, collections
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) if "End_Timestamp" in r else 0))
for k, d in sorted(cnt.items(), key=lambda kv: -sum(x[1] for x in kv[1].get("GRBM_GUI_ACTIVE", [(0, 0)]))):
    line = f"{k:62s}"
    for name, vals in sorted(d.items()):
        v = sum(x[0] for x in vals) / len(vals); t = sum(x[1] for x in vals) / len(vals)
        line += f" {name}={v:.4g}"
        if name in ("GRBM_GUI_ACTIVE", "GRBM_COUNT") and t > 0: line += f" ({v / t:.3f} cycles/ns over {t / 1000:.1f} us)"
    print(line)
PY
head -30 $OUT/summary.txt; tail -3 $OUT/bench.log
