#!/bin/bash
# SQ counters of the default rasterizer kernels and of the quadrant-row kernels (why the row kernels lose), one pass each.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/rows_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile"
for v in default rows; do
  F=""; [ $v = rows ] && F="--row-kernels"
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $OUT/sq_$v -o pmc -- $BENCH $F > $OUT/sq_$v.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS --output-format csv -d $OUT/sq2_$v -o pmc -- $BENCH $F > $OUT/sq2_$v.log 2>&1
done
python - <<'PY'
import csv, glob, os, re
from collections import defaultdict
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "rows_pmc")
for d in sorted(glob.glob(out + "/sq*_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("lfs::", "")[:40]
            if "raster" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
        print("==", os.path.basename(d))
        for k in sorted(acc):
            n = len(cnt[k])
            print(f"{k:42s} n={n} " + " ".join(f"{c}={v/n:.4g}" for c, v in sorted(acc[k].items())))
PY
