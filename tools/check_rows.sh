#!/bin/bash
# First GPU call for the experimental quadrant-row rasterizer kernels (csrc/lfs_raster_rows.cuh; logic verified on the CPU emulator only):
#   gpurun --timeout 600 -- 'bash tools/check_rows.sh'
# 1. their parity tests against the default kernels (forward bit-identical), 2. A/B bench lines, 3. kernel stats of the row path.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/rows
mkdir -p "$OUT"
cd "$REPO"
(LFS_EXPERIMENTAL_ROWS=1 timeout 300 python -m pytest tests/test_gpu_raster_rows.py -q --tb=short 2>&1 | tail -40) > "$OUT/tests.txt"
timeout 120 python bench.py --no-cpu-baseline > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
timeout 120 python bench.py --no-cpu-baseline --row-kernels > "$OUT/bench_rows.json" 2> "$OUT/bench_rows.err"
timeout 120 python bench.py --no-cpu-baseline --row-kernels --row-lists merged > "$OUT/bench_rows_merged.json" 2> "$OUT/bench_rows_merged.err"
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --row-kernels > "$OUT/bench_trace.log" 2>&1
cd "$REPO"
tail -5 "$OUT/tests.txt"
python - <<'PY'
import json, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "rows")
for name in ("bench_default", "bench_rows", "bench_rows_merged"):
    try:
        d = json.loads(open(os.path.join(out, name + ".json")).read())
        print(name, d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items() if "raster" in k})
    except Exception as e:
        print(name, "failed:", e)
PY
