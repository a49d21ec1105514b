#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$REPO"; OUT=$REPO/gpurun_out/rows2; mkdir -p $OUT
(LFS_EXPERIMENTAL_ROWS=1 timeout 300 python -m pytest tests/test_gpu_raster_rows.py -q --tb=short 2>&1 | tail -30) > "$OUT/tests.txt"
tail -3 $OUT/tests.txt
for cfg in "default:" "rows:--row-kernels" "rows_merged:--row-kernels --row-lists merged"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  timeout 120 python bench.py --no-cpu-baseline $flags > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
done
export LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_rows_NO_ASM.so
for cfg in "noasm_rows:--row-kernels" "noasm_rows_merged:--row-kernels --row-lists merged"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  timeout 120 python bench.py --no-cpu-baseline $flags > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
done
python - <<'PY'
import json, os, glob
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "rows2")
for f in sorted(glob.glob(out + "/bench_*.json")):
    try:
        d = json.loads(open(f).read())
        print(os.path.basename(f), d["value"], d["ms_per_step"], d["config"]["n_isects"], {k: v["avg_ms"] for k, v in d["kernels"].items() if "raster" in k})
    except Exception as e:
        print(f, "failed:", e)
PY
