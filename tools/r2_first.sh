#!/bin/bash
# Round-2 first GPU call: quadrant-row kernels (tools/check_rows.sh) + the DPP variant of the SH group sums.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
bash tools/check_rows.sh
OUT=$REPO/gpurun_out/shdpp
mkdir -p "$OUT"
export LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_shdpp.so
(timeout 300 python -m pytest tests/test_gpu_projection_sh.py tests/test_gpu_fused.py -q --tb=short 2>&1 | tail -15) > "$OUT/tests.txt"
timeout 120 python bench.py --no-cpu-baseline > "$OUT/bench_shdpp.json" 2> "$OUT/bench_shdpp.err"
tail -3 "$OUT/tests.txt"
python - <<'PY'
import json, os
d = json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/shdpp/bench_shdpp.json")).read())
print("shdpp", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
PY
