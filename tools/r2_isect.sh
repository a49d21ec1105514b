#!/bin/bash
# two-pass scatter: parity tests + A/B + launch sequence.   gpurun --timeout 900 -- 'bash tools/r2_isect.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/isect; mkdir -p $OUT; cd $REPO
timeout 600 python -m pytest tests/test_gpu_intersect.py tests/test_gpu_fused.py tests/test_gpu_headline_parity.py tests/test_gpu_refk_golden.py -x -q 2>&1 | tail -5
for i in 1 2; do
timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline > $OUT/bench_two_pass_$i.json 2> $OUT/err.txt || tail -5 $OUT/err.txt
timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --debug-flags 32 > $OUT/bench_one_pass_$i.json 2> $OUT/err.txt || tail -5 $OUT/err.txt
done
python - <<'PY'
import json, os, glob
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "isect")
for f in sorted(glob.glob(out + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d["kernels"]
        print(os.path.basename(f), d["value"], d["ms_per_step"], {n: k[n]["avg_ms"] for n in ("isect_count_scan", "isect_scatter", "isect_tile_sort") if n in k})
    except Exception as e:
        print(f, "failed", e)
PY
bash tools/r2_seq.sh
