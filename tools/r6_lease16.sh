#!/bin/bash
# Round 6, lease 16: (1) the intersect + step tests on the count kernel that scans in its last workgroup; (2) A/B with the order rotated every round: the tail kernel's
# register / prefetch forms, count + scan as one launch (default) or two (LFS_DEBUG_FLAGS=256); (3) how the driver's command (5 warm-up + 20 timed steps = 35 ms after
# start-up) compares with longer warm-ups and longer timed regions on the same box
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease16; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
timeout 900 python -m pytest tests/test_gpu_intersect.py tests/test_gpu_gut_step.py tests/test_gpu_headline_parity.py tests/test_gpu_fused.py -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc $?: $(tail -1 $OUT/tests.log)"
grep -n "^FAILED\|^ERROR" $OUT/tests.log | head -20
run() { # name, library suffix ("" = default), extra environment
  local name=$1 lib=$2
  local E="${3:-}"
  [ -n "$lib" ] && E="$E LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$lib.so"
  env $E timeout 300 python bench.py --no-cpu-baseline --no-ops-route --steps 200 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']; print('[$name]', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {n: k[n]['avg_ms'] for n in ('tail_sh_finish_adam', 'activations_projection_ut', 'isect_count_scan', 'raster_bwd', 'raster_fwd') if n in k})"
}
V=(default two_launch_count_scan tail_early tail_ke tail_d8 tail_ke_d8 tail_ke_nt tail_e_d8)
for r in 0 1 2 3; do
  for i in 0 1 2 3 4 5 6 7; do
    v=${V[$(( (i + 3 * r) % 8 ))]}
    case $v in default) run default "";; two_launch_count_scan) run $v "" LFS_DEBUG_FLAGS=256;; *) run $v $v;; esac
  done
done 2>&1 | tee $OUT/ab.txt
python - <<'PY' | tee $OUT/ab_summary.txt
import re, statistics, collections
rows = collections.defaultdict(list)
for line in open('gpurun_out/r6_lease16/ab.txt'):
    m = re.match(r'\[(\S+)\] (\S+) (\S+) (\S+) (\{.*\})', line)
    if m: rows[m.group(1)].append((float(m.group(2)), float(m.group(3)), float(m.group(4)), eval(m.group(5))))
for k, v in rows.items():
    print(f"{k:24s} img/s median {statistics.median(x[0] for x in v):8.2f}  min {min(x[0] for x in v):8.2f} max {max(x[0] for x in v):8.2f} | ms/step median {statistics.median(x[1] for x in v):.4f} | bwd live {statistics.median(x[2] for x in v):.4f} | tail {statistics.median(x[3].get('tail_sh_finish_adam', 0) for x in v):.4f} | count+scan {statistics.median(x[3].get('isect_count_scan', 0) for x in v):.4f}  n={len(v)}")
PY
# warm-up sensitivity of the driver's command
for spec in "5 20" "5 20" "30 20" "100 20" "300 20" "5 100" "5 300" "5 20"; do set -- $spec
  timeout 300 python bench.py --gpus 1 --steps $2 --warmup $1 --no-cpu-baseline --no-ops-route 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('warmup $1 steps $2:', d['value'], d['ms_per_step'], 'bwd live', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])"
done 2>&1 | tee $OUT/warmup_sensitivity.txt
