#!/bin/bash
# Round 6, lease 18: (1) raster / step / parity tests on the build whose forward marks the entries nothing composited (LFS_FWD_MARK=1, marks stored through a second
# pointer so that the walker's entry loads stay scalar); (2) same-box A/B, order rotated: default (marks) vs nomark vs a fast-math projection kernel (timing only);
# (3) what the fast-math projection changes in its outputs (tools/proj_fast_probe.py)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease18; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_gut_step.py tests/test_gpu_headline_parity.py tests/test_gpu_fused.py tests/test_gpu_aniso.py tests/test_gpu_intersect.py -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc $?: $(tail -1 $OUT/tests.log)"
grep -n "^FAILED\|^ERROR" $OUT/tests.log | head -20
run() { # name, library suffix ("" = default)
  local name=$1 lib=$2 E=""
  [ -n "$lib" ] && E="LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$lib.so"
  env $E timeout 300 python bench.py --no-cpu-baseline --no-ops-route --steps 200 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']; print('[$name]', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {n: k[n]['avg_ms'] for n in ('raster_bwd', 'raster_fwd', 'activations_projection_ut', 'raster_cull', 'tail_sh_finish_adam') if n in k})"
}
V=(default nomark projfast)
for r in 0 1 2 3; do
  for i in 0 1 2; do
    v=${V[$(( (i + r) % 3 ))]}
    case $v in default) run default "";; *) run $v $v;; esac
  done
done 2>&1 | tee $OUT/ab.txt
python tools/proj_fast_probe.py --save /tmp/proj_default.npz 2>/dev/null | tail -1 | tee $OUT/proj_probe.txt
LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_projfast.so python tools/proj_fast_probe.py --compare /tmp/proj_default.npz 2>/dev/null | tail -1 | tee -a $OUT/proj_probe.txt
LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_projcontract.so python tools/proj_fast_probe.py --compare /tmp/proj_default.npz 2>/dev/null | tail -1 | tee -a $OUT/proj_probe.txt
