#!/bin/bash
# Round 6, lease 15: the fused-tail step on the rebuilt libraries - the whole GPU suite (no -x), alternating A/B of the tail kernel's register / prefetch forms
# (LFS_TAIL_KEEP / LFS_TAIL_EARLY / LFS_TAIL_DEPTH / LFS_TAIL_NT variant libraries of tools/build_variant.py), the driver's command, and the round's clean
# rocprofv3 evidence (tools/profile.sh: trace + 4 PMC passes on the library that ships)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease15; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc $?: $(tail -1 $OUT/tests.log)"
grep -n "^FAILED\|^ERROR" $OUT/tests.log | head -20
run() { # name, library suffix ("" = default), extra environment
  local name=$1 lib=$2
  local E="${3:-}"
  [ -n "$lib" ] && E="$E LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$lib.so"
  env $E timeout 300 python bench.py --no-cpu-baseline --no-ops-route --steps 300 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']; print('[$name]', d['value'], d['ms_per_step'], {n: k[n]['avg_ms'] for n in ('tail_sh_finish_adam', 'activations_projection_ut', 'raster_bwd', 'raster_fwd') if n in k})"
}
for r in 1 2; do
  run default ""
  run memset_before_bwd "" LFS_DEBUG_FLAGS=128
  for v in tail_keep tail_early tail_ke tail_d8 tail_d2 tail_nt proj_noslp; do run $v $v; done
done 2>&1 | tee $OUT/ab_tail.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_default.json
python -c "
import json; d = json.load(open('$OUT/bench_default.json')); print('driver command:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('step_form'), d.get('ops_route', {}).get('ms_per_step'))"
bash tools/profile.sh r06_lease15 > $OUT/profile.log 2>&1; tail -60 $OUT/profile.log
# BASELINE configs[3] / configs[4] on the library that ships
mkdir -p $OUT/side
timeout 600 python bench.py --workload syn-c --views-per-rank 8 --steps 10 --warmup 3 --no-cpu-baseline --no-ops-route 2>$OUT/side/c4.err | tail -1 > $OUT/side/bench_config4.json
timeout 600 python bench.py --workload syn-d --strategy mcmc --bilateral-grid --loss l1_ssim --steps 20 --warmup 5 --no-cpu-baseline --no-ops-route 2>$OUT/side/c5.err | tail -1 > $OUT/side/bench_config5.json
for f in $OUT/side/bench_config4.json $OUT/side/bench_config5.json; do python -c "
import json; d = json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d['config'].get('step_form'), {k: v['avg_ms'] for k, v in list(d['kernels'].items())[:8]})"; done
