#!/bin/bash
# Round 6, lease 34: the final source of the round (lease 31 + the dominant kernel timed through its dispatch packet): whole GPU suite, the driver's command x3, the rocprofv3
# evidence (trace + 4 PMC passes of the driver's command, headline path only), configs[3] / configs[4] side lines, the round-6 base library on the same box
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease34; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
timeout 1800 python -m pytest tests/ -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc $?: $(tail -1 $OUT/tests.log)"
grep -n "^FAILED\|^ERROR" $OUT/tests.log | head -20
for r in 1 2 3; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_default_$r.json
  python -c "
import json; d = json.load(open('$OUT/bench_default_$r.json')); print('driver command:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['config'].get('step_form'), d.get('ops_route', {}).get('ms_per_step'))"
done 2>&1 | tee $OUT/driver_command.txt
bash tools/profile.sh r06_lease34 > $OUT/profile.log 2>&1; head -40 gpurun_out/prof_r06_lease34/summary.txt | cut -c1-200
mkdir -p $OUT/side
timeout 600 python bench.py --workload syn-c --views-per-rank 8 --steps 10 --warmup 3 --no-cpu-baseline --no-ops-route 2>$OUT/side/c4.err | tail -1 > $OUT/side/bench_config4.json
timeout 600 python bench.py --workload syn-d --strategy mcmc --bilateral-grid --loss l1_ssim --steps 20 --warmup 5 --no-cpu-baseline --no-ops-route 2>$OUT/side/c5.err | tail -1 > $OUT/side/bench_config5.json
python -c "
import json
for f in ('bench_config4', 'bench_config5'):
    d = json.load(open('$OUT/side/' + f + '.json')); print(f, d['value'], d['ms_per_step'], d['config'].get('workload'), {k: v['avg_ms'] for k, v in list(d['kernels'].items())[:8]})"
LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_r6base.so timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-ops-route 2>/dev/null | tail -1 > $OUT/bench_r6base_same_box.json
python -c "
import json; d = json.load(open('$OUT/bench_r6base_same_box.json')); print('r6base, driver command, same box:', d['value'], d['ms_per_step'], d['roofline']['frac'])"
