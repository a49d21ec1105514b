#!/bin/bash
# Round-3 evidence in one GPU call: profile passes, launch sequence, other BASELINE configs, fastgs, drop-in route, PSNR criterion.  gpurun --timeout 2400 -- 'bash tools/r3_final.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; OUT=gpurun_out/r03_final; mkdir -p $OUT
bash tools/profile.sh r03 > $OUT/profile.log 2>&1
python tools/step_sequence.py gpurun_out/prof_r03/trace > $OUT/step_launch_sequence.txt 2>&1
python bench.py --steps 300 --warmup 20 > $OUT/bench_synb_steps300.json 2> $OUT/bench_synb_steps300.err
python bench.py --path ops --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_synb_ops_route.json 2>/dev/null
python bench.py --rasterizer fastgs --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_synb_fastgs.json 2>/dev/null
python bench.py --workload syn-c --views-per-rank 8 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_config4_synC_3M_8views_per_rank_1gpu.json 2>/dev/null
python bench.py --workload syn-d --strategy mcmc --bilateral-grid --loss l1_ssim --steps 100 --warmup 12 --no-cpu-baseline > $OUT/bench_config5_synD_2M_mcmc_bilateral_1gpu.json 2>/dev/null
LFS_DIST_FORCE_COLLECTIVES=1 python bench.py --gpus 1 --steps 50 --warmup 8 --no-cpu-baseline --replicated > $OUT/bench_synb_replicated_world1_rccl.json 2>/dev/null
LFS_DIST_FORCE_COLLECTIVES=1 python bench.py --gpus 1 --steps 50 --warmup 8 --no-cpu-baseline --sh-sharded > $OUT/bench_synb_sh_sharded_world1_rccl.json 2>/dev/null
timeout 1200 python tests/convergence_l1ssim.py --hip --seeds 0 1 2 3 4 --oracle-json profiles/r02/convergence_l1ssim_oracle.json > $OUT/convergence.log 2>$OUT/convergence.err
tail -1 $OUT/convergence.log > $OUT/convergence_l1ssim_hip.json
for i in 1 2 3; do timeout 300 python tests/convergence_check.py --iters 7000 --oracle-iters 0 2>/dev/null | tail -1 >> $OUT/convergence_mse_cxx_step_7k.jsonl; done
for f in $OUT/bench_*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['config'].get('parallelism'))
except Exception as e: print('$f', 'FAILED', e)"; done
tail -3 $OUT/convergence.log
