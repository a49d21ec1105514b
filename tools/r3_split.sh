#!/bin/bash
# The C++ split step on the non-MSE configurations: parity tests, then BASELINE config 5 and the forced-RCCL multi-rank lines.  gpurun --timeout 900 -- 'bash tools/r3_split.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; OUT=gpurun_out/r03_split; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_gut_step.py tests/test_gpu_fused.py tests/test_gpu_bilateral.py tests/test_gpu_strategies.py -x -q -m gpu -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
python bench.py --workload syn-d --strategy mcmc --bilateral-grid --loss l1_ssim --steps 100 --warmup 12 --no-cpu-baseline > $OUT/bench_config5_synD_2M_mcmc_bilateral_1gpu.json 2>$OUT/config5.err
python bench.py --loss l1_ssim --steps 100 --warmup 12 --no-cpu-baseline > $OUT/bench_synb_l1ssim.json 2>/dev/null
LFS_DIST_FORCE_COLLECTIVES=1 python bench.py --gpus 1 --steps 50 --warmup 8 --no-cpu-baseline --replicated > $OUT/bench_synb_replicated_world1_rccl.json 2>/dev/null
LFS_DIST_FORCE_COLLECTIVES=1 python bench.py --gpus 1 --steps 50 --warmup 8 --no-cpu-baseline --sh-sharded > $OUT/bench_synb_sh_sharded_world1_rccl.json 2>/dev/null
for f in $OUT/bench_*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['config'].get('parallelism'))
except Exception as e: print('$f', 'FAILED', e)"; done
tail -3 $OUT/config5.err
