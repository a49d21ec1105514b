#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_headline_parity.py tests/test_gpu_pipeline.py tests/test_gpu_dp2.py tests/test_gpu_strategies.py tests/test_gpu_rccl_world1.py tests/test_gpu_convergence.py -x -q 2>&1 | tail -3
for i in 1 2; do
python bench.py --workload syn-c --views-per-rank 8 --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config4', d['value'], d['ms_per_step'])"
python bench.py --workload syn-d --strategy mcmc --bilateral-grid --loss l1_ssim --steps 100 --warmup 12 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config5', d['value'], d['ms_per_step'])"
LFS_DIST_FORCE_COLLECTIVES=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 python bench.py --gpus 1 --sh-sharded --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sharded world 1', d['value'], d['ms_per_step'], {k: v['avg_ms'] for k, v in d['kernels'].items()})"
done
