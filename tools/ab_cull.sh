#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
timeout 300 python -m pytest tests/test_gpu_raster.py tests/test_gpu_fused.py -x -q 2>&1 | tail -1
for r in 1 2; do for v in default cull1 cull2; do
  if [ $v = default ]; then unset LFS_GSPLAT_LIB; else export LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$v.so; fi
  for w in "--workload syn-c --views-per-rank 8 --steps 6 --warmup 3" "--steps 200 --warmup 20"; do
  timeout 200 python bench.py --no-cpu-baseline $w 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['config']['workload'][:5], d['value'], d['ms_per_step'], d['kernels']['raster_cull'])"
done; done; done
