#!/bin/bash
# where does raster_bwd's time go? measurement-only builds (wrong gradients): lim1 = no atomics, lim2 = no cross-lane reduction and no atomics
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
for r in 1 2; do for v in default lim1 lim2; do
  if [ $v = default ]; then unset LFS_GSPLAT_LIB; else export LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$v.so; fi
  timeout 200 python bench.py --no-cpu-baseline --steps 100 --warmup 10 --no-inline-all 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l[:1] == chr(123)][-1]); print('[$v]', d['ms_per_step'], d['kernels']['raster_bwd'], d['kernels']['raster_fwd']['avg_ms'])"
done; done
