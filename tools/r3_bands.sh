#!/bin/bash
# same-box A/B of the workgroup -> XCD chunk mapping (LFS_XCD_BANDS 1 / 4 / 8 = default / 16): gpurun --timeout 900 -- 'bash tools/r3_bands.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
for r in 1 2; do for v in bands1 default bands4 bands16; do
  if [ $v = default ]; then unset LFS_GSPLAT_LIB; else export LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$v.so; fi
  timeout 200 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']; print('[$v]', d['value'], d['ms_per_step'], {n: k[n]['avg_ms'] for n in ('raster_bwd', 'raster_fwd', 'raster_cull') if n in k})"
done; done
