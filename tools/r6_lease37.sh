#!/bin/bash
# Round 6, lease 37: the tail kernel with its workgroups transposed onto the chunks of 64 Gaussians (-DLFS_TAIL_SPREAD=16 / 125, variants built with tools/build_variant.py from a
# source state that was not committed): step tests on the variants, tools/tail_variance.py (three trainers per process) for default / P = 16 / P = 125, twice
cd ${GRAFT_REPO_ROOT:-/root/repo}
for V in r6spread16 r6spread125; do LFS_GSPLAT_LIB=$PWD/lichtfeld-studio_amd/liblfs_gsplat_$V.so timeout 600 python -m pytest tests/test_gpu_gut_step.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -1; done
for r in 1 2; do for V in default r6spread16 r6spread125; do
  if [ $V = default ]; then unset LFS_GSPLAT_LIB; else export LFS_GSPLAT_LIB=$PWD/lichtfeld-studio_amd/liblfs_gsplat_$V.so; fi
  echo "== $V"; python tools/tail_variance.py 2 100 2>&1 | grep "window 1" | cut -c1-200
done; done
