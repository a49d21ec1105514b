#!/bin/bash
# scatter variants A/B + tests.   gpurun --timeout 900 -- 'bash tools/r2_isect2.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/isect2; mkdir -p $OUT; cd $REPO
timeout 600 python -m pytest tests/test_gpu_intersect.py tests/test_gpu_fused.py tests/test_gpu_headline_parity.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -5
for i in 1 2; do for v in default; do
  if [ $v = default ]; then unset LFS_GSPLAT_LIB; else export LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$v.so; fi
  timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2> $OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']; print('[$v]', d['value'], d['ms_per_step'], {n: k[n]['avg_ms'] for n in ('isect_count_scan', 'isect_scatter', 'isect_tile_sort')})" || tail -5 $OUT/err.txt
done; done
unset LFS_GSPLAT_LIB
bash tools/r2_seq.sh
