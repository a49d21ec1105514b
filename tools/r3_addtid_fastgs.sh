REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
timeout 600 python -m pytest tests/test_gpu_fastgs.py tests/test_gpu_refk_golden.py tests/test_gpu_raster_reference.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
for r in 1 2 3; do for v in default fgds2; do
  if [ $v = default ]; then unset LFS_GSPLAT_LIB; else export LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$v.so; fi
  timeout 200 python bench.py --rasterizer fastgs --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('[$v]', d['value'], d['ms_per_step'], {k: v['avg_ms'] for k, v in d['kernels'].items() if 'blend' in k})"
done; done
