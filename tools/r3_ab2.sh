# round-3 A/B session 2: projection specialisation tests, fastgs LDS-reduce variant, SH op kernels, config-4 line
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out/r03f
timeout 900 python3 -m pytest tests/test_gpu_projection_sh.py tests/test_gpu_small_ops.py tests/test_gpu_pipeline.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4
LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_fgldsred.so timeout 600 python3 -m pytest tests/test_gpu_fastgs.py tests/test_gpu_raster_reference.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
for r in 1 2; do for v in default fgldsred; do
  if [ $v = default ]; then unset LFS_GSPLAT_LIB; else export LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$v.so; fi
  timeout 300 python bench.py --rasterizer fastgs --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[fastgs $v]', d['value'], d['ms_per_step'], {k: v['avg_ms'] for k, v in d['kernels'].items()})"
done; done; unset LFS_GSPLAT_LIB
python tools/bench_sh_ops.py 2>/dev/null | tail -1 | tee gpurun_out/r03f/sh_ops.json
python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03f/bench_synb.json; python -c "
import json
d=json.loads(open('gpurun_out/r03f/bench_synb.json').read()); print('SYN-B', d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"
python bench.py --workload syn-c --views-per-rank 8 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03f/bench_config4.json; python -c "
import json
d=json.loads(open('gpurun_out/r03f/bench_config4.json').read()); print('CONFIG4', d['value'], d['ms_per_step'], {k:(v['avg_ms'],v['launches_per_step']) for k,v in d['kernels'].items()})"
