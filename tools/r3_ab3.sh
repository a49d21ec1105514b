REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out/r03g
timeout 900 python3 -m pytest tests/test_gpu_projection_sh.py tests/test_gpu_fused.py tests/test_gpu_refk_golden.py tests/test_gpu_fastgs.py tests/test_gpu_gut_step.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
python tools/bench_sh_ops.py 2>/dev/null | tail -1 | tee gpurun_out/r03g/sh_ops.json
for i in 1 2; do python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03g/bench_synb_$i.json; python -c "
import json
d=json.loads(open('gpurun_out/r03g/bench_synb_$i.json').read()); print('SYN-B', d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"; done
python bench.py --path ops --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03g/bench_ops.json; python -c "
import json
d=json.loads(open('gpurun_out/r03g/bench_ops.json').read()); print('OPS', d['value'], d['ms_per_step'], {k:(v['avg_ms'],v['launches_per_step']) for k,v in d['kernels'].items()})"
