mkdir -p gpurun_out/r03e
timeout 600 python3 -m pytest tests/test_gpu_torch_ops.py tests/test_gpu_gut_step.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -8
python bench.py --path ops --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03e/bench_ops_path.json 2> gpurun_out/r03e/bench_ops_path.err; tail -3 gpurun_out/r03e/bench_ops_path.err
python -c "
import json
d=json.loads(open('gpurun_out/r03e/bench_ops_path.json').read().strip().splitlines()[-1]); print('OPS', d['value'], d['ms_per_step'], sum(v['avg_ms']*v['launches_per_step'] for v in d['kernels'].values()), {k:(v['avg_ms'],v['launches_per_step']) for k,v in d['kernels'].items()})"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03e/bench_step_path.json 2>/dev/null
python -c "
import json
d=json.loads(open('gpurun_out/r03e/bench_step_path.json').read().strip().splitlines()[-1]); print('STEP', d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()}, d['roofline'])"
