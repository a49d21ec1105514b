#!/bin/bash
# The driver's round-end commands, verbatim, on a fresh lease (each gpurun call is one):  gpurun --timeout 1500 -- 'bash tools/r3_driver_verbatim.sh <tag>'
# -> gpurun_out/verbatim_<tag>/{pytest.log,smoke.log,bench.json}
TAG=${1:-a}; REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; OUT=gpurun_out/verbatim_$TAG; mkdir -p $OUT
python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
python3 -c 'import sys; sys.path.insert(0, "."); import __graft_entry__ as e
f = getattr(e, "smoke", None)
f(); print("__SMOKE_OK__")' > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?" >> $OUT/bench.err
tail -3 $OUT/pytest.log; tail -3 $OUT/smoke.log; python -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('BENCH', d['value'], d['ms_per_step'], 'kernel sum', round(sum(v['avg_ms']*v['launches_per_step'] for v in d['kernels'].values()),4), d['roofline']['frac'], d['library'])"
