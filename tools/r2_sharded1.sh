#!/bin/bash
# the SH-sharded step with its real RCCL collectives on ONE GPU (world 1, LFS_DIST_FORCE_COLLECTIVES).   gpurun --timeout 600 -- 'bash tools/r2_sharded1.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; export LFS_DIST_FORCE_COLLECTIVES=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511
for f in "" "--replicated"; do
python bench.py --gpus 1 $([ -z "$f" ] && echo --sh-sharded) $f --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l[:1] == chr(123)][-1]); print('[$f]', d['value'], d['ms_per_step'], d['config']['parallelism'], d['collectives']['per_step']); print({k: (v['avg_ms'], v['launches_per_step']) for k, v in d['kernels'].items()}, sum(v['avg_ms'] * v['launches_per_step'] for v in d['kernels'].values()))"
done
cd /tmp && export TMPDIR=/tmp && mkdir -p $REPO/gpurun_out/sharded1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/sharded1/trace -o trace -- python $REPO/bench.py --gpus 1 --sh-sharded --steps 6 --warmup 4 --no-cpu-baseline --no-profile > $REPO/gpurun_out/sharded1/bench.log 2>&1
python $REPO/tools/step_sequence.py $REPO/gpurun_out/sharded1/trace > $REPO/gpurun_out/sharded1/step_sequence.txt 2>&1; cat $REPO/gpurun_out/sharded1/step_sequence.txt
rm -rf $REPO/gpurun_out/sharded1/trace
