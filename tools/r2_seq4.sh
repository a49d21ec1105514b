#!/bin/bash
# launch sequence of one config-4 step (8 views): gaps and the launches that are not ours.   gpurun --timeout 600 -- 'bash tools/r2_seq4.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/seq4; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python $REPO/bench.py --workload syn-c --views-per-rank 8 --steps 4 --warmup 3 --no-cpu-baseline --no-profile > $OUT/bench.log 2>&1
python $REPO/tools/step_sequence.py $OUT/trace sh_views_fwd_kernel > $OUT/step_sequence.txt 2>&1
awk '$3 > 3.0 || /step wall/ || /start_us/ || /not lfs/' $OUT/step_sequence.txt | head -60
grep -c "" $OUT/step_sequence.txt
rm -rf $OUT/trace
