#!/bin/bash
# launch sequence of one steady-state SYN-B step:   gpurun --timeout 600 -- 'bash tools/r2_seq.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/seq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python $REPO/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-profile > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-300
python $REPO/tools/step_sequence.py $OUT/trace > $OUT/step_sequence.txt 2>&1; cat $OUT/step_sequence.txt
rm -rf $OUT/trace
