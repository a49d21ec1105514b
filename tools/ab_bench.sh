#!/bin/bash
# A/B on ONE box (boxes differ by +-5 %): bash tools/ab_bench.sh "<flags A>" "<flags B>" [repeats]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
for r in $(seq 1 ${3:-2}); do for f in "$1" "$2"; do
  timeout 200 python bench.py --no-cpu-baseline $f 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('[$f]', d['value'], d['ms_per_step'], {k: v['avg_ms'] for k, v in d['kernels'].items()})"
done; done
