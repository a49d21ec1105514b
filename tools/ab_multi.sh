#!/bin/bash
# same-box A/B of several library builds against the default one, alternating: bash tools/ab_multi.sh <repeats> <variant> [<variant> ...]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
R=$1; shift
for r in $(seq 1 $R); do for v in default "$@"; do
  if [ $v = default ]; then unset LFS_GSPLAT_LIB; else export LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$v.so; fi
  timeout 200 python bench.py --no-cpu-baseline --steps 300 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('[$v]', d['value'], d['ms_per_step'], {k: v['avg_ms'] for k, v in d['kernels'].items()})"
done; done
