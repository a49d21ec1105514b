#!/bin/bash
# Round 6, lease 2: VALU / SALU issue cost per instruction form (tools/valu_rate.hip) and the first real-kernel experiment: record prefetch in the list walker.
#   gpurun --timeout 900 -- 'bash tools/r6_lease2.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease2; mkdir -p $OUT
hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o /tmp/valu_rate 2>/dev/null && timeout 200 /tmp/valu_rate > $OUT/valu_rate.json; python - <<'PY'
import json
d = json.load(open('gpurun_out/r6_lease2/valu_rate.json'))
for r in d['rows']:
    print(f"{r['form']:40s} w/simd={r['waves_per_simd']} {r['ns_per_instr_per_simd']:.3f} ns  {r['memtime_ticks_per_instr_per_simd']:.3f} ticks")
PY
bash tools/ab_lib.sh walkpf 3 2>&1 | tee $OUT/ab_walkpf.txt
