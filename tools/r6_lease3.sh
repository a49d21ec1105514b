#!/bin/bash
# Round 6, lease 3: tools/valu_rate.hip with the rasterizer's other instruction forms (compares, selects, swaps, DPP, exec masks)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease3; mkdir -p $OUT
hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o /tmp/valu_rate 2>/dev/null && timeout 300 /tmp/valu_rate > $OUT/valu_rate.json; python - <<'PY'
import json
d = json.load(open('gpurun_out/r6_lease3/valu_rate.json'))
for r in d['rows']:
    if r['waves_per_simd'] in (1, 7):
        print(f"{r['form']:75s} w/simd={r['waves_per_simd']} {r['ns_per_instr_per_simd']:.3f} ns")
PY
