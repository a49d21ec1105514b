#!/bin/bash
# Round 6, lease 5: tools/hbm_stream.hip with access widths (4 / 12 / 16 bytes), workgroup shapes and sh_fwd's coefficient stream
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r6_lease5; mkdir -p $OUT
hipcc -O3 --offload-arch=gfx950 tools/hbm_stream.hip -o /tmp/hbm_stream 2>/dev/null && timeout 300 /tmp/hbm_stream > $OUT/hbm_stream_ceiling.json; python - <<'PY'
import json
d = json.load(open('gpurun_out/r6_lease5/hbm_stream_ceiling.json'))
for r in d['sweep'][111:]:
    print(f"{r['kernel']:40s} wg/cu={r['workgroups_per_cu']} u={r['unroll']} {r['ms']:.4f} ms {r['tb_s']:.3f} TB/s")
print(d['best'])
PY
