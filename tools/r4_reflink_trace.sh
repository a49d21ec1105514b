#!/bin/bash
# tools/r4_reflink_trace.sh: kernel trace of the REFERENCE'S OWN training-step sequence (oracle/_ref/libref_links_gpu.so: rasterize() + mse_loss + backward() +
# FusedAdam::step() + zero_grad, linked to liblfs_gsplat_torch.so) on SYN-B -> which launches of a step are the backend's and which are libtorch's glue
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/reflink_trace; mkdir -p $OUT
cat > /tmp/reflink_run.py <<PY
import sys, numpy as np
sys.path.insert(0, "$REPO"); sys.path.insert(0, "$REPO/tests")
import oracle
import lichtfeld_studio_amd
from lichtfeld_studio_amd import scenes
sc = scenes.syn_b(n=1_000_000, n_views=1)
vm = sc.viewmats[0].numpy(); K = sc.Ks[0].numpy()
args = (sc.means.numpy(), sc.sh0.numpy(), sc.shN.numpy(), sc.raw_scales.numpy(), sc.raw_quats.numpy(), sc.raw_opacities.numpy(), 3, 3, vm[:3, :3].copy(), vm[:3, 3].copy(),
        float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), sc.width, sc.height, (0.0, 0.0, 0.0), scenes.target_image(sc.height, sc.width, seed=43).numpy(),
        [1.6e-4, 2.5e-3, 2.5e-3 / 20, 5e-3, 1e-3, 5e-2], 3000)
r = oracle.ref_links_mse_train_steps(0, *args, 12, timed_from=4)
print("ms per step", r["ms_per_step"])
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python /tmp/reflink_run.py > $OUT/run.log 2>&1
tail -2 $OUT/run.log
python - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 12
tot_lfs = tot_other = 0.0
lines = []
for r in rows:
    ms = float(r["TotalDurationNs"]) / 1e6 / steps
    name = r["Name"].split("(")[0][:90]
    if "lfs::" in name: tot_lfs += ms
    else: tot_other += ms
    if ms > 0.004: lines.append(f"{ms:8.4f} ms/step  x{int(r['Calls'])/steps:5.1f}  {name}")
open("$OUT/summary.txt", "w").write(f"reference L2 sequence linked to liblfs_gsplat_torch.so, SYN-B, 12 steps (incl. 4 warm-up): kernel time per step: backend (lfs::) {tot_lfs:.3f} ms, everything else (libtorch) {tot_other:.3f} ms\n" + "\n".join(lines) + "\n")
print(open("$OUT/summary.txt").read())
PY
