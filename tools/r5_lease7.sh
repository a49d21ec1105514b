#!/bin/bash
# Round 5, seventh lease: is the decline of MCMC + 3DGUT training a property of the K8 defect or of the configuration? Repeated runs (float atomics: every run is another trajectory)
# of both libraries on the 12 500- and the 30 000-iteration schedule, held-out PSNR (3DGUT renderer) along the way
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r5_lease7; mkdir -p $OUT
D=/tmp/syn_colmap
python tools/make_synthetic_colmap.py $D --views 96 --width 1296 --height 840 --gaussians 300000 --points 60000 > $OUT/make.log 2>&1 || { tail -20 $OUT/make.log; exit 1; }
for it in 12500 30000; do
  runs=4; [ $it = 30000 ] && runs=2
  for r in $(seq 1 $runs); do for v in default noreorth; do
    if [ $v = default ]; then unset LFS_GSPLAT_LIB; else export LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$v.so; fi
    timeout 400 python tools/train_colmap.py -d $D --gut --strategy mcmc -i $it --eval --eval-every 2500 -o /tmp/scale_out_x > $OUT/mcmc_gut_${it}_${v}_run$r.json 2> $OUT/mcmc_gut_${it}_${v}_run$r.err
    python - <<PY
import json
try:
    d = json.loads(open("$OUT/mcmc_gut_${it}_${v}_run$r.json").read().strip().splitlines()[-1])
    print("[$it $v run $r] final", d["psnr_gut"], "curve", [(c["iteration"], c["psnr"]) for c in d.get("psnr_curve", [])], "flat>=10:", d["psnr_curve"][-1]["frac_aspect_ge_10"])
except Exception as e:
    print("[$it $v run $r] failed", e)
PY
  done; done
done
