#!/bin/bash
# Round 5, fifth lease: A/B of the finish pass' workgroup size, PSNR on the flat-disk task (8 oracle seeds) for the default and the noreorth library, a second fuzz seed
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r5_lease5; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
for v in finish512 finish1024; do bash tools/ab_lib.sh $v 3 2>&1 | tee $OUT/ab_$v.txt; done
for v in default noreorth; do
  if [ $v = default ]; then unset LFS_GSPLAT_LIB; else export LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$v.so; fi
  S=$(python -c "import json; print(' '.join(sorted(json.load(open('profiles/r05/convergence_mse_flat50_oracle.json'))['seeds'], key=int)))")
  timeout 900 python tests/convergence_l1ssim.py --hip --loss mse --flat 50 --seeds $S --atomic-runs 2 --det-runs 1 --oracle-json profiles/r05/convergence_mse_flat50_oracle.json > $OUT/psnr_flat_$v.log 2>&1
  tail -1 $OUT/psnr_flat_$v.log > $OUT/psnr_flat_$v.json; python -c "import json; print('[$v flat]', json.dumps(json.load(open('$OUT/psnr_flat_$v.json'))['summary']))"
done
unset LFS_GSPLAT_LIB
FUZZ_SEED=47 FUZZ_S=400 FUZZ_SEC=360 FUZZ_CASES=3000 bash tools/r5_fuzz_only.sh; cp gpurun_out/r5_fuzz/fuzz_gpu.txt $OUT/fuzz_gpu_seed47.txt
