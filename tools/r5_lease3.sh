#!/bin/bash
# Round 5, third lease: the differential fuzzer against the GPU library (conditioning-aware projection bar), A/B of sh_fwd with 12-byte row loads against the round-4 kernel,
# PSNR on the flat-disk task for the oracle seeds that exist, SH tests on the new kernel
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r5_lease3; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
timeout 300 python -m pytest tests/test_gpu_projection_sh.py tests/test_gpu_fused.py tests/test_gpu_gut_step.py "tests/test_gpu_refk_golden.py::test_hip_sh_matches_the_reference_kernels" -q -m gpu -p no:cacheprovider > $OUT/sh_tests.log 2>&1; echo "sh tests rc $?: $(tail -1 $OUT/sh_tests.log)"
bash tools/ab_lib.sh shfwd_r4 3 2>&1 | tee $OUT/ab_shfwd.txt
timeout 600 python tools/fuzz_emulated.py --gpu --oracle --flat 0.4 --cases 2500 --seconds 560 --seed 31 > $OUT/fuzz_gpu.txt 2>&1; echo "fuzz rc $?"; tail -44 $OUT/fuzz_gpu.txt | cut -c1-220
for v in default noreorth; do
  if [ $v = default ]; then unset LFS_GSPLAT_LIB; else export LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$v.so; fi
  S=$(python -c "import json; print(' '.join(sorted(json.load(open('profiles/r05/convergence_mse_flat50_oracle.json'))['seeds'], key=int)))")
  timeout 600 python tests/convergence_l1ssim.py --hip --loss mse --flat 50 --seeds $S --atomic-runs 2 --det-runs 1 --oracle-json profiles/r05/convergence_mse_flat50_oracle.json > $OUT/psnr_flat_$v.log 2>&1
  tail -1 $OUT/psnr_flat_$v.log > $OUT/psnr_flat_$v.json; python -c "import json; print('[$v flat]', json.dumps(json.load(open('$OUT/psnr_flat_$v.json'))['summary']))"
done
