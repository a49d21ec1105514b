#!/bin/bash
# Round 5, first GPU lease: K8's re-orthogonalised foot vector as the default build - hardware evidence.
#   gpurun --timeout 1500 -- 'bash tools/r5_lease1.sh'
# 1. the new / changed parity tests on the shipped library: flat-Gaussian reference-kernel goldens, the aniso probe, the flat-disk SYN-B headline case, culling on flat disks,
#    the edge sizes (never on hardware before the driver's round-4 run), the libtorch staging-slot tests
# 2. same-box A/B of the bench: default (reorth) against liblfs_gsplat_noreorth.so
# 3. the differential fuzzer against the GPU library (40 % flat-disk cases)
# 4. the whole -m gpu suite on the default library
# 5. rocprofv3 trace + PMC passes of the bench command (tools/profile.sh) on the library that ships
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=gpurun_out/r5_lease1; mkdir -p $OUT
python -c "import lichtfeld_studio_amd as l; print(l.load_library().lfs_version().decode())" 2>&1 | tail -1 | tee $OUT/library.txt
timeout 600 python -m pytest tests/test_gpu_000_canary.py tests/test_gpu_aniso.py "tests/test_gpu_refk_golden.py::test_hip_rasterization_matches_reference_kernel" \
  "tests/test_gpu_raster.py::test_cell_culling_is_conservative" tests/test_gpu_zz_edge_sizes.py tests/test_gpu_torch_ops.py tests/test_gpu_headline_parity.py \
  -q -s -m gpu -p no:cacheprovider > $OUT/new_tests.log 2>&1
echo "new tests rc $?: $(tail -1 $OUT/new_tests.log)"; grep -n "FAILED\|Error\|flat\|aniso\|48x48" $OUT/new_tests.log | cut -c1-260 | head -80
# the same flat-Gaussian tests on the library WITHOUT the step: they must fail there (the tests have teeth)
LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_noreorth.so timeout 300 python -m pytest tests/test_gpu_aniso.py "tests/test_gpu_refk_golden.py::test_hip_rasterization_matches_reference_kernel" \
  "tests/test_gpu_headline_parity.py" -q -s -m gpu -p no:cacheprovider -k "flat or aniso" > $OUT/new_tests_noreorth.log 2>&1
echo "the flat tests on the noreorth library rc $? (expected: failures): $(tail -1 $OUT/new_tests_noreorth.log)"; grep -n "syn_b_flat\] bwd\|^FAILED" $OUT/new_tests_noreorth.log | cut -c1-220 | head -30
bash tools/ab_lib.sh noreorth 3 2>&1 | tee $OUT/ab_reorth.txt
timeout 420 python tools/fuzz_emulated.py --gpu --oracle --flat 0.4 --cases 2000 --seconds 380 --seed 31 > $OUT/fuzz_gpu.txt 2>&1; echo "fuzz rc $?"; tail -40 $OUT/fuzz_gpu.txt | cut -c1-200
LFS_NOISE_LOG=$REPO/$OUT/noise.jsonl timeout 600 python -m pytest tests/ -q -m gpu -p no:cacheprovider -x > $OUT/suite.log 2>&1; echo "suite rc $?: $(tail -1 $OUT/suite.log)"; grep -n "FAILED\|mean gap" $OUT/suite.log | head
bash tools/profile.sh r05a > $OUT/profile.log 2>&1; tail -60 gpurun_out/prof_r05a/summary.txt
