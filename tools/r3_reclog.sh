#!/bin/bash
# log2-opacity scaled records (LFS_REC_LOG2=1, default) against the round-2 records: parity tests on the default build, then a same-box A/B.  gpurun --timeout 900 -- 'bash tools/r3_reclog.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
timeout 600 python -m pytest tests/test_gpu_refk_golden.py tests/test_gpu_headline_parity.py tests/test_gpu_raster.py tests/test_gpu_fused.py tests/test_gpu_gut_step.py tests/test_gpu_raster_reference.py tests/test_gpu_torch_ops.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -15
bash tools/ab_lib.sh reclin 2
