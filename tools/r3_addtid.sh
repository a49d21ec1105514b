#!/bin/bash
# wave_sum16_atomic_lds with ds_write_addtid_b32 + ds_read_b128 (default) against the ds_write2_b32 / ds_read2_b32 form (variant ds2): parity tests, then a same-box A/B.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
timeout 600 python -m pytest tests/test_gpu_refk_golden.py tests/test_gpu_headline_parity.py tests/test_gpu_raster.py tests/test_gpu_fused.py tests/test_gpu_gut_step.py tests/test_gpu_raster_reference.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4
bash tools/ab_lib.sh ds2 3
