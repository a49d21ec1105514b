#!/bin/bash
# Round 6, lease 38: which level does the tail kernel draw (0.29 or 0.32 - 0.34 ms, profiles/r06/tail_kernel_placement.txt) in fresh processes under different allocator settings?
# tools/tail_variance.py, first trainer, second window; 6 processes per setting, alternating
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r6_lease38; mkdir -p $OUT
for r in 1 2 3 4 5 6; do
  for cfg in default expandable nocache roundup; do
    unset PYTORCH_HIP_ALLOC_CONF PYTORCH_NO_HIP_MEMORY_CACHING PYTORCH_NO_CUDA_MEMORY_CACHING
    case $cfg in
      expandable) export PYTORCH_HIP_ALLOC_CONF=expandable_segments:True;;
      nocache) export PYTORCH_NO_HIP_MEMORY_CACHING=1 PYTORCH_NO_CUDA_MEMORY_CACHING=1;;
      roundup) export PYTORCH_HIP_ALLOC_CONF=roundup_power2_divisions:1;;
    esac
    echo "[$cfg] $(LFS_TAILVAR_TRAINERS=1 python tools/tail_variance.py 2 60 2>&1 | grep 'trainer 0 window 1' | sed 's/.*scopes timed) //')"
  done
done 2>&1 | tee $OUT/alloc_settings.txt
