#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$REPO"; mkdir -p gpurun_out/rowsdbg
for v in "" rows_NO_ASM rows_NOP; do
  if [ -n "$v" ]; then export LFS_GSPLAT_LIB=$REPO/lichtfeld-studio_amd/liblfs_gsplat_$v.so; fi
  timeout 200 python tools/debug_rows.py > gpurun_out/rowsdbg/out_${v:-default}.txt 2>&1
  cat gpurun_out/rowsdbg/out_${v:-default}.txt | tail -42
done
