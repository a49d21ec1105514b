cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6_lease23
for v in default r6base; do
  if [ $v = default ]; then unset LFS_GSPLAT_LIB; else export LFS_GSPLAT_LIB=$PWD/lichtfeld-studio_amd/liblfs_gsplat_$v.so; fi
  for mode in "" "--deterministic"; do for r in 1 2 3; do
    echo "== $v $mode run $r"; python tools/aniso_probe.py --gpu $mode 2>&1 | tail -6 | cut -c1-140
  done; done
done > gpurun_out/r6_lease23/aniso_noise.txt 2>&1
cat gpurun_out/r6_lease23/aniso_noise.txt | awk '/==/ {print} /0.0040|0.0010|0.0005/ {print}'
