#!/usr/bin/env python3
"""What a fast-math build of the projection kernel changes (round 6): run lfs_projection_ut_3dgs_fused on one view of SYN-B with the library named by
LFS_GSPLAT_LIB (default build when unset), save or compare its outputs, and time 50 launches.

    python tools/proj_fast_probe.py --save /tmp/proj_default.npz
    LFS_GSPLAT_LIB=.../liblfs_gsplat_projfast.so python tools/proj_fast_probe.py --compare /tmp/proj_default.npz

The bars of the reference's own test (tests/test_garden_data.cpp:247-275): radii +-1, floats 1e-4."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lichtfeld_studio_amd as lfs  # noqa: E402
from lichtfeld_studio_amd import ops, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--save"); ap.add_argument("--compare"); ap.add_argument("--n", type=int, default=1_000_000)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    sc = scenes.syn_b(n=a.n)
    quats = torch.nn.functional.normalize(sc.raw_quats, dim=-1).to(dev)
    scales, opac = sc.raw_scales.exp().to(dev), torch.sigmoid(sc.raw_opacities).to(dev)
    means = sc.means.to(dev)
    vm, K = sc.viewmats[0:1].contiguous().to(dev), sc.Ks[0:1].contiguous().to(dev)
    run = lambda: ops.projection_ut_3dgs_fused(means, quats, scales, opac, vm, None, K, sc.width, sc.height, 0.3, 0.01, 1e4, 0.0, True, lfs.CameraModelType.PINHOLE)
    out = run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    res = {"library": lfs.load_library().lfs_version().decode(), "ms_per_launch_with_allocs": e0.elapsed_time(e1) / 50}
    arrs = {k: v.cpu().numpy() for k, v in zip(("radii", "means2d", "depths", "conics", "comp"), out)}
    if a.save:
        np.savez(a.save, **arrs)
    if a.compare:
        ref = np.load(a.compare)
        r0, r1 = ref["radii"][0], arrs["radii"][0]
        vis0, vis1 = (r0 > 0).all(-1), (r1 > 0).all(-1)
        both = vis0 & vis1
        d = np.abs(r0[both] - r1[both])
        res.update(visible_ref=int(vis0.sum()), visibility_flips=int((vis0 != vis1).sum()), radii_differ=int((d > 0).any(-1).sum()), radii_max_abs=int(d.max()))
        for k in ("means2d", "depths", "conics", "comp"):
            x, y = ref[k][0][both], arrs[k][0][both]
            res[k + "_max_abs"] = float(np.abs(x - y).max()); res[k + "_max_rel"] = float((np.abs(x - y) / np.maximum(np.abs(x), 1e-6)).max())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
