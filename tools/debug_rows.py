"""Developer diagnostic (GPU): where do the quadrant-row kernels differ from the default ones? Prints, per scene variant, the number of differing
pixels, their position inside the 8x8 cell (row = quadrant, l = lane in row) and a few samples.  LFS_GSPLAT_LIB selects a library variant."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lichtfeld_studio_amd as lfs
import oracle
from lichtfeld_studio_amd import ops
from gpu_util import make_gaussians, n, pinhole_K, rel_l2, small_rotation_viewmat, t
from test_gpu_raster import _lists

lib = lfs.load_library()
print("library:", os.environ.get("LFS_GSPLAT_LIB", "default"))
for case in ("rgb", "rolling"):
    rng = np.random.default_rng(sum(map(ord, case)))
    N, W, H, ts, C, cdim = 4000, 160, 112, 16, 1, 3
    shutter, vm1 = lfs.ShutterType.GLOBAL, None
    if case == "rolling": shutter, vm1 = lfs.ShutterType(3), small_rotation_viewmat(rng, 0.12, 0.2)[None]
    means, quats, scales, opac = make_gaussians(rng, N)
    vm0 = np.stack([small_rotation_viewmat(rng, 0.05, 0.1)])
    K = pinhole_K(0.8 * W, W, H, C)
    colors = rng.random((C, N, cdim)).astype(np.float32)
    opacs = np.tile(opac[None], (C, 1))
    bgc = rng.random((C, cdim)).astype(np.float32)
    offs, flat = _lists(oracle, means, quats, scales, opac, vm0, vm1, K, W, H, ts, lfs.CameraModelType.PINHOLE, shutter, None, None, None)
    args = (t(means), t(quats), t(scales), t(colors), t(opacs), t(bgc), None, W, H, ts, t(vm0), t(vm1), t(K), lfs.CameraModelType.PINHOLE, None,
            shutter, None, None, None, t(offs, torch.int32), t(flat, torch.int32))
    rc, ra, li = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)
    for flags in (4, 5, 12):
        lib.lfs_set_debug_flags(flags)
        qc, qa, ql = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)
        qc2, qa2, ql2 = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)
        lib.lfs_set_debug_flags(1 if flags & 1 else 0)
        rc_, ra_, li_ = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)
        lib.lfs_set_debug_flags(0)
        d = (rc_ != qc).any(-1)[0] | (ra_ != qa)[0, ..., 0] | (li_ != ql)[0]
        ii, jj = torch.nonzero(d, as_tuple=True)
        print(f"[{case} flags={flags}] differing pixels {int(d.sum())} / {d.numel()}  (run-to-run identical: {bool(torch.equal(qc, qc2) and torch.equal(ql, ql2))}); "
              f"max |dcolor| {float((rc_ - qc).abs().max()):.3e}, last_ids differ {int((li_ != ql).sum())}, alpha differ {int((ra_ != qa).sum())}")
        if len(ii):
            ci, cj = (ii % 8).cpu().numpy(), (jj % 8).cpu().numpy()
            row = (ci // 4) * 2 + cj // 4
            l = (ci % 4) * 4 + cj % 4
            print("   by row:", np.bincount(row, minlength=4).tolist(), " by l:", np.bincount(l, minlength=16).tolist())
            big = (rc_ - qc).abs().amax(-1)[0] > 1e-4
            print("   pixels with |dcolor| > 1e-4:", int(big.sum()))
            for k in range(min(6, len(ii))):
                i, j = int(ii[k]), int(jj[k])
                print(f"   ({i},{j}) default c={rc_[0, i, j].tolist()} a={float(ra_[0, i, j, 0]):.6f} last={int(li_[0, i, j])} | rows c={qc[0, i, j].tolist()} a={float(qa[0, i, j, 0]):.6f} last={int(ql[0, i, j])}")
    # backward
    v_rc, v_ra = torch.randn_like(rc), torch.randn_like(ra)
    g0 = ops.rasterize_to_pixels_from_world_3dgs_bwd(*args, ra, li, v_rc, v_ra)
    lib.lfs_set_debug_flags(4)
    g1 = ops.rasterize_to_pixels_from_world_3dgs_bwd(*args, ra, li, v_rc, v_ra)
    lib.lfs_set_debug_flags(0)
    print(f"[{case}] bwd rel-L2 rows vs default:", {k: f"{rel_l2(n(b), n(a)):.2e}" for k, a, b in zip(["means", "quats", "scales", "colors", "opac"], g0, g1)})
