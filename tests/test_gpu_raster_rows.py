"""The quadrant-row rasterizer kernels (csrc/lfs_raster_rows.cuh, lfs_set_debug_flags bit 2): an opt-in alternative to the default 8x8-cell kernels
that round 2 verified on the GPU (after a DPP hazard fix: tools/debug_rows.py) and measured SLOWER (profiles/r02/raster_rows_vs_default_pmc.txt) - they
stay in the tree as a tested negative result. The forward has to be BIT-identical to the default kernels (same per-pixel operation sequence; the
quadrant lists only drop entries that cannot reach alpha >= 1/255 on any ray of the quadrant), the backward equal up to the float-atomic summation order."""
import os

import numpy as np
import pytest
import torch

from gpu_util import make_gaussians, n, pinhole_K, rel_l2, small_rotation_viewmat, t

pytestmark = pytest.mark.gpu


def _rows(lfs, fn, flags=4):
    lib = lfs.load_library()
    try:
        lib.lfs_set_debug_flags(flags)
        return fn()
    finally:
        lib.lfs_set_debug_flags(0)


@pytest.mark.parametrize("case", ["rgb", "ragged_nobg", "cdim1", "cdim4", "ts8", "ts32", "two_cameras", "masks", "dense", "rolling"])
def test_row_kernels_match_the_default_kernels(lfs, oracle_mod, case):
    from lichtfeld_studio_amd import ops
    from test_gpu_raster import _lists
    rng = np.random.default_rng(sum(map(ord, case)))
    N, W, H, ts, C, cdim, bg, masks, shutter, vm1, kw = 4000, 160, 112, 16, 1, 3, True, None, lfs.ShutterType.GLOBAL, None, {}
    if case == "ragged_nobg": W, H, bg = 203, 117, False
    if case == "cdim1": cdim = 1
    if case == "cdim4": cdim = 4
    if case == "ts8": ts = 8
    if case == "ts32": ts = 32
    if case == "two_cameras": C = 2
    if case == "dense": kw = dict(spread=0.4, smin=0.05, smax=0.3)
    if case == "rolling": shutter, vm1 = lfs.ShutterType(3), small_rotation_viewmat(rng, 0.12, 0.2)[None]
    means, quats, scales, opac = make_gaussians(rng, N, **kw)
    vm0 = np.stack([small_rotation_viewmat(rng, 0.05 + 0.1 * c, 0.1) for c in range(C)])
    K = pinhole_K(0.8 * W, W, H, C)
    colors = rng.random((C, N, cdim)).astype(np.float32)
    opacs = np.tile(opac[None], (C, 1))
    bgc = rng.random((C, cdim)).astype(np.float32) if bg else None
    tw, th = (W + ts - 1) // ts, (H + ts - 1) // ts
    if case == "masks": masks = rng.random((C, th, tw)) > 0.3
    offs, flat = _lists(oracle_mod, means, quats, scales, opac, vm0, vm1, K, W, H, ts, lfs.CameraModelType.PINHOLE, shutter, None, None, None)
    args = (t(means), t(quats), t(scales), t(colors), t(opacs), t(bgc), t(masks, torch.bool), W, H, ts, t(vm0), t(vm1), t(K), lfs.CameraModelType.PINHOLE, None,
            shutter, None, None, None, t(offs, torch.int32), t(flat, torch.int32))
    rc, ra, li = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)
    qc, qa, ql = _rows(lfs, lambda: ops.rasterize_to_pixels_from_world_3dgs_fwd(*args))
    assert float(ra.max()) > 0.05
    assert torch.equal(rc, qc) and torch.equal(ra, qa) and torch.equal(li, ql)
    v_rc, v_ra = torch.randn_like(rc), torch.randn_like(ra)
    g0 = ops.rasterize_to_pixels_from_world_3dgs_bwd(*args, ra, li, v_rc, v_ra)
    g1 = _rows(lfs, lambda: ops.rasterize_to_pixels_from_world_3dgs_bwd(*args, ra, li, v_rc, v_ra))
    for name, a, b in zip(["v_means", "v_quats", "v_scales", "v_colors", "v_opacities"], g0, g1):
        assert torch.isfinite(b).all(), name
        assert rel_l2(n(b), n(a)) < 1e-4, (name, rel_l2(n(b), n(a)))
    # the quadrant lists built in one pass from the tile lists (bit 3) instead of split from the cell lists
    mc, ma, ml = _rows(lfs, lambda: ops.rasterize_to_pixels_from_world_3dgs_fwd(*args), flags=12)
    assert torch.equal(rc, mc) and torch.equal(ra, ma) and torch.equal(li, ml)
    g2 = _rows(lfs, lambda: ops.rasterize_to_pixels_from_world_3dgs_bwd(*args, ra, li, v_rc, v_ra), flags=12)
    for name, a, b in zip(["v_means", "v_quats", "v_scales", "v_colors", "v_opacities"], g0, g2):
        assert rel_l2(n(b), n(a)) < 1e-4, ("merged lists", name, rel_l2(n(b), n(a)))


def test_row_kernels_full_size_and_fused_step(lfs):
    """SYN-B through the model-level rasterizer (prepared backward, fused MSE): image bit-identical, gradients close."""
    from lichtfeld_studio_amd import scenes
    from lichtfeld_studio_amd.rasterizer import Camera, SplatModel, rasterize
    dev = torch.device("cuda:0")
    sc = scenes.syn_b(n=1_000_000, n_views=4).to(dev)
    model = SplatModel(sc.means, sc.sh0, sc.shN, sc.raw_scales, sc.raw_quats, sc.raw_opacities, 3)
    cam = Camera(sc.viewmats[1:2].contiguous(), sc.Ks[1:2].contiguous(), sc.width, sc.height)
    with torch.no_grad():
        a = rasterize(cam, model, torch.zeros(3, device=dev))
        b = _rows(lfs, lambda: rasterize(cam, model, torch.zeros(3, device=dev)))
    assert torch.equal(a.image, b.image) and torch.equal(a.alpha, b.alpha)
