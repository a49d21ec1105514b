"""ORACLE — TEST INFRASTRUCTURE ONLY. The input sets of the golden vectors generated from the reference's OWN kernels
(oracle/make_golden_refk.py -> tests/golden/refk_*.npz): one place for the generator and for the tests that replay them.
Inputs are stored in the .npz files as well, so the tests never depend on this module reproducing the same random bits."""
from __future__ import annotations

import numpy as np

from . import FISHEYE, GLOBAL, PINHOLE


def _gaussians(rng, N, spread=1.0, zmin=3.0, smin=0.01, smax=0.06, flat=None):
    means = rng.standard_normal((N, 3)).astype(np.float32) * spread
    means[:, 2] = np.abs(means[:, 2]) + zmin
    quats = rng.standard_normal((N, 4)).astype(np.float32)
    quats /= np.linalg.norm(quats, axis=-1, keepdims=True)      # Fwd.cu:314: "quats need to be normalized before passing in"
    scales = (rng.random((N, 3)) * (smax - smin) + smin).astype(np.float32)
    opac = (rng.random(N) * 0.8 + 0.1).astype(np.float32)
    if flat is not None:   # FLAT Gaussians (round 5): one random axis `flat` times thinner than the largest of the other two - the shape trained scenes are made of
        thin = rng.integers(0, 3, N)
        scales[np.arange(N), thin] = 0.0
        scales[np.arange(N), thin] = scales.max(-1) / np.float32(flat)
    return means, quats.astype(np.float32), scales, opac


def _viewmat(rng, angle=0.05, shift=0.1):
    ax = rng.standard_normal(3); ax /= np.linalg.norm(ax)
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(angle) * Kx + (1 - np.cos(angle)) * Kx @ Kx
    m = np.eye(4); m[:3, :3] = R; m[:3, 3] = rng.standard_normal(3) * shift
    return m.astype(np.float32)


def _K(f, W, H, C=1):
    return np.tile(np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], np.float32), (C, 1, 1))


# name -> dict(N, W, H, tile, C, cdim, bg, masks, camera_model, rs_type, radial, tangential, thin_prism, ut_params, rolling, seed, ...)
RASTER_CASES = {
    # BASELINE.json configs[0] shape: 10k Gaussians, 1 camera, 256x256 (SYN-A geometry: fx = fy = 200, identity view matrix, bg 0)
    "syn_a": dict(N=10000, W=256, H=256, seed=42, syn_a=True, bg=False),
    "rgb_bg": dict(N=1500, W=112, H=80, seed=1),
    "ragged_nobg": dict(N=1500, W=123, H=85, seed=2, bg=False),
    "cdim1": dict(N=1500, W=112, H=80, seed=3, cdim=1),
    "cdim4": dict(N=1500, W=112, H=80, seed=4, cdim=4),
    # (no C = 2 rasterizer case: the reference kernels index means / quats / scales with the FLATTENED id g in [0, C*N) - Fwd.cu:201-207 "only
    #  support 1 camera for now" - i.e. they read and, in the backward, write out of bounds for the second camera; run under the emulator that
    #  corrupted the heap. The HIP kernels and the oracle index geometry with g % N, tests/test_gpu_raster.py::test_raster_two_cameras.)
    "masks": dict(N=1500, W=112, H=80, seed=6, masks=True),
    "dense": dict(N=1500, W=112, H=80, seed=7, spread=0.4, smin=0.05, smax=0.3),
    "opencv_distortion": dict(N=1500, W=112, H=80, seed=8, radial=[-0.1, 0.02, 0.0, 0.0, 0.0, 0.0], tangential=[0.001, -0.002], thin_prism=[0.0005, 0.0, -0.0004, 0.0]),
    "fisheye": dict(N=1500, W=96, H=96, seed=9, camera_model=FISHEYE, radial=[0.01, -0.002, 0.0, 0.0]),
    "rolling_top_bottom": dict(N=1500, W=112, H=80, seed=10, rs_type=0, rolling=True),
    "rolling_right_left": dict(N=1500, W=112, H=80, seed=11, rs_type=3, rolling=True),
    # round 5: flat Gaussians - aspect ratio 10 .. 80 (every case above draws its three scales from one range: aspect <= 6). The regime in which the foot-vector form
    # of K8 needs its re-orthogonalisation step (csrc/raster.hip LFS_BWD_REORTH); the reference's cross-product form is well conditioned here.
    "flat10": dict(N=1500, W=112, H=80, seed=12, smin=0.02, flat=10.0),
    "flat20": dict(N=1500, W=112, H=80, seed=13, smin=0.02, flat=20.0),
    "flat40": dict(N=1500, W=112, H=80, seed=14, smin=0.02, flat=40.0),
    "flat80": dict(N=1500, W=112, H=80, seed=15, smin=0.02, flat=80.0),
    "flat40_fisheye": dict(N=1500, W=96, H=96, seed=16, smin=0.02, flat=40.0, camera_model=FISHEYE, radial=[0.01, -0.002, 0.0, 0.0]),
    "flat40_rolling": dict(N=1500, W=112, H=80, seed=17, smin=0.02, flat=40.0, rs_type=0, rolling=True),
}

PROJECTION_CASES = {
    "pinhole": dict(N=2500, W=320, H=200, seed=21),
    "pinhole_two_cameras_no_opacity": dict(N=1200, W=320, H=200, seed=22, C=2, no_opacity=True),
    "pinhole_any_sigma_point": dict(N=2000, W=200, H=160, seed=23, ut_params=(0.1, 2.0, 0.0, 0.1, 0.0), spread=3.0),
    "pinhole_compensations_clip": dict(N=2000, W=200, H=160, seed=24, calc_compensations=True, radius_clip=2.0, eps2d=0.3),
    "opencv_distortion": dict(N=2000, W=320, H=200, seed=25, radial=[-0.1, 0.02, 0.001, 0.01, -0.002, 0.0005], tangential=[0.001, -0.002], thin_prism=[0.0005, 0.0002, -0.0004, 0.0001]),
    "fisheye": dict(N=2000, W=256, H=256, seed=26, camera_model=FISHEYE, radial=[0.01, -0.002, 0.0005, 0.0], spread=2.5),
    "rolling_0": dict(N=1500, W=320, H=200, seed=27, rs_type=0, rolling=True),
    "rolling_1": dict(N=1500, W=320, H=200, seed=28, rs_type=1, rolling=True),
    "rolling_2": dict(N=1500, W=320, H=200, seed=29, rs_type=2, rolling=True),
    "rolling_3": dict(N=1500, W=320, H=200, seed=30, rs_type=3, rolling=True),
}


def build_inputs(cfg: dict) -> dict:
    """the arrays a case feeds to projection / rasterization (float32 / int32 numpy)"""
    rng = np.random.default_rng(cfg["seed"])
    N, W, H, C = cfg["N"], cfg["W"], cfg["H"], cfg.get("C", 1)
    cdim = cfg.get("cdim", 3)
    if cfg.get("syn_a"):   # lichtfeld_studio_amd.scenes.syn_a geometry, generated with numpy here (the bits are stored in the file)
        means = rng.standard_normal((N, 3)).astype(np.float32); means[:, 2] = np.abs(means[:, 2]) + 3
        quats = rng.standard_normal((N, 4)).astype(np.float32); quats /= np.linalg.norm(quats, axis=-1, keepdims=True)
        scales = (rng.random((N, 3)) * 0.05 + 0.01).astype(np.float32)
        opac = (rng.random(N) * 0.8 + 0.1).astype(np.float32)
        vm0 = np.eye(4, dtype=np.float32)[None]
        K = np.array([[[200.0, 0, 128], [0, 200.0, 128], [0, 0, 1]]], np.float32)
    else:
        means, quats, scales, opac = _gaussians(rng, N, spread=cfg.get("spread", 1.0), smin=cfg.get("smin", 0.01), smax=cfg.get("smax", 0.06), flat=cfg.get("flat"))
        vm0 = np.stack([_viewmat(rng, 0.05 + 0.1 * c, 0.1) for c in range(C)])
        K = _K(0.8 * W, W, H, C)
    vm1 = np.stack([_viewmat(rng, 0.12, 0.2) @ vm0[c] for c in range(C)]).astype(np.float32) if cfg.get("rolling") else None
    d = dict(means=means, quats=quats.astype(np.float32), scales=scales, opacities=opac, viewmats0=vm0, Ks=K,
             colors=rng.random((C, N, cdim)).astype(np.float32),
             opacities_cn=(np.tile(opac[None], (C, 1)) * rng.uniform(0.8, 1.0, (C, 1))).astype(np.float32))
    if vm1 is not None:
        d["viewmats1"] = vm1
    if cfg.get("bg", True):
        d["backgrounds"] = rng.random((C, cdim)).astype(np.float32)
    for k in ("radial", "tangential", "thin_prism"):
        if cfg.get(k) is not None:
            d[k] = np.tile(np.asarray(cfg[k], np.float32)[None], (C, 1))
    tile = cfg.get("tile", 16)
    if cfg.get("masks"):
        d["masks"] = rng.random((C, (H + tile - 1) // tile, (W + tile - 1) // tile)) > 0.3
    return d


def case_meta(cfg: dict) -> dict:
    return dict(W=cfg["W"], H=cfg["H"], tile=cfg.get("tile", 16), camera_model=cfg.get("camera_model", PINHOLE), rs_type=cfg.get("rs_type", GLOBAL),
                ut_params=np.asarray(cfg.get("ut_params", (0.1, 2.0, 0.0, 0.1, 1.0)), np.float32), eps2d=cfg.get("eps2d", 0.3),
                radius_clip=cfg.get("radius_clip", 0.0), calc_compensations=bool(cfg.get("calc_compensations", False)))
