"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

One training image of the 3DGUT hot path on the CPU, stage by stage, with every intermediate kept:
activations -> projection_ut_3dgs_fused -> spherical harmonics (+0.5, clamp_min 0) -> intersect_tile /
intersect_offset -> rasterize fwd -> clamped MSE against `target` -> rasterize bwd -> SH bwd -> activation
backward (raw-parameter gradients). Call order and constants are those of the reference's L2 glue
(`src/training/rasterization/rasterizer.cpp:176-181,224-344`, `rasterizer_autograd.cpp`). Used by
tests/test_gpu_headline_parity.py (HIP vs oracle at BASELINE.json's full sizes) and by bench.py's cpu_baseline leg
(`parity_vs_oracle`); the product never imports it.
"""
from __future__ import annotations

import time

import numpy as np

from . import (intersect_offset, intersect_tile, projection_ut_3dgs_fused, rasterize_bwd, rasterize_fwd,
               spherical_harmonics_bwd, spherical_harmonics_fwd)


def scene_arrays(scene):
    """numpy views of a lichtfeld_studio_amd.scenes.Scene (CPU tensors)."""
    return dict(means=scene.means.numpy(), raw_quats=scene.raw_quats.numpy(), raw_scales=scene.raw_scales.numpy(),
                raw_opacities=scene.raw_opacities.numpy(), sh0=scene.sh0.numpy(), shN=scene.shN.numpy(),
                viewmats=scene.viewmats.numpy(), Ks=scene.Ks.numpy(), width=scene.width, height=scene.height, sh_degree=scene.sh_degree)


def train_image(sa: dict, view: int, target_chw: np.ndarray, tile: int = 16, backward: bool = True, bwd_dtype=np.float32, loss_fn=None) -> dict:
    """-> dict of every stage's outputs; `seconds` = wall time of the whole image. `loss_fn(raw_render_chw, target_chw) -> (loss, dL/d raw_render_chw)`
    replaces the clamped MSE (tests/convergence_l1ssim.py passes the reference's L1 + D-SSIM photometric loss, trainer.cpp:115-128)."""
    t0 = time.perf_counter()
    W, H, deg = sa["width"], sa["height"], sa["sh_degree"]
    means, raw_q, raw_s, raw_o = sa["means"], sa["raw_quats"], sa["raw_scales"], sa["raw_opacities"]
    qn = np.linalg.norm(raw_q, axis=-1, keepdims=True)
    quats, scales = raw_q / qn, np.exp(raw_s)
    opac = (1.0 / (1.0 + np.exp(-raw_o))).astype(np.float32)
    sh = np.concatenate([sa["sh0"], sa["shN"]], 1)
    vm, Kmat = sa["viewmats"][view:view + 1], sa["Ks"][view:view + 1]
    out = dict(quats=quats, scales=scales, opacities=opac)
    radii, m2, d, con, _ = projection_ut_3dgs_fused(means, quats, scales, opac, vm, None, Kmat, W, H)
    out.update(radii=radii, means2d=m2, depths=d, conics=con)
    mask = (radii[0] > 0).all(-1)
    campos = np.linalg.inv(vm[0].astype(np.float64))[:3, 3].astype(np.float32)
    dirs = means - campos
    col = spherical_harmonics_fwd(deg, dirs, sh, mask)
    colors = np.maximum(col + 0.5, 0.0).astype(np.float32)
    out.update(visible=mask, colors=colors)
    tw, th = (W + tile - 1) // tile, (H + tile - 1) // tile
    tpg, ids, flat = intersect_tile(m2, radii, d, 1, tile, tw, th, True)
    offs = intersect_offset(ids, 1, tw, th)
    out.update(tiles_per_gauss=tpg, isect_ids=ids, flatten_ids=flat, offsets=offs)
    bg = np.zeros((1, 3), np.float32)
    args = (means, quats, scales, colors[None], opac[None], bg, None, W, H, tile, vm, None, Kmat, 0, 4, None, None, None, offs, flat)
    rc, ra, li = rasterize_fwd(*args)
    out.update(render=rc, alpha=ra, last_ids=li)
    raw = rc[0].transpose(2, 0, 1)
    img = np.clip(raw, 0.0, 1.0)
    out["loss"] = float(((img.astype(np.float64) - target_chw) ** 2).mean())
    if backward:
        if loss_fn is not None:
            out["loss"], v_img = loss_fn(np.ascontiguousarray(raw), target_chw)
            v_img = np.asarray(v_img, np.float32)
        else:
            v_img = (2.0 * (img - target_chw) / img.size * ((raw >= 0) & (raw <= 1))).astype(np.float32)
        v_rc = np.ascontiguousarray(v_img.transpose(1, 2, 0))[None]
        out["v_render"] = v_rc
        gm, gq, gs, gc, go = rasterize_bwd(*args, ra, li, v_rc, np.zeros_like(ra), dtype=bwd_dtype)
        out.update(v_means_raster=gm, v_quats=gq, v_scales=gs, v_colors=gc[0], v_opacities=go[0])
        g_col = np.where(col + 0.5 > 0, gc[0], 0.0).astype(np.float32)
        g_sh, g_dirs = spherical_harmonics_bwd(deg, dirs, sh, mask, g_col, True)
        out["grads"] = dict(means=(gm + g_dirs).astype(np.float32), sh0=g_sh[:, :1], shN=g_sh[:, 1:],
                            raw_scales=(gs * scales).astype(np.float32),
                            raw_quats=((gq - (gq * quats).sum(-1, keepdims=True) * quats) / qn).astype(np.float32),
                            raw_opacities=(go[0] * opac * (1 - opac)).astype(np.float32))
    out["seconds"] = time.perf_counter() - t0
    return out


def rel_l2(a, b) -> float:
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def compare_step(hip: dict, orc: dict) -> dict:
    """HIP results (numpy, same keys as train_image where present) against the oracle's: the numbers a bench line / test quotes."""
    r = {}
    if "radii" in hip:
        r["radii_mismatch"] = int((hip["radii"] != orc["radii"]).any(-1).sum())
        if "means2d" in hip:
            r["means2d_max_abs"] = float(np.abs(hip["means2d"] - orc["means2d"]).max())
    if "flatten_ids" in hip:
        same_len = len(hip["flatten_ids"]) == len(orc["flatten_ids"])
        r["n_isects"] = [int(len(hip["flatten_ids"])), int(len(orc["flatten_ids"]))]
        r["flatten_ids_exact"] = bool(same_len and np.array_equal(hip["flatten_ids"], orc["flatten_ids"]))
        if "isect_ids" in hip:
            r["isect_ids_exact"] = bool(same_len and np.array_equal(hip["isect_ids"], orc["isect_ids"]))
        if "offsets" in hip:
            r["offsets_exact"] = bool(np.array_equal(hip["offsets"], orc["offsets"]))
    if "n_isects" in hip:
        r["n_isects"] = [int(hip["n_isects"]), int(len(orc["flatten_ids"]))]
    if "render" in hip:
        d = np.abs(hip["render"] - orc["render"])
        r["image_mean_abs"] = float(d.mean())
        r["image_max_abs"] = float(d.max())
        r["image_px_beyond_1_255"] = int((d.max(-1) > 1 / 255 + 1e-4).sum())
        r["alpha_mean_abs"] = float(np.abs(hip["alpha"] - orc["alpha"]).mean())
        if "last_ids" in hip:
            r["last_ids_equal_frac"] = float((hip["last_ids"] == orc["last_ids"]).mean())
    if "loss" in hip:
        r["loss"] = [float(hip["loss"]), float(orc["loss"])]
    if "grads" in hip and "grads" in orc:
        r["grad_rel_l2"] = {k: rel_l2(hip["grads"][k], orc["grads"][k]) for k in orc["grads"]}
    return r
