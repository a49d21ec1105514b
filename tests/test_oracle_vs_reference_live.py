"""CPU, build container only (skips without oracle/_ref/libref_raster_full.so): RANDOMISED differential test of the oracle (the checker of the full-size GPU parity
tests) against the reference's whole gsplat library run on the CPU - beyond the fixed golden cases. Each seed draws a camera model (pinhole with / without OpenCV
distortion, fisheye), a shutter, image and tile sizes, a scene; the oracle's projection -> SH -> intersection -> rasterization forward + backward must agree with the
reference's own operators on the same inputs within the bars of tests/test_oracle_refk_golden.py. Integers exact where the inputs are the reference's own (the
intersection is fed the reference's projection), so a single differing radius cannot cascade."""
import numpy as np
import pytest

import oracle
from test_oracle_refk_golden import check_projection, raster_bwd_rows

pytestmark = pytest.mark.skipif(not oracle.have_ref("libref_raster_full.so"), reason="oracle/_ref/libref_raster_full.so not built (make -C oracle refgsplat; needs /root/reference)")


@pytest.fixture()
def whole(oracle_mod):
    oracle_mod.REFK_WHOLE_LIBRARY = True
    yield oracle_mod
    oracle_mod.REFK_WHOLE_LIBRARY = False


def draw(seed):
    g = np.random.default_rng(seed)
    N, W, H = int(g.integers(150, 500)), int(g.integers(40, 130)), int(g.integers(40, 110))
    model = int(g.choice([0, 0, 2]))                                                     # PINHOLE, PINHOLE, FISHEYE
    means = g.standard_normal((N, 3)) * [1.5, 1.2, 1.0]
    means[:, 2] = np.abs(means[:, 2]) + 2.5
    f = float(g.uniform(50, 110))
    K = np.array([[[f, 0, W / 2 + g.uniform(-2, 2)], [0, f * g.uniform(0.9, 1.1), H / 2 + g.uniform(-2, 2)], [0, 0, 1]]])
    def pose(scale):
        q = g.standard_normal(4) * scale + [1, 0, 0, 0]
        w, x, y, z = q / np.linalg.norm(q)
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        m = np.eye(4)
        m[:3, :3], m[:3, 3] = R, g.standard_normal(3) * 0.2
        return m
    vm0 = pose(0.08)
    rolling = int(g.choice([4, 4, 0, 1, 2, 3]))                                          # GLOBAL twice as likely as each rolling direction
    vm1 = None if rolling == 4 else vm0 @ np.linalg.inv(pose(0.01)) @ np.eye(4)
    radial = tangential = thin = None
    if model == 0 and g.random() < 0.5:
        radial, tangential, thin = g.standard_normal((1, 6)) * 0.02, g.standard_normal((1, 2)) * 0.005, g.standard_normal((1, 4)) * 0.003
    if model == 2:
        radial = g.standard_normal((1, 4)) * 0.02
    f32 = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
    return dict(N=N, W=W, H=H, tile=int(g.choice([8, 16, 16])), camera_model=model, rs_type=rolling, means=f32(means), quats=f32(g.standard_normal((N, 4))),
                scales=f32(np.exp(g.uniform(np.log(0.02), np.log(0.3), (N, 3)))), opacities=f32(1 / (1 + np.exp(-g.standard_normal(N) * 1.5))), viewmats0=f32(vm0[None]),
                viewmats1=f32(None if vm1 is None else vm1[None]), Ks=f32(K), radial=f32(radial), tangential=f32(tangential), thin_prism=f32(thin),
                degree=int(g.integers(0, 4)), coeffs=f32(g.standard_normal((N, 16, 3)) * 0.4), bg=f32(g.random((1, 3))) if g.random() < 0.6 else None,
                v_render=f32(g.standard_normal((1, H, W, 3))), v_alpha=f32(g.standard_normal((1, H, W, 1))))


@pytest.mark.parametrize("seed", range(16))
def test_random_scene_oracle_equals_the_reference_operators(whole, seed):
    o, d = whole, draw(seed)
    kw = dict(camera_model=d["camera_model"], rs_type=d["rs_type"], radial_coeffs=d["radial"], tangential_coeffs=d["tangential"], thin_prism_coeffs=d["thin_prism"])
    ref_p = o.refk_projection_ut(d["means"], d["quats"], d["scales"], d["opacities"], d["viewmats0"], d["viewmats1"], d["Ks"], d["W"], d["H"], **kw)
    our_p = o.projection_ut_3dgs_fused(d["means"], d["quats"], d["scales"], d["opacities"], d["viewmats0"], d["viewmats1"], d["Ks"], d["W"], d["H"], **kw)
    ref = dict(radii=ref_p[0], means2d=ref_p[1], depths=ref_p[2], conics=ref_p[3], compensations=None)
    vis = (ref["radii"] > 0).all(-1)
    for k in ("means2d", "depths", "conics"):                                            # rows of culled Gaussians are at::empty in the reference
        ref[k] = np.where(vis[..., None] if ref[k].ndim == 3 else vis, ref[k], 0)
    check_projection(ref, *our_p)
    # SH on the reference's visibility mask
    dirs = d["means"] - np.linalg.inv(d["viewmats0"][0].astype(np.float64))[:3, 3].astype(np.float32)
    mask = vis[0]
    col_ref, col = o.refk_sh_fwd(d["degree"], dirs, d["coeffs"], mask), o.spherical_harmonics_fwd(d["degree"], dirs, d["coeffs"], mask)
    assert np.abs(np.where(mask[:, None], col, 0) - col_ref).max() <= 2e-6 * max(np.abs(col_ref).max(), 1e-30)
    # intersection on the reference's projection: integers exact
    tw, th = (d["W"] + d["tile"] - 1) // d["tile"], (d["H"] + d["tile"] - 1) // d["tile"]
    tpg_r, ids_r, flat_r = o.refk_intersect_tile(ref["means2d"], ref["radii"], ref["depths"], d["tile"], tw, th, True)
    tpg, ids, flat = o.intersect_tile(ref["means2d"], ref["radii"], ref["depths"], 1, d["tile"], tw, th, True)
    assert np.array_equal(tpg, tpg_r) and np.array_equal(ids, ids_r) and np.array_equal(flat, flat_r)
    offs_r, offs = o.refk_intersect_offset(ids_r, 1, tw, th), o.intersect_offset(ids, 1, tw, th)
    assert np.array_equal(offs, offs_r)
    if len(ids) == 0:
        return
    # rasterization forward + backward on the reference's lists
    colors = np.maximum(col_ref + 0.5, 0)[None].astype(np.float32)
    quats = (d["quats"] / np.linalg.norm(d["quats"], axis=-1, keepdims=True)).astype(np.float32)      # the rasterizer takes normalised quaternions (Fwd.cu:313)
    args = (d["means"], quats, d["scales"], colors,
            d["opacities"][None], d["bg"], None, d["W"], d["H"], d["tile"], d["viewmats0"], d["viewmats1"], d["Ks"], d["camera_model"], d["rs_type"], d["radial"],
            d["tangential"], d["thin_prism"], offs_r, flat_r)
    rc_r, ra_r, li_r = o.refk_rasterize_fwd(*args)
    rc, ra, li = o.rasterize_fwd(*args)
    assert np.abs(rc - rc_r).max() < 1e-4 and np.abs(ra.reshape(ra_r.shape) - ra_r).max() < 1e-4 and (li == li_r).mean() > 0.999, seed
    g_r = o.refk_rasterize_bwd(*args, ra_r, li_r, d["v_render"], d["v_alpha"])
    g_o = o.rasterize_bwd(*args, ra_r, li_r, d["v_render"], d["v_alpha"])
    refd = dict(zip(("v_means", "v_quats", "v_scales", "v_colors", "v_opacities"), g_r))
    for nme, a, b in raster_bwd_rows(refd, g_o):
        if np.abs(b).max() == 0:
            continue
        e = float(np.linalg.norm(a.astype(np.float64) - b) / (np.linalg.norm(b.astype(np.float64)) + 1e-30))
        assert e < 2e-4, (seed, nme, e)


# ---- the default (fastgs / EWA) path: oracle_fastgs.hpp against the reference's own forward_wrapper / backward_wrapper on random scenes ----------------------------
fast_live = pytest.mark.skipif(not oracle.have_ref("libref_fast_raster.so"), reason="oracle/_ref/libref_fast_raster.so not built (make -C oracle reffast; needs /root/reference)")


@fast_live
@pytest.mark.parametrize("seed", range(8))
def test_random_scene_fastgs_oracle_equals_the_reference_wrappers(oracle_mod, seed):
    import math
    import test_oracle_refk_fastgs_golden as fg
    o = oracle_mod
    g = np.random.default_rng(100 + seed)
    N, W, H, deg = int(g.integers(300, 1200)), int(g.integers(48, 160)), int(g.integers(48, 128)), int(g.integers(0, 4))
    means = g.standard_normal((N, 3)) * g.uniform(0.5, 2.0)
    means[:, 2] = np.abs(means[:, 2]) + 2.5
    ang = g.uniform(-0.3, 0.3)
    R = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
    w2c = np.eye(4)
    w2c[:3, :3], w2c[:3, 3] = R, g.standard_normal(3) * 0.15
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    d = dict(means=f32(means), scales_raw=f32(np.log(g.uniform(0.02, 0.3, (N, 3)))), rot_raw=f32(g.standard_normal((N, 4))), opac_raw=f32(g.standard_normal(N) * 1.5),
             sh0=f32(g.standard_normal((N, 1, 3)) * 0.5), sh_rest=f32(g.standard_normal((N, 15, 3)) * 0.2), w2c=f32(w2c), cam_pos=f32(-R.T @ w2c[:3, 3]),
             active_sh_bases=(deg + 1) ** 2, W=W, H=H, fx=float(g.uniform(50, 90)), fy=float(g.uniform(50, 90)), cx=W / 2 + 0.3, cy=H / 2 - 0.2,
             g_image=f32(g.standard_normal((3, H, W))), g_alpha=f32(g.standard_normal((1, H, W))))
    dens = f32(g.uniform(0, 2, (2, N))) if seed % 2 == 0 else None
    o.REFK_FASTGS_WHOLE = True
    try:
        r = o.refk_fastgs_fwd_bwd(*fg.scene_args(d), d["g_image"], d["g_alpha"], densification_info=dens)
    finally:
        o.REFK_FASTGS_WHOLE = False
    for k, v in r.items():
        d["out_" + k] = v
    f = o.fastgs_forward(*fg.scene_args(d))
    fg.check_forward(d, f["image"], f["alpha"], len(f["ids"]), int((f["n_touched"] > 0).sum()))
    got = o.fastgs_backward(f, *fg.scene_args(d), d["g_image"], d["g_alpha"], densification_info=dens)
    for (gk, _), a in zip(fg.GRADS, got[:6]):
        ref = d["out_" + gk]
        a = np.asarray(a).reshape(ref.shape)
        if np.abs(ref).max() == 0:
            assert np.abs(a).max() == 0, gk
            continue
        assert fg.rel_l2(a, ref) < 5e-4, (seed, gk, fg.rel_l2(a, ref))
    if dens is not None:
        assert np.abs(np.asarray(got[6]) - d["out_densification_info"]).max() < 5e-4 * np.abs(d["out_densification_info"]).max()
