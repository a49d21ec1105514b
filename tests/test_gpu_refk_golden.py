"""GPU: the HIP kernels (through the C ABI) against golden vectors generated from the REFERENCE'S OWN device kernels
(tests/golden/refk_*.npz - gsplat/ProjectionUT3DGSFused.cu, RasterizeToPixelsFromWorld3DGS{Fwd,Bwd}.cu, RelocationCUDA.cu, QuatToRotmatCUDA.cu
and fastgs' adam_kernels.cuh run on the CPU by oracle/ref_kernels.cpp; generator: oracle/make_golden_refk.py). No oracle in between: this is
"HIP vs the reference" for the ops SURVEY.md §8c lists as unpinned, incl. BASELINE.json configs[0]'s shape (10k Gaussians, 256x256: `syn_a`).
Bars: SURVEY.md §8c - integer radii +-1 on < 0.2 %, means2d 1e-2 px (UT fp32 noise floor), forward mean |diff| <= 2e-6 and last_ids >= 99.9 %,
backward relative L2 <= 2e-4 with the alpha-threshold flip rows counted (gpu_util.rows_check), Adam bit-exact. Round 5: six cases of FLAT Gaussians (aspect 10 - 80,
pinhole / fisheye / rolling shutter: `flat*`), same bars - except that a tensor on which the reference kernel itself is further than 1e-4 from the fp64 oracle
(dL/dscales at aspect >= 40) is held to the fp64 oracle at twice the reference kernel's own distance."""
import os

import numpy as np
import pytest
import torch

import refk_util as ru
from gpu_util import n, rows_check, t
from test_oracle_refk_golden import check_projection, check_raster_fwd, raster_bwd_rows

pytestmark = pytest.mark.gpu
PROJ = ru.projection_cases()


@pytest.mark.parametrize("name", sorted(PROJ))
def test_hip_projection_matches_reference_kernel(lfs, name):
    from lichtfeld_studio_amd import ops
    d = PROJ[name]
    ut = lfs.UnscentedTransformParameters(*[float(x) for x in d["ut_params"][:4]], bool(d["ut_params"][4]))
    g = ops.projection_ut_3dgs_fused(t(d["means"]), t(d["quats"]), t(d["scales"]), None if d["no_opacity"] else t(d["opacities"]), t(d["viewmats0"]),
                                     t(d["viewmats1"]), t(d["Ks"]), d["W"], d["H"], d["eps2d"], 0.01, 1e4, d["radius_clip"], d["calc_compensations"],
                                     lfs.CameraModelType(d["camera_model"]), ut, lfs.ShutterType(d["rs_type"]), t(d["radial"]), t(d["tangential"]), t(d["thin_prism"]))
    check_projection(d, *[None if x is None else n(x) for x in g])


@pytest.mark.parametrize("name", ru.RASTER_CASES)
def test_hip_rasterization_matches_reference_kernel(lfs, name):
    from lichtfeld_studio_amd import ops
    d = ru.raster_case(name)
    args = (t(d["means"]), t(d["quats"]), t(d["scales"]), t(d["colors"]), t(d["opacities_cn"]), t(d["backgrounds"]), t(d["masks"], torch.bool), d["W"], d["H"],
            d["tile"], t(d["viewmats0"]), t(d["viewmats1"]), t(d["Ks"]), lfs.CameraModelType(d["camera_model"]), None, lfs.ShutterType(d["rs_type"]),
            t(d["radial"]), t(d["tangential"]), t(d["thin_prism"]), t(d["offsets"], torch.int32), t(d["flatten_ids"], torch.int32))
    rc, ra, li = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)
    check_raster_fwd(d, n(rc), n(ra), n(li), mean_bar=2e-6)
    g = ops.rasterize_to_pixels_from_world_3dgs_bwd(*args, t(d["alpha"]), t(d["last_ids"], torch.int32), t(d["v_render"]), t(d["v_alpha"]))
    o64 = None
    for k, (nme, a, b) in enumerate(raster_bwd_rows(d, [n(x) for x in g])):
        e, flips, rest = rows_check(a, b, bar=2e-4, max_flips=2)
        print(f"refk {name} {nme}: rel-L2 {e:.2e}, flip rows {flips}, without them {rest:.2e}")
        assert np.isfinite(a).all()
        if rest < 2e-4:
            continue
        # Flat Gaussians (round 5, aspect 40 - 80): dL/dscale of the THIN axis is ill-conditioned in fp32 in the reference's own arithmetic as well - its kernel sits
        # 2 - 3e-4 away from the fp64 oracle on `flat80` (two fp32 results that far from the truth are up to the sum of both apart). Where the 2e-4 bar against the
        # reference kernel does not hold, the test is the one that matters: the HIP result must be as close to the TRUTH (fp64 oracle) as the reference kernel is,
        # up to a factor 2 - and only the flat cases may take this branch.
        assert name.startswith("flat"), (nme, e, flips, rest)
        if o64 is None:
            import oracle
            o64 = raster_bwd_rows(d, oracle.rasterize_bwd(*ru.oracle_raster_args(d), d["alpha"], d["last_ids"], d["v_render"], d["v_alpha"], dtype=np.float64))
        truth = o64[k][1]
        e_ref = rows_check(b, truth, bar=2e-4, max_flips=2)[2]
        e_hip, flips64, rest64 = rows_check(a, truth, bar=max(2e-4, 2 * e_ref), max_flips=2)
        print(f"refk {name} {nme}: against the fp64 oracle HIP {rest64:.2e} ({flips64} flip rows), the reference kernel {e_ref:.2e}")
        assert rest64 < max(2e-4, 2 * e_ref), (nme, e_hip, flips64, rest64, e_ref)


def test_hip_small_ops_match_reference_kernels(lfs):
    from lichtfeld_studio_amd import ops
    s = ru.small_ops()
    no, ns = ops.relocation(t(s["reloc_opacities"]), t(s["reloc_scales"]), t(s["reloc_ratios"], torch.int32), t(s["binoms"]), 51)
    np.testing.assert_allclose(n(no), s["reloc_new_opacities"], rtol=1e-5, atol=1e-7)
    small = s["reloc_ratios"] <= 12     # larger ratios sum alternating binomial terms: ill-conditioned in fp32 on every implementation
    np.testing.assert_allclose(n(ns)[small], s["reloc_new_scales"][small], rtol=2e-3)
    ok = np.isfinite(s["reloc_new_scales"]).all(-1)
    assert np.median(np.abs(n(ns)[ok] - s["reloc_new_scales"][ok]) / (np.abs(s["reloc_new_scales"][ok]) + 1e-12)) < 1e-3
    m = t(s["noise_means"]).clone()
    ops.add_noise(t(s["noise_raw_opacities"]), t(s["noise_raw_scales"]), t(s["noise_raw_quats"]), t(s["noise_noise"]), m, float(s["noise_lr"]))
    moved = np.abs(s["noise_means_out"] - s["noise_means"]).max()
    np.testing.assert_allclose(n(m), s["noise_means_out"], rtol=1e-5, atol=1e-6 * max(1.0, moved))
    np.testing.assert_allclose(n(ops.quats_to_rotmats(t(s["quats"]))), s["rotmats"], rtol=0, atol=1e-6)
    lr, b1, b2, eps = [float(x) for x in s["adam_hyper"]]
    p, mm, v = t(s["adam_p0"]).clone(), torch.zeros(s["adam_p0"].shape, device="cuda:0"), torch.zeros(s["adam_p0"].shape, device="cuda:0")
    for k in range(1, 6):
        bc1, bc2 = float(np.float32(1.0 / (1.0 - b1 ** k))), float(np.float32(1.0 / np.sqrt(1.0 - b2 ** k)))
        ops.adam_step_wrapper(p, mm, v, t(s["adam_grads"][k - 1]), lr, b1, b2, eps, bc1, bc2)
        assert np.array_equal(n(p), s[f"p{k}"]) and np.array_equal(n(mm), s[f"m{k}"]) and np.array_equal(n(v), s[f"v{k}"]), k   # bit-exact


# ---- the fastgs (EWA) rasterizer against the reference's own fastgs code (tests/golden/refk_fastgs.npz) ---------------------------------------
import test_oracle_refk_fastgs_golden as fg  # noqa: E402


@pytest.mark.parametrize("name", sorted(fg.CASES))
def test_hip_fastgs_matches_the_reference_rasterizer(lfs, name):
    """csrc/fastgs_*.hip through fastgs.forward_wrapper / backward_wrapper against forward.cu / backward.cu of the reference run on the CPU: instance
    count identical, image / alpha to the fast-exp noise, gradients to 5e-4 with threshold-flip rows counted, densification_info accumulated alike."""
    import torch
    from gpu_util import n, rows_check, t
    from lichtfeld_studio_amd import fastgs
    d = fg.CASES[name]
    N, W, H = d["means"].shape[0], int(d["W"]), int(d["H"])
    s = fastgs.FastGSSettings(t(d["cam_pos"]), int(d["active_sh_bases"]), W, H, float(d["fx"]), float(d["fy"]), float(d["cx"]), float(d["cy"]), 0.01, 1e10)
    dev = [t(d[k]) for k in ("means", "scales_raw", "rot_raw", "opac_raw", "sh0", "sh_rest", "w2c")]
    image, alpha, pws, iws, n_inst = fastgs.forward_wrapper(*dev, s)
    fg.check_forward(d, n(image), n(alpha), n_inst)
    dens = t(d["densification_info_in"]) if "densification_info_in" in d else torch.zeros(0, device="cuda:0")
    g = fastgs.backward_wrapper(dens, t(d["g_image"]), t(d["g_alpha"]), image, alpha, *[t(d[k]) for k in ("means", "scales_raw", "rot_raw", "sh0", "sh_rest")],
                                pws, iws, t(d["w2c"]), s, n_inst)
    for (gk, _), got in zip(fg.GRADS, g):
        ref = d["out_" + gk]
        got = n(got).reshape(ref.shape)
        assert np.isfinite(got).all(), gk
        if np.abs(ref).max() == 0:
            assert np.abs(got).max() == 0, gk
            continue
        e, flips, rest = rows_check(got.reshape(N, -1), ref.reshape(N, -1), bar=5e-4, max_flips=3)
        print(f"fastgs vs reference {name} {gk}: rel-L2 {e:.2e}, flip rows {flips}, without them {rest:.2e}")
        assert rest < 5e-4, (name, gk, e, flips, rest)
    if "densification_info_in" in d:
        ref = d["out_densification_info"]
        assert np.abs(n(dens) - ref).max() < 2e-3 * np.abs(ref).max()


# ---- fused SSIM and the bilateral grid against the reference's own kernels (tests/golden/refk_loss.npz) ------------------------------------------
import test_oracle_refk_loss_golden as lg  # noqa: E402


@pytest.mark.parametrize("name", lg.SSIM)
def test_hip_fusedssim_matches_the_reference_kernel(lfs, name):
    """csrc/ssim.hip through losses.fusedssim / fusedssim_backward against fusedssimCUDA / fusedssim_backwardCUDA of the reference run on the CPU."""
    from gpu_util import n, t
    from lichtfeld_studio_amd import losses
    d = lg.CASES[name]
    a, b = t(d["img1"]), t(d["img2"])
    m, d1, d2, d3 = losses.fusedssim(lg.C1, lg.C2, a, b, True)
    lg.close(n(m), d["ssim_map"], 3e-6, "ssim_map")
    for got, key in zip((d1, d2, d3), ("dm_dmu1", "dm_dsigma1_sq", "dm_dsigma12")):
        lg.close(n(got), d[key], 3e-5, key)
    g = losses.fusedssim_backward(lg.C1, lg.C2, a, b, t(d["dL_dmap"]), d1, d2, d3)
    lg.close(n(g), d["dL_dimg1"], 3e-5, "dL_dimg1")
    # and with the reference's derivative maps handed in: the backward kernel alone
    g2 = losses.fusedssim_backward(lg.C1, lg.C2, a, b, t(d["dL_dmap"]), t(d["dm_dmu1"]), t(d["dm_dsigma1_sq"]), t(d["dm_dsigma12"]))
    lg.close(n(g2), d["dL_dimg1"], 1e-5, "dL_dimg1 from the reference's maps")


@pytest.mark.parametrize("name", lg.SLICE)
def test_hip_bilateral_slice_matches_the_reference_kernel(lfs, name):
    from gpu_util import n, t
    from lichtfeld_studio_amd import bilateral_grid as bg
    d = lg.CASES[name]
    lg.close(n(bg.slice_forward(t(d["grid"]), t(d["rgb"]))), d["output"], 3e-6, "slice output")
    gg, gr = bg.slice_backward(t(d["grid"]), t(d["rgb"]), t(d["grad_output"]))
    lg.close(n(gg), d["grad_grid"], 3e-5, "grad_grid")
    lg.close(n(gr), d["grad_rgb"], 3e-5, "grad_rgb")


@pytest.mark.parametrize("name", lg.TV)
def test_hip_bilateral_tv_matches_the_reference_kernel(lfs, name):
    from gpu_util import n, t
    from lichtfeld_studio_amd import bilateral_grid as bg
    d = lg.CASES[name]
    loss = float(bg.tv_loss_forward(t(d["grids"])).item())
    assert abs(loss - float(d["tv_loss"])) <= 1e-5 * float(d["tv_loss"])
    lg.close(n(bg.tv_loss_backward(t(d["grids"]), float(d["grad_output"]))), d["grad_grids"], 3e-6, "tv grad")


# ---- the neighbour distances behind the initial scales against the reference's own function (tests/golden/ref_splat_io.npz) ----------------------------------
@pytest.mark.parametrize("name", sorted({k.split("/")[1] for k in np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_splat_io.npz")).files
                                         if k.startswith("knn/")}))
def test_hip_mean_neighbor_distances_equal_the_reference_function(lfs, name):
    """loader.mean_neighbor_distances (host-built nanoflann tree, GPU walk: csrc/dataprep.hip) against compute_mean_neighbor_distances of the reference's
    splat_data.cpp run on the CPU: bit-identical, including the eps = 10 approximation the reference's query makes."""
    from gpu_util import n, t
    from lichtfeld_studio_amd import loader
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_splat_io.npz"))
    got = n(loader.mean_neighbor_distances(t(z[f"knn/{name}/points"])))
    assert np.array_equal(got, z[f"knn/{name}/mean_dist"]), (name, np.abs(got - z[f"knn/{name}/mean_dist"]).max())


@pytest.mark.parametrize("name", ["sfm_700_deg3_default_json", "dups_160_deg1_mcmc_json"])
def test_hip_init_model_from_pointcloud_equals_the_reference_function(lfs, name):
    """loader.init_model_from_pointcloud against SplatData::init_model_from_pointcloud of the reference's splat_data.cpp run on the CPU: means and the neighbour
    distances behind the scales are the same bits; log / sqrt / the colour conversion and the median distance differ by the GPU's libm (1e-6)."""
    from gpu_util import n
    from lichtfeld_studio_amd import loader
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_splat_io.npz"))
    g = lambda k: z[f"init/{name}/{k}"]
    deg, init_scaling, init_opacity = g("config")
    model, scene_scale = loader.init_model_from_pointcloud(loader.PointCloud(g("positions"), g("colors")), g("scene_center"), int(deg), float(init_scaling),
                                                          float(init_opacity))
    N = g("positions").shape[0]
    assert np.array_equal(n(model.means), g("out_means"))
    np.testing.assert_allclose(n(model.sh0), g("out_sh0"), rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(n(model.raw_scales), g("out_scaling"), rtol=2e-6, atol=2e-6)
    assert abs(scene_scale - float(g("out_scene_scale"))) <= 2e-6 * float(g("out_scene_scale"))
    assert model.shN.shape == (N,) + g("out_shN").shape[1:] and float(model.shN.detach().abs().max()) == 0 and float(np.abs(g("out_shN")).max()) == 0
    assert np.array_equal(n(model.raw_quats), np.tile(g("out_rotation")[0], (N, 1))) and list(g("out_rotation")[0]) == [1, 0, 0, 0]
    np.testing.assert_allclose(n(model.raw_opacities), np.full(N, g("out_opacity")[0, 0]), rtol=1e-6, atol=1e-7)
    assert model.active_sh_degree == 0 and model.max_sh_degree == int(deg)


# ---- the loss-side host code: fused_ssim wrapper / photometric loss / BilateralGrid against the reference's own host code (refk_loss.npz "host/...") ----------
@pytest.mark.parametrize("name", ["photometric_40x52", "photometric_small_9x40"])
def test_hip_photometric_loss_matches_the_reference_host_code(lfs, name):
    """losses.photometric_loss (autograd) and the fused losses.photometric_loss_fwd_bwd against Trainer::compute_photometric_loss over fused_ssim.cuh, incl. the
    wrapper's behaviour for images of <= 10 rows (whole map in the forward, no SSIM gradient in the backward)."""
    from gpu_util import n, t
    from lichtfeld_studio_amd import losses
    h = lg.host
    img, gt = t(h(f"{name}/rendered")), t(h(f"{name}/gt"))
    x = img.clone().requires_grad_(True)
    loss = losses.photometric_loss(x, gt, 0.2)
    loss.backward()
    assert abs(float(loss.detach()) - float(h(f"{name}/loss"))) <= 2e-6
    lg.close(n(x.grad), h(f"{name}/grad"), 3e-5, "d loss / d rendered")
    for pad in ("valid", "same"):
        y = img.clone().requires_grad_(True)
        v = losses.fused_ssim(y, gt, pad, True)
        v.backward()
        assert abs(float(v.detach()) - float(h(f"{name}/ssim_{pad}"))) <= 2e-6
        ref = h(f"{name}/ssim_{pad}_grad")
        if np.abs(ref).max() == 0:
            assert float(y.grad.abs().max()) == 0                # the <= 10 rows case of fused_ssim.cuh:92-97
        else:
            lg.close(n(y.grad), ref, 3e-5, f"ssim {pad} gradient")
    acc = torch.zeros(1, device="cuda:0")
    v = losses.photometric_loss_fwd_bwd(img.permute(1, 2, 0).contiguous(), gt, 0.2, 1.0, acc)      # the trainer's fused form, HWC render
    assert abs(float(acc) - float(h(f"{name}/loss"))) <= 2e-6
    lg.close(n(v.permute(2, 0, 1)), h(f"{name}/grad"), 3e-5, "fused d loss / d rendered")


def test_hip_bilateral_grid_component_matches_the_reference_host_code(lfs):
    """bilateral_grid.BilateralGrid (identity initialisation, apply(): clamp / layout / slice, tv_loss) through autograd, and the trainer's fused no-autograd entry
    points, against gs::training::BilateralGrid of the reference."""
    from gpu_util import n, t
    from lichtfeld_studio_amd import bilateral_grid as bgm
    h = lg.host
    delta, rgb, v_out, w_tv = h("bilateral/delta"), h("bilateral/rgb"), h("bilateral/v_out"), float(h("bilateral/tv_weight"))
    bg = bgm.BilateralGrid(3, 6, 5, 4)
    assert np.array_equal(n(bg.parameters()), h("bilateral/identity"))
    with torch.no_grad():
        bg.parameters().add_(t(delta))
    x = t(rgb).requires_grad_(True)
    out = bg.apply(x, 1)
    tv = bg.tv_loss()
    ((out * t(v_out)).sum() + w_tv * tv).backward()
    lg.close(n(out), h("bilateral/out"), 3e-6, "apply")
    assert abs(float(tv.detach()) - float(h("bilateral/tv"))) <= 1e-5 * float(h("bilateral/tv"))
    lg.close(n(bg.parameters().grad), h("bilateral/g_grids"), 3e-5, "d / d grids")
    lg.close(n(x.grad), h("bilateral/g_rgb"), 3e-5, "d / d rgb")
    # the fused path of the trainer: same numbers without autograd (CHW image, clamp inside the kernels, gradients accumulated in place)
    bg2 = bgm.BilateralGrid(3, 6, 5, 4)
    with torch.no_grad():
        bg2.parameters().add_(t(delta))
    img = t(rgb)
    out2 = bg2.apply_fused(img, 1, chw=True)
    lg.close(n(out2), h("bilateral/out"), 3e-6, "apply_fused")
    g_img = bg2.apply_fused_backward(img, 1, t(v_out), chw=True)
    acc = torch.zeros(1, device="cuda:0")
    bg2.tv_loss_fused(w_tv, acc)
    lg.close(n(bg2.parameters().grad), h("bilateral/g_grids"), 3e-5, "fused d / d grids")
    lg.close(n(g_img), h("bilateral/g_rgb"), 3e-5, "fused d / d rgb")
    assert abs(float(acc) - w_tv * float(h("bilateral/tv"))) <= 1e-5 * w_tv * float(h("bilateral/tv"))


# ---- K2 / K9 and K3 - K6 against the reference's own kernels (tests/golden/refk_sh_isect.npz) -----------------------------------------------------------------
import refk_sh_isect_util as shi  # noqa: E402
import test_oracle_refk_sh_isect_golden as shg  # noqa: E402


@pytest.mark.parametrize("name", sorted(shi.SH_CASES))
def test_hip_sh_matches_the_reference_kernels(lfs, name):
    from gpu_util import n, t
    from lichtfeld_studio_amd import ops
    c, g = shi.SH_CASES[name], lambda k: shg.GOLD[f"sh/{name}/{k}"]
    dirs, coeffs, masks, v = shi.sh_inputs(c)
    d, co, m = t(dirs), t(coeffs), t(masks, torch.bool)
    col = n(ops.spherical_harmonics_fwd(c["degree"], d, co, m))
    shg.close(np.where(masks[:, None], col, 0), g("colors"), 3e-6, "colors")
    shg.close(n(ops.spherical_harmonics_fwd(c["degree"], d, co, None)), g("colors_unmasked"), 3e-6, "colors without mask")
    v_coeffs, v_dirs = ops.spherical_harmonics_bwd(coeffs.shape[1], c["degree"], d, co, m, t(v), True)
    shg.close(n(v_coeffs), g("v_coeffs"), 3e-6, "v_coeffs")
    shg.close(n(v_dirs), g("v_dirs"), 1e-5, "v_dirs")


@pytest.mark.parametrize("name", sorted(shi.ISECT_CASES))
def test_hip_intersection_matches_the_reference_kernels(lfs, name):
    from gpu_util import n, t
    from lichtfeld_studio_amd import ops
    c, g = shi.ISECT_CASES[name], lambda k: shg.GOLD[f"isect/{name}/{k}"]
    m, r, d = shi.isect_inputs(c)
    tw, th = (c["W"] + c["tile"] - 1) // c["tile"], (c["H"] + c["tile"] - 1) // c["tile"]
    tpg, ids, flat = ops.intersect_tile(t(m), t(r, torch.int32), t(d), None, None, c["C"], c["tile"], tw, th, True)
    assert np.array_equal(n(tpg), g("tiles_per_gauss")) and np.array_equal(n(ids), g("isect_ids")) and np.array_equal(n(flat), g("flatten_ids"))
    _, ids_u, flat_u = ops.intersect_tile(t(m), t(r, torch.int32), t(d), None, None, c["C"], c["tile"], tw, th, False)
    assert np.array_equal(n(ids_u), g("isect_ids_unsorted")) and np.array_equal(n(flat_u), g("flatten_ids_unsorted"))
    assert np.array_equal(n(ops.intersect_offset(ids, c["C"], tw, th)), g("offsets"))
