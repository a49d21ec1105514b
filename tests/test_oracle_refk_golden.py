"""CPU: the oracle restatement (oracle/oracle_ops.hpp) against golden vectors generated from the REFERENCE'S OWN device kernels
(tests/golden/refk_*.npz; oracle/make_golden_refk.py runs gsplat/ProjectionUT3DGSFused.cu, RasterizeToPixelsFromWorld3DGS{Fwd,Bwd}.cu,
RelocationCUDA.cu, QuatToRotmatCUDA.cu and fastgs' adam_kernels.cuh, compiled in place as host code under oracle/ref_emul/). This is the pin
SURVEY.md §8c says the reference's tests do not provide for these ops. The same files check the HIP kernels in tests/test_gpu_refk_golden.py.

Measured when the files were generated (IEEE fp32 on both sides): projection - visibility sets identical, radii identical up to one +-1 in
10 cases, means2d <= 4e-3 px (the unscented transform's -99 / +16.67 weights), conics <= 2e-3 relative; forward image bit-identical in 7 of 11
cases and <= 6e-6 elsewhere, last_ids identical; backward 1e-7 .. 1e-5 relative L2; relocation / add_noise / quats_to_rotmats <= 1 ulp;
Adam bit-identical over 5 chained steps."""
import os

import numpy as np
import pytest

import refk_util as ru

PROJ = ru.projection_cases()


def _proj(oracle, d, fn=None):
    fn = fn or oracle.projection_ut_3dgs_fused
    return fn(d["means"], d["quats"], d["scales"], None if d["no_opacity"] else d["opacities"], d["viewmats0"], d["viewmats1"], d["Ks"], d["W"], d["H"],
              eps2d=d["eps2d"], radius_clip=d["radius_clip"], calc_compensations=d["calc_compensations"], camera_model=d["camera_model"],
              ut_params=d["ut_params"], rs_type=d["rs_type"], radial_coeffs=d["radial"], tangential_coeffs=d["tangential"], thin_prism_coeffs=d["thin_prism"])


def check_projection(d, radii, m2, depths, conics, comp):
    """shared with the GPU test: tolerance model of SURVEY.md §8c for K1"""
    vis_ref, vis = (d["radii"] > 0).all(-1), (radii > 0).all(-1)
    assert (vis_ref != vis).mean() < 2e-3, int((vis_ref != vis).sum())          # measured: 0
    both = vis_ref & vis
    dr = np.abs(radii - d["radii"])[both]
    assert dr.max() <= 1 and (dr > 0).any(-1).mean() < 2e-3, (int(dr.max()), float((dr > 0).any(-1).mean()))
    assert np.abs(m2 - d["means2d"])[both].max() < 1e-2                          # px; fp32 noise floor of the UT, measured <= 4e-3
    assert np.abs(depths - d["depths"])[both].max() < 1e-5 * max(1.0, float(np.abs(d["depths"]).max()))
    scale = np.abs(d["conics"])[both].max(-1, keepdims=True)
    assert (np.abs(conics - d["conics"])[both] / scale).max() < 1e-2              # measured <= 2e-3 (rolling shutter), <= 7e-5 otherwise
    if d["compensations"] is not None:
        assert np.abs(comp - d["compensations"])[both].max() < 1e-5


@pytest.mark.parametrize("name", sorted(PROJ))
def test_oracle_projection_matches_reference_kernel(oracle_mod, name):
    d = PROJ[name]
    check_projection(d, *_proj(oracle_mod, d))


def check_raster_fwd(d, rc, ra, li, mean_bar=1e-6):
    e = np.abs(rc - d["render"])
    assert e.mean() < mean_bar and (e.max(-1) > 1 / 255 + 1e-4).mean() < 1e-3, (e.mean(), e.max())
    assert np.abs(ra - d["alpha"]).mean() < mean_bar
    assert (li == d["last_ids"]).mean() > 0.999


def raster_bwd_rows(d, g):
    """[(name, ours as rows, reference rows)]: one row per Gaussian (per (camera, Gaussian) for colours / opacities)"""
    names = ["v_means", "v_quats", "v_scales", "v_colors", "v_opacities"]
    out = []
    for nme, a in zip(names, g):
        b = d[nme]
        a = np.asarray(a)
        out.append((nme, a.reshape(-1, 1) if nme == "v_opacities" else a.reshape(-1, a.shape[-1]), b.reshape(-1, 1) if nme == "v_opacities" else b.reshape(-1, b.shape[-1])))
    return out


@pytest.mark.parametrize("name", ru.RASTER_CASES)
def test_oracle_rasterization_matches_reference_kernel(oracle_mod, name):
    d = ru.raster_case(name)
    args = ru.oracle_raster_args(d)
    rc, ra, li = oracle_mod.rasterize_fwd(*args)
    assert np.abs(rc - d["render"]).max() < 1e-4 and (li == d["last_ids"]).mean() > 0.9999    # measured: <= 6e-6, identical
    check_raster_fwd(d, rc, ra, li)
    g = oracle_mod.rasterize_bwd(*args, d["alpha"], d["last_ids"], d["v_render"], d["v_alpha"])
    for nme, a, b in raster_bwd_rows(d, g):
        e = float(np.linalg.norm(a.astype(np.float64) - b) / (np.linalg.norm(b.astype(np.float64)) + 1e-30))
        assert e < 1e-4, (nme, e)                                                             # measured: <= 1e-5


def test_oracle_small_ops_match_reference_kernels(oracle_mod):
    s = ru.small_ops()
    no, ns = oracle_mod.relocation(s["reloc_opacities"], s["reloc_scales"], s["reloc_ratios"], s["binoms"], 51)
    assert np.abs(no - s["reloc_new_opacities"]).max() < 1e-6
    assert (np.abs(ns - s["reloc_new_scales"]) / np.abs(s["reloc_new_scales"])).max() < 1e-5
    m = oracle_mod.add_noise(s["noise_raw_opacities"], s["noise_raw_scales"], s["noise_raw_quats"], s["noise_noise"], s["noise_means"], float(s["noise_lr"]))
    moved = np.abs(s["noise_means_out"] - s["noise_means"]).max()
    assert moved > 1e-3 and np.abs(m - s["noise_means_out"]).max() < 1e-6 * max(1.0, moved)
    assert np.abs(oracle_mod.quats_to_rotmats(s["quats"]) - s["rotmats"]).max() < 1e-6
    lr, b1, b2, eps = s["adam_hyper"]
    p, mm, v = s["adam_p0"], np.zeros_like(s["adam_p0"]), np.zeros_like(s["adam_p0"])
    for t in range(1, 6):
        bc1, bc2 = np.float32(1.0 / (1.0 - b1 ** t)), np.float32(1.0 / np.sqrt(1.0 - b2 ** t))
        p, mm, v = oracle_mod.adam_step(p, mm, v, s["adam_grads"][t - 1], lr, b1, b2, eps, bc1, bc2)
        assert np.array_equal(p, s[f"p{t}"]) and np.array_equal(mm, s[f"m{t}"]) and np.array_equal(v, s[f"v{t}"]), t


def test_golden_files_reproduce_from_the_reference_kernels(oracle_mod):
    """Only where /root/reference exists (the build container): the committed files are what the reference's kernels produce today."""
    if not os.path.exists("/root/reference/gsplat/ProjectionUT3DGSFused.cu"):
        pytest.skip("reference tree not present")
    oracle_mod.build(ref=True)
    assert oracle_mod.refk_lib() is not None
    d = PROJ["opencv_distortion"]
    out = _proj(oracle_mod, d, oracle_mod.refk_projection_ut)
    assert np.array_equal(out[0], d["radii"]) and np.array_equal(out[1], d["means2d"]) and np.array_equal(out[3], d["conics"])
    d = ru.raster_case("rolling_top_bottom")
    args = ru.oracle_raster_args(d)
    rc, ra, li = oracle_mod.refk_rasterize_fwd(*args)
    assert np.array_equal(rc, d["render"]) and np.array_equal(li, d["last_ids"])
    g = oracle_mod.refk_rasterize_bwd(*args, d["alpha"], d["last_ids"], d["v_render"], d["v_alpha"])
    assert np.array_equal(g[0], d["v_means"]) and np.array_equal(g[1], d["v_quats"]) and np.array_equal(g[4], d["v_opacities"])
    s = ru.small_ops()
    p, m, v = oracle_mod.refk_adam_step(s["adam_p0"], np.zeros_like(s["adam_p0"]), np.zeros_like(s["adam_p0"]), s["adam_grads"][0],
                                        *s["adam_hyper"], np.float32(1.0 / (1.0 - s["adam_hyper"][1])), np.float32(1.0 / np.sqrt(1.0 - s["adam_hyper"][2])))
    assert np.array_equal(p, s["p1"])


@pytest.mark.skipif(not __import__("oracle").have_ref("libref_raster_full.so"), reason="oracle/_ref/libref_raster_full.so not built (make -C oracle refgsplat; needs /root/reference)")
def test_every_golden_case_regenerates_through_the_references_own_operators(oracle_mod):
    """The committed refk_*.npz files were generated through restated launch sequences (ref_kernels.cpp: allocation + <<<grid, block>>> geometry per kernel). Here the
    SAME inputs go through the reference's own operator layer - gsplat/*.cpp calling the launch functions of gsplat/*.cu, all compiled in place
    (ref_gsplat_all.cpp + ref_gsplat_capi.cpp) - and every output must equal the file bit for bit: all 10 projection cases (every camera model and shutter), all 11
    rasterization cases forward + backward, relocation / add_noise / quat->R, and the SH / intersection file."""
    import refk_sh_isect_util as shi
    o = oracle_mod
    o.REFK_WHOLE_LIBRARY = True
    try:
        assert o.refk_lib() is o.ref_raster_lib(full=True)
        for name, d in PROJ.items():
            radii, m2, depths, conics, comp = _proj(o, d, o.refk_projection_ut)
            for got, key in ((radii, "radii"), (m2, "means2d"), (depths, "depths"), (conics, "conics")):
                vis = (d["radii"] > 0).all(-1)                                      # (rows of culled Gaussians are at::empty in the operator)
                assert np.array_equal(got[vis], d[key][vis]), (name, key)
            assert np.array_equal(radii, d["radii"]), name
            if d["compensations"] is not None:
                assert np.array_equal(comp[(d["radii"] > 0).all(-1)], d["compensations"][(d["radii"] > 0).all(-1)]), name
        for name in ru.RASTER_CASES:
            d = ru.raster_case(name)
            args = ru.oracle_raster_args(d)
            rc, ra, li = o.refk_rasterize_fwd(*args)
            px = np.ones(d["last_ids"].shape, bool)
            if d["masks"] is not None:       # a masked-out tile gets its colours (the background) but neither alpha nor last_ids: at::empty in the operator, 0 in the file
                px = np.kron(np.asarray(d["masks"], bool), np.ones((int(d["tile"]), int(d["tile"])), bool))[:, :int(d["H"]), :int(d["W"])]
            assert np.array_equal(rc, d["render"]) and np.array_equal(ra.reshape(d["alpha"].shape)[px], d["alpha"][px]) and np.array_equal(li[px], d["last_ids"][px]), name
            g = o.refk_rasterize_bwd(*args, d["alpha"], d["last_ids"], d["v_render"], d["v_alpha"])
            for got, key in zip(g, ("v_means", "v_quats", "v_scales", "v_colors", "v_opacities")):
                assert np.array_equal(got.reshape(d[key].shape), d[key]), (name, key)
        s = ru.small_ops()
        no, ns = o.refk_relocation(s["reloc_opacities"], s["reloc_scales"], s["reloc_ratios"], s["binoms"], 51)
        assert np.array_equal(no, s["reloc_new_opacities"]) and np.array_equal(ns, s["reloc_new_scales"], equal_nan=True)
        assert np.array_equal(o.refk_add_noise(s["noise_raw_opacities"], s["noise_raw_scales"], s["noise_raw_quats"], s["noise_noise"], s["noise_means"], float(s["noise_lr"])),
                              s["noise_means_out"])
        assert np.array_equal(o.refk_quats_to_rotmats(s["quats"]), s["rotmats"])
        G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", shi.GOLD))
        from oracle import make_golden_refk_sh_isect as mg
        for name, c in shi.SH_CASES.items():
            for k, v in mg.run_sh(c).items():
                assert np.array_equal(v, G[f"sh/{name}/{k}"], equal_nan=True), (name, k)
        for name, c in shi.ISECT_CASES.items():
            for k, v in mg.run_isect(c).items():
                assert np.array_equal(v, G[f"isect/{name}/{k}"]), (name, k)
    finally:
        o.REFK_WHOLE_LIBRARY = False
