"""CPU: the ORACLE's composition of the hot path (oracle/liboracle.so operators chained as tests/test_gpu_pipeline.py::_oracle_step chains them - the checker behind
the full-size parity tests) against the REFERENCE'S OWN rasterize() + autograd + Camera run on the CPU (tests/golden/ref_raster.npz,
oracle/make_golden_ref_raster.py). Same bars as the GPU test of the product (tests/test_gpu_raster_reference.py): radii exact, image / alpha 2e-5, gradients 1e-4
relative L2. Where oracle/_ref/libref_raster_full.so exists (the build container) the file also regenerates bit for bit from the whole reference library."""
import os

import numpy as np
import pytest

import oracle
import refraster_util as U

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", U.GOLD))


def oracle_render_backward(o, c, s):
    means, raw_q, raw_s, raw_o = s["means"], s["rotation"], s["scaling"], s["opacity"]
    qn = np.linalg.norm(raw_q, axis=-1, keepdims=True)
    quats, scales, opac = (raw_q / qn).astype(np.float32), np.exp(raw_s), (1 / (1 + np.exp(-raw_o))).astype(np.float32)
    sh = np.concatenate([s["sh0"], s["shN"]], 1)
    W, H, deg = c["W"], c["H"], c["active"]
    vm = np.eye(4, dtype=np.float32)
    vm[:3, :3], vm[:3, 3] = s["R"], s["T"]
    K = np.array([[c["focal"], 0, W / 2 + 0.5], [0, c["focal"] * 1.05, H / 2 - 0.25], [0, 0, 1]], np.float32)
    vm, K = vm[None], K[None]
    radii, m2, d, _, _ = o.projection_ut_3dgs_fused(means, quats, scales, opac, vm, None, K, W, H)
    mask = (radii[0] > 0).all(-1)
    dirs = means - np.linalg.inv(vm[0].astype(np.float64))[:3, 3].astype(np.float32)
    col = o.spherical_harmonics_fwd(deg, dirs, sh, mask)
    colors = np.maximum(col + 0.5, 0)[None]
    tw, th = (W + 15) // 16, (H + 15) // 16
    _, ids, flat = o.intersect_tile(m2, radii, d, 1, 16, tw, th, True)
    offs = o.intersect_offset(ids, 1, tw, th)
    bg = None if c["bg"] is None else np.array([c["bg"]], np.float32)
    rc, ra, li = o.rasterize_fwd(means, quats, scales, colors, opac[None], bg, None, W, H, 16, vm, None, K, 0, 4, None, None, None, offs, flat)
    img_raw = rc[0].transpose(2, 0, 1)
    v_img = s["v_image"] * ((img_raw >= 0) & (img_raw <= 1))                       # the final clamp(0, 1) of rasterizer.cpp:399
    v_rc = np.ascontiguousarray(v_img.transpose(1, 2, 0))[None].astype(np.float32)
    v_ra = np.zeros_like(ra) if s["v_alpha"] is None else np.ascontiguousarray(s["v_alpha"].reshape(1, H, W, 1))
    gm, gq, gs, gc, go = o.rasterize_bwd(means, quats, scales, colors, opac[None], bg, None, W, H, 16, vm, None, K, 0, 4, None, None, None, offs, flat, ra, li, v_rc, v_ra)
    g_col = np.where(col + 0.5 > 0, gc[0], 0).astype(np.float32)
    g_sh, g_dirs = o.spherical_harmonics_bwd(deg, dirs, sh, mask, g_col, True)
    return dict(image=np.clip(img_raw, 0, 1), alpha=ra[0].transpose(2, 0, 1), radii=radii[0].max(-1), g_means=gm + g_dirs, g_sh0=g_sh[:, :1], g_shN=g_sh[:, 1:],
                g_scaling=gs * scales, g_rotation=(gq - (gq * quats).sum(-1, keepdims=True) * quats) / qn, g_opacity=go[0] * opac * (1 - opac))


@pytest.mark.parametrize("name", sorted(U.CASES))
def test_oracle_composition_equals_the_reference_render_path(oracle_mod, name):
    c = U.CASES[name]
    r = oracle_render_backward(oracle_mod, c, U.scene(c))
    g = lambda k: GOLD[f"{name}/{k}"]
    assert np.array_equal(r["radii"], g("radii"))
    assert np.abs(r["image"] - g("image")).max() <= 2e-5 and np.abs(r["alpha"] - g("alpha")).max() <= 2e-5
    for k in ("g_means", "g_sh0", "g_shN", "g_scaling", "g_rotation", "g_opacity"):
        ref, got = g(k).astype(np.float64), np.asarray(r[k], np.float64).reshape(g(k).shape)
        if np.abs(ref).max() == 0:
            assert np.abs(got).max() == 0, k
            continue
        e = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        assert e < 1e-4, (k, e)


@pytest.mark.parametrize("name", sorted(U.CASES))
def test_trainer_loss_restatement_equals_the_reference_train_step(oracle_mod, name):
    """"train/..." entries: the loss Trainer::train_step composes (rasterize -> L1 + 0.2 D-SSIM "valid" -> + 0.01 mean(scale) + 0.01 mean(opacity)) and its
    gradients, from the reference's own code. Here: the oracle's render + the torch SSIM restatement (tests/ssim_reference.py) + the regularisers' closed-form
    gradients (scale_reg / 3N * exp(s), opacity_reg / N * o (1 - o)) - the composition the product's fused step implements."""
    import torch
    import ssim_reference as ssim
    from oracle import make_golden_ref_raster as mg
    c = U.CASES[name]
    s = U.scene(c)
    g = lambda k: GOLD[f"{name}/train/{k}"]
    gt = torch.from_numpy(U.target_image(c)).double()
    base = oracle_render_backward(oracle_mod, c, dict(s, v_image=np.zeros_like(s["v_image"]), v_alpha=None))       # forward only: the clamped image
    x = torch.from_numpy(base["image"]).double().requires_grad_(True)
    photo = ssim.photometric_loss(x[None], gt[None], mg.TRAIN["lambda_dssim"])
    photo.backward()
    r = oracle_render_backward(oracle_mod, c, dict(s, v_image=x.grad.numpy().astype(np.float32), v_alpha=None))   # backward with dL/dimage of the photometric loss
    N = c["N"]
    scales, opac = np.exp(s["scaling"].astype(np.float64)), 1 / (1 + np.exp(-s["opacity"].astype(np.float64)))
    loss = float(photo.detach()) + mg.TRAIN["scale_reg"] * scales.mean() + mg.TRAIN["opacity_reg"] * opac.mean()
    assert abs(loss - float(g("loss"))) <= 2e-6 * float(g("loss"))
    r["g_scaling"] = r["g_scaling"] + mg.TRAIN["scale_reg"] / (3 * N) * scales
    r["g_opacity"] = r["g_opacity"] + mg.TRAIN["opacity_reg"] / N * opac * (1 - opac)
    for k in ("g_means", "g_sh0", "g_shN", "g_scaling", "g_rotation", "g_opacity"):
        ref, got = g(k).astype(np.float64), np.asarray(r[k], np.float64).reshape(g(k).shape)
        if np.abs(ref).max() == 0:
            assert np.abs(got).max() == 0, k
            continue
        e = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        assert e < 1e-4, (k, e)


@pytest.mark.skipif(not oracle.have_ref("libref_raster_full.so"), reason="oracle/_ref/libref_raster_full.so not built (make -C oracle refgsplat; needs /root/reference)")
def test_golden_file_regenerates_from_the_whole_reference_gsplat_library():
    """rasterize() + autograd + Camera over the reference's entire gsplat library (every .cu with its launchers, every .cpp operator) on the CPU: bit for bit"""
    from oracle import make_golden_ref_raster as mg
    for name, c in U.CASES.items():
        r = mg.run(c, full=True)
        for k, v in r.items():
            assert np.array_equal(v, GOLD[f"{name}/{k}"]), (name, k)
        for k, v in mg.run_train(c).items():
            if k != "image":
                assert np.array_equal(v, GOLD[f"{name}/train/{k}"]), (name, "train", k)
        if f"{name}/train_bilateral/loss" in GOLD.files:
            for k, v in mg.run_train(c, bilateral=True).items():
                if k != "image":
                    close = np.array_equal(v, GOLD[f"{name}/train_bilateral/{k}"]) if k != "g_grids" else \
                        np.abs(v - GOLD[f"{name}/train_bilateral/{k}"]).max() <= 1e-6 * np.abs(v).max()        # (atomics into the grid: order-dependent last bits)
                    assert close, (name, "train_bilateral", k)


@pytest.mark.skipif(not oracle.have_ref("libref_raster.so"), reason="oracle/_ref/libref_raster.so not built (needs /root/reference)")
def test_restated_launch_sequences_reproduce_the_whole_library():
    """The same render path over the RESTATED host launch sequences (ref_raster_shim.cpp: kernels of libref_kernels.so, SH / intersection through the reference's
    tests/torch_impl.cpp) - the form the per-operator golden files of round 2 were generated with: integers identical, floats within 5e-6 of the tensor maximum
    (the SH polynomial is associated differently in torch_impl and in the kernel)."""
    from oracle import make_golden_ref_raster as mg
    for name, c in U.CASES.items():
        r = mg.run(c, full=False)
        for k, v in r.items():
            g = GOLD[f"{name}/{k}"]
            if v.dtype.kind in "iu":
                assert np.array_equal(v, g), (name, k)
            else:
                assert np.abs(v.astype(np.float64) - g).max() <= 5e-6 * max(np.abs(g).max(), 1e-30), (name, k)


@pytest.mark.skipif(not oracle.have_ref("libref_fast_raster.so"), reason="oracle/_ref/libref_fast_raster.so not built (make -C oracle reffast; needs /root/reference)")
def test_fastgs_golden_files_regenerate_from_the_whole_reference_fastgs_path():
    """(1) tests/golden/ref_fast_raster.npz from fast_rasterize() + autograd + rasterization_api.cu + forward.cu / backward.cu, bit for bit; (2) the wrapper-level file
    refk_fastgs.npz - generated through the restated wrapper of ref_kernels_fastgs.cpp - through the reference's own forward_wrapper / backward_wrapper, bit for bit."""
    from oracle import make_golden_ref_fast_raster as mg
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", mg.GOLD))
    for name in mg.CASES:
        for k, v in mg.run(name).items():
            assert np.array_equal(v, G[f"{name}/{k}"]), (name, k)
        for k, v in mg.run_train(name).items():
            assert np.array_equal(v, G[f"{name}/train/{k}"]), (name, "train", k)
    import test_oracle_refk_fastgs_golden as fg
    oracle.REFK_FASTGS_WHOLE = True
    try:
        for name, d in fg.CASES.items():
            r = oracle.refk_fastgs_fwd_bwd(*fg.scene_args(d), d["g_image"], d["g_alpha"], densification_info=d.get("densification_info_in"))
            for k, v in r.items():
                assert np.array_equal(np.asarray(v).reshape(d["out_" + k].shape), d["out_" + k]), (name, k)
    finally:
        oracle.REFK_FASTGS_WHOLE = False
