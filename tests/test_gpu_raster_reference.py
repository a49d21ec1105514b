"""GPU: the product's training-time render path - rasterizer.rasterize() (the mirror of gs::training::rasterize) and its backward - against the REFERENCE'S OWN
rasterize() + autograd Functions + Camera + SplatData activations run end to end on the CPU (tests/golden/ref_raster.npz, oracle/make_golden_ref_raster.py:
rasterizer.cpp, rasterizer_autograd.cpp, camera.cpp compiled in place against libtorch over the reference's kernels). This pins the COMPOSITION of SURVEY.md §8
rows a1-a6 - which tensors are activated how, the SH direction and mask, the +0.5 / clamp_min on colours, the background, the final clamp, the path of dL/dmeans
through the SH directions - to the reference itself; the operators are pinned one by one in test_gpu_refk_golden.py.
Bars: camera matrices exact; radii exact; image / alpha max-abs 2e-5 (fast exp on the GPU, K7's own tolerance in SURVEY §8c); gradients relative L2 <= 1e-4 per
tensor with threshold-flip rows counted (gpu_util.rows_check; measured: 3e-6 .. 2e-5, no flip row)."""
import os

import numpy as np
import pytest
import torch

import refraster_util as U
from gpu_util import n, rows_check, t

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", U.GOLD))


@pytest.mark.parametrize("name", sorted(U.CASES))
def test_render_and_backward_follow_the_reference_render_path(lfs, name):
    from lichtfeld_studio_amd import loader
    from lichtfeld_studio_amd.rasterizer import Camera, SplatModel, rasterize
    c, g = U.CASES[name], lambda k: GOLD[f"{name}/{k}"]
    s = U.scene(c)
    W, H = c["W"], c["H"]
    # the camera as the loader builds it from COLMAP quantities (loader.world_to_view / intrinsics mirror camera.cpp:15-23, 77-98)
    cam = loader.CameraData(0, 1, 0, W, H, np.float32(c["focal"]), np.float32(c["focal"] * 1.05), np.float32(W / 2 + 0.5), np.float32(H / 2 - 0.25), s["R"], s["T"],
                            np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32), "view", "")
    viewmat, K = loader.world_to_view(cam), loader.intrinsics(cam, W, H)
    assert np.array_equal(viewmat, g("viewmat")) and np.array_equal(K, g("K"))
    mk = lambda a: t(a).contiguous().requires_grad_(True)
    model = SplatModel(mk(s["means"]), mk(s["sh0"]), mk(s["shN"]), mk(s["scaling"]), mk(s["rotation"]), mk(s["opacity"]), c["sh_degree"], active_sh_degree=c["active"])
    out = rasterize(Camera(t(viewmat).unsqueeze(0), t(K).unsqueeze(0), W, H), model, None if c["bg"] is None else t(np.array(c["bg"], np.float32)))
    image, alpha = n(out.image), n(out.alpha).reshape(1, H, W)
    assert image.shape == (3, H, W)
    radii = n(out.radii).reshape(-1)
    assert np.array_equal(radii, g("radii")), int((radii != g("radii")).sum())
    e_img, e_alpha = np.abs(image - g("image")).max(), np.abs(alpha - g("alpha")).max()
    print(f"{name}: image max-abs {e_img:.2e}, alpha max-abs {e_alpha:.2e}")
    assert e_img <= 2e-5 and e_alpha <= 2e-5
    loss = (out.image * t(s["v_image"])).sum()
    if s["v_alpha"] is not None:
        loss = loss + (out.alpha.reshape(1, H, W) * t(s["v_alpha"])).sum()
    loss.backward()
    for key, p in zip(("g_means", "g_sh0", "g_shN", "g_scaling", "g_rotation", "g_opacity"), model.parameters()):
        ref = g(key)
        got = n(p.grad).reshape(ref.shape) if p.grad is not None else np.zeros_like(ref)
        assert np.isfinite(got).all(), key
        if np.abs(ref).max() == 0:
            assert np.abs(got).max() == 0, key                      # e.g. shN at active degree 0: no gradient at all
            continue
        e, flips, rest = rows_check(got.reshape(c["N"], -1), ref.reshape(c["N"], -1), bar=1e-4, max_flips=3)
        print(f"{name} {key}: rel-L2 {e:.2e}, flip rows {flips}, without them {rest:.2e}")
        assert rest < 1e-4, (key, e, flips, rest)
        assert np.array_equal(np.abs(got).reshape(c["N"], -1).max(1) > 0, np.abs(ref).reshape(c["N"], -1).max(1) > 0) or key in ("g_means",), key


FAST = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fast_raster.npz"))


FAST_CASES = sorted({k.split("/")[0] for k in FAST.files})


@pytest.mark.parametrize("name", FAST_CASES)
def test_fastgs_render_and_backward_follow_the_reference_render_path(lfs, name):
    """fastgs.fast_rasterize (the mirror of gs::training::fast_rasterize, the reference's default training path) + autograd backward against the reference's own
    fast_rasterizer.cpp + fast_rasterizer_autograd.cpp + rasterization_api.cu + forward.cu / backward.cu + Camera run end to end on the CPU
    (tests/golden/ref_fast_raster.npz): image / alpha 5e-6 max-abs, gradients 2e-5 relative L2 with threshold-flip rows counted (measured: 2e-7 and 3e-7 .. 5e-7, no flip row), densification_info accumulated alike."""
    from lichtfeld_studio_amd import fastgs, loader
    from lichtfeld_studio_amd.rasterizer import Camera, SplatModel
    c, g = U.CASES[name], lambda k: FAST[f"{name}/{k}"]
    s = U.scene(c)
    W, H = c["W"], c["H"]
    cam = loader.CameraData(0, 1, 0, W, H, np.float32(c["focal"]), np.float32(c["focal"] * 1.05), np.float32(W / 2 + 0.5), np.float32(H / 2 - 0.25), s["R"], s["T"],
                            np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32), "view", "")
    mk = lambda a: t(a).contiguous().requires_grad_(True)
    model = SplatModel(mk(s["means"]), mk(s["sh0"]), mk(s["shN"]), mk(s["scaling"]), mk(s["rotation"]), mk(s["opacity"]), c["sh_degree"], active_sh_degree=c["active"])
    dens = t(g("densification_info_in")).clone()
    out = fastgs.fast_rasterize(Camera(t(loader.world_to_view(cam)).unsqueeze(0), t(loader.intrinsics(cam, W, H)).unsqueeze(0), W, H), model, t(g("bg")), dens)
    e_img, e_alpha = np.abs(n(out.image) - g("image")).max(), np.abs(n(out.alpha).reshape(1, H, W) - g("alpha")).max()
    print(f"fastgs {name}: image max-abs {e_img:.2e}, alpha max-abs {e_alpha:.2e}")
    assert e_img <= 5e-6 and e_alpha <= 5e-6
    ((out.image * t(s["v_image"])).sum() + (out.alpha.reshape(1, H, W) * t(g("v_alpha"))).sum()).backward()
    for key, p in zip(("g_means", "g_sh0", "g_shN", "g_scaling", "g_rotation", "g_opacity"), model.parameters()):
        ref = g(key)
        got = n(p.grad).reshape(ref.shape)
        assert np.isfinite(got).all(), key
        if np.abs(ref).max() == 0:
            assert np.abs(got).max() == 0, key
            continue
        e, flips, rest = rows_check(got.reshape(c["N"], -1), ref.reshape(c["N"], -1), bar=2e-5, max_flips=3)
        print(f"fastgs {name} {key}: rel-L2 {e:.2e}, flip rows {flips}, without them {rest:.2e}")
        assert rest < 2e-5, (key, e, flips, rest)
    ref = g("densification_info")
    assert np.abs(n(dens) - ref).max() <= 2e-3 * np.abs(ref).max()


@pytest.mark.parametrize("name", sorted(U.CASES))
def test_fused_training_step_follows_the_reference_train_step(lfs, name):
    """fused.render_and_backward - THE hot path of the trainer: forward, the fused L1 + D-SSIM loss kernels, backward with the regularisers folded into the activation
    backward - against the loss Trainer::train_step composes and its gradients, computed by the reference's own rasterize() + autograd + fused_ssim over its whole
    gsplat library and ssim.cu on the CPU ("train/..." entries of tests/golden/ref_raster.npz): loss 5e-6 relative, gradients 1e-4 relative L2 per tensor with
    threshold-flip rows counted."""
    from lichtfeld_studio_amd import fused, loader
    from lichtfeld_studio_amd.rasterizer import Camera, SplatModel
    c, g = U.CASES[name], lambda k: GOLD[f"{name}/train/{k}"]
    s = U.scene(c)
    W, H = c["W"], c["H"]
    cam = loader.CameraData(0, 1, 0, W, H, np.float32(c["focal"]), np.float32(c["focal"] * 1.05), np.float32(W / 2 + 0.5), np.float32(H / 2 - 0.25), s["R"], s["T"],
                            np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32), "view", "")
    mk = lambda a: t(a).contiguous().requires_grad_(True)
    model = SplatModel(mk(s["means"]), mk(s["sh0"]), mk(s["shN"]), mk(s["scaling"]), mk(s["rotation"]), mk(s["opacity"]), c["sh_degree"], active_sh_degree=c["active"])
    camera = Camera(t(loader.world_to_view(cam)).unsqueeze(0), t(loader.intrinsics(cam, W, H)).unsqueeze(0), W, H)
    grads = [torch.zeros_like(p) for p in model.parameters()]
    loss = torch.zeros(1, device="cuda:0")
    fused.render_and_backward(camera, model, None if c["bg"] is None else t(np.array(c["bg"], np.float32)), t(U.target_image(c)), 1.0, grads, loss, accumulate=False,
                              loss="l1_ssim", lambda_dssim=0.2, scale_reg=0.01, opacity_reg=0.01)
    torch.cuda.synchronize()
    # the fused step reports the photometric part of the loss only: the regularisers enter through their gradients (folded into the activation backward), their
    # VALUES - which the reference adds to the number it logs, trainer.cpp:686-697 - are not computed on the hot path
    reg = 0.01 * np.exp(s["scaling"].astype(np.float64)).mean() + 0.01 * (1 / (1 + np.exp(-s["opacity"].astype(np.float64)))).mean()
    print(f"train step {name}: loss {float(loss):.6f} + regularisers {reg:.6f} vs {float(g('loss')):.6f}")
    assert abs(float(loss) + reg - float(g("loss"))) <= 5e-6 * float(g("loss"))
    for key, got in zip(("g_means", "g_sh0", "g_shN", "g_scaling", "g_rotation", "g_opacity"), grads):
        ref = g(key)
        got = n(got).reshape(ref.shape)
        assert np.isfinite(got).all(), key
        if np.abs(ref).max() == 0:
            assert np.abs(got).max() == 0, key
            continue
        e, flips, rest = rows_check(got.reshape(c["N"], -1), ref.reshape(c["N"], -1), bar=1e-4, max_flips=3)
        print(f"train step {name} {key}: rel-L2 {e:.2e}, flip rows {flips}, without them {rest:.2e}")
        assert rest < 1e-4, (key, e, flips, rest)


@pytest.mark.parametrize("name", FAST_CASES)
def test_fastgs_fused_training_step_follows_the_reference_train_step(lfs, name):
    """fastgs.render_and_backward (the fused, autograd-free training step of the default path: forward, the fused L1 + D-SSIM kernels on the un-clamped CHW image,
    backward, densification_info) against fast_rasterize() -> compute_photometric_loss -> backward() of the reference's own code ("train/..." entries of
    tests/golden/ref_fast_raster.npz): loss 5e-6 relative, gradients 2e-5 relative L2, densification_info alike."""
    from lichtfeld_studio_amd import fastgs, loader
    from lichtfeld_studio_amd.rasterizer import SplatModel
    c, g = U.CASES[name], lambda k: FAST[f"{name}/train/{k}"]
    s = U.scene(c)
    W, H = c["W"], c["H"]
    cam = loader.CameraData(0, 1, 0, W, H, np.float32(c["focal"]), np.float32(c["focal"] * 1.05), np.float32(W / 2 + 0.5), np.float32(H / 2 - 0.25), s["R"], s["T"],
                            np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32), "view", "")
    mk = lambda a: t(a).contiguous().requires_grad_(True)
    model = SplatModel(mk(s["means"]), mk(s["sh0"]), mk(s["shN"]), mk(s["scaling"]), mk(s["rotation"]), mk(s["opacity"]), c["sh_degree"], active_sh_degree=c["active"])
    w2c, K = t(loader.world_to_view(cam)).unsqueeze(0), loader.intrinsics(cam, W, H)
    R, tt = s["R"].astype(np.float64), s["T"].astype(np.float64)
    settings = fastgs.FastGSSettings(t((-(R.T @ tt)).astype(np.float32)), (c["active"] + 1) ** 2, W, H, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), 0.01, 1e10)
    grads = [torch.zeros_like(p) for p in model.parameters()]
    loss = torch.zeros(1, device="cuda:0")
    dens = t(FAST[f"{name}/densification_info_in"]).clone()
    fastgs.render_and_backward(settings, w2c, model, t(U.target_image(c)), 1.0, grads, loss, dens, loss="l1_ssim", lambda_dssim=0.2)
    torch.cuda.synchronize()
    print(f"fastgs train step {name}: loss {float(loss):.6f} vs {float(g('loss')):.6f}")
    assert abs(float(loss) - float(g("loss"))) <= 5e-6 * float(g("loss"))
    for key, got in zip(("g_means", "g_sh0", "g_shN", "g_scaling", "g_rotation", "g_opacity"), grads):
        ref = g(key)
        got = n(got).reshape(ref.shape)
        assert np.isfinite(got).all(), key
        if np.abs(ref).max() == 0:
            assert np.abs(got).max() == 0, key
            continue
        e, flips, rest = rows_check(got.reshape(c["N"], -1), ref.reshape(c["N"], -1), bar=2e-5, max_flips=3)
        print(f"fastgs train step {name} {key}: rel-L2 {e:.2e}, flip rows {flips}, without them {rest:.2e}")
        assert rest < 2e-5, (key, e, flips, rest)
    ref = g("densification_info")
    assert np.abs(n(dens) - ref).max() <= 2e-3 * np.abs(ref).max()


def test_fused_training_step_with_bilateral_grid_follows_the_reference_train_step(lfs):
    """BASELINE config 5's step: fused.render_and_backward(bilateral=...) - the rendered image goes through the per-image bilateral grid before the L1 + D-SSIM loss -
    plus the fused TV term, against rasterize() -> BilateralGrid::apply -> compute_photometric_loss -> regularisers -> tv_loss of the reference's own code
    ("train_bilateral/..." entries): loss 5e-6, parameter gradients 1e-4, grid gradients 2e-5 of their maximum."""
    from lichtfeld_studio_amd import bilateral_grid as bgm, fused, loader
    from lichtfeld_studio_amd.rasterizer import Camera, SplatModel
    name = "deg1_96x64_background"
    c, g = U.CASES[name], lambda k: GOLD[f"{name}/train_bilateral/{k}"]
    s = U.scene(c)
    W, H = c["W"], c["H"]
    cam = loader.CameraData(0, 1, 0, W, H, np.float32(c["focal"]), np.float32(c["focal"] * 1.05), np.float32(W / 2 + 0.5), np.float32(H / 2 - 0.25), s["R"], s["T"],
                            np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32), "view", "")
    mk = lambda a: t(a).contiguous().requires_grad_(True)
    model = SplatModel(mk(s["means"]), mk(s["sh0"]), mk(s["shN"]), mk(s["scaling"]), mk(s["rotation"]), mk(s["opacity"]), c["sh_degree"], active_sh_degree=c["active"])
    camera = Camera(t(loader.world_to_view(cam)).unsqueeze(0), t(loader.intrinsics(cam, W, H)).unsqueeze(0), W, H)
    b = U.BILATERAL
    grid = bgm.BilateralGrid(b["n_images"], b["gW"], b["gH"], b["gL"])
    with torch.no_grad():
        grid.parameters().add_(t(U.bilateral_delta()))
    grid.grids.grad = torch.zeros_like(grid.grids)
    grads = [torch.zeros_like(p) for p in model.parameters()]
    loss = torch.zeros(1, device="cuda:0")
    fused.render_and_backward(camera, model, t(np.array(c["bg"], np.float32)), t(U.target_image(c)), 1.0, grads, loss, accumulate=False, loss="l1_ssim", lambda_dssim=0.2,
                              scale_reg=0.01, opacity_reg=0.01, bilateral=grid, image_idx=b["image_idx"])
    grid.tv_loss_fused(b["tv_weight"], loss)                 # the trainer adds the TV term once per step (trainer.py _bilateral_step; trainer.cpp:699-706)
    torch.cuda.synchronize()
    reg = 0.01 * np.exp(s["scaling"].astype(np.float64)).mean() + 0.01 * (1 / (1 + np.exp(-s["opacity"].astype(np.float64)))).mean()
    print(f"bilateral train step: loss {float(loss):.6f} + regularisers {reg:.6f} vs {float(g('loss')):.6f}")
    assert abs(float(loss) + reg - float(g("loss"))) <= 5e-6 * float(g("loss"))
    for key, got in zip(("g_means", "g_sh0", "g_shN", "g_scaling", "g_rotation", "g_opacity"), grads):
        ref = g(key)
        e, flips, rest = rows_check(n(got).reshape(c["N"], -1), ref.reshape(c["N"], -1), bar=1e-4, max_flips=3)
        print(f"bilateral train step {key}: rel-L2 {e:.2e}, flip rows {flips}, without them {rest:.2e}")
        assert rest < 1e-4, (key, e, flips, rest)
    gg, ref = n(grid.grids.grad), g("g_grids")
    e = np.abs(gg - ref).max() / np.abs(ref).max()
    print(f"bilateral train step g_grids: max-abs / max {e:.2e}")
    assert e <= 2e-5
