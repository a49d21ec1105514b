"""GPU parity: bilateral-grid slice fwd / bwd and TV loss (csrc/bilateral_grid.hip) against oracle/bilateral.py
(float64 = truth). Tolerances (fp32, stated): forward |diff| <= 2e-5 * (1 + |ref|); grad_rgb rel-L2 <= 2e-5 after setting
aside pixels whose luma sits within 1e-4 of a z-cell boundary (the discontinuity mask :150 flips there); grad_grid rel-L2 <= 2e-5
(float atomics: order-dependent rounding only); TV loss rel 1e-5, TV grad rel-L2 1e-6."""
import numpy as np
import pytest
import torch

from gpu_util import n, noise_allclose, noise_check, rel_l2, t
from oracle import bilateral as ob
from lichtfeld_studio_amd.capi import LfsError

pytestmark = pytest.mark.gpu


def _case(seed, L, H, W, h, w, lo=0.02, hi=0.98):
    rng = np.random.default_rng(seed)
    grid = (np.broadcast_to(np.eye(4)[:3].reshape(12)[:, None, None, None], (12, L, H, W)) + 0.3 * rng.standard_normal((12, L, H, W))).astype(np.float32)
    rgb = (rng.random((h, w, 3)) * (hi - lo) + lo).astype(np.float32)
    go = rng.standard_normal((h, w, 3)).astype(np.float32)
    return grid, rgb, go


# LDS window path (big images), generic path (grid finer than the pixel tiles), ragged sizes, L = 1
CASES = [dict(L=8, H=16, W=16, h=270, w=480), dict(L=8, H=16, W=16, h=67, w=131), dict(L=4, H=40, W=90, h=24, w=100),
         dict(L=1, H=3, W=2, h=9, w=70), dict(L=8, H=16, W=16, h=2, w=2)]


@pytest.mark.parametrize("cfg", CASES)
def test_slice_forward_backward_match_oracle(lfs, cfg):
    from lichtfeld_studio_amd import bilateral_grid as bg
    grid, rgb, go = _case(3, **cfg)
    out = n(bg.slice_forward(t(grid), t(rgb)))
    ref = ob.slice_forward(grid, rgb, np.float64)
    assert np.all(np.abs(out - ref) <= 2e-5 * (1 + np.abs(ref))), np.abs(out - ref).max()
    gg, gr = bg.slice_backward(t(grid), t(rgb), t(go))
    rgg, rgr = ob.slice_backward(grid, rgb, go, np.float64)
    noise_check(f"bilateral grid grad vs fp64 oracle {cfg}", rel_l2(n(gg), rgg), 2e-5)
    z = (0.299 * rgb[..., 0].astype(np.float64) + 0.587 * rgb[..., 1] + 0.114 * rgb[..., 2]) * (cfg["L"] - 1)
    ok = np.abs(z - np.round(z)) > 1e-4
    assert ok.mean() > 0.99 or cfg["L"] == 1
    if cfg["L"] > 1:
        assert rel_l2(n(gr)[ok], rgr[ok]) < 2e-5, rel_l2(n(gr)[ok], rgr[ok])
    else:
        assert rel_l2(n(gr), rgr) < 2e-5


def test_slice_chw_and_clamp_extensions(lfs):
    """chw layout == permuted hwc bit for bit; clamp_input == torch.clamp before the op, gradient masked outside [0, 1]."""
    from lichtfeld_studio_amd import bilateral_grid as bg
    grid, rgb, go = _case(5, 8, 16, 16, 120, 200, lo=-0.3, hi=1.3)
    G, x, g = t(grid), t(rgb), t(go)
    ref_out = bg.slice_forward(G, x.clamp(0, 1))
    out = bg.slice_forward(G, x, clamp_input=True)
    assert torch.equal(out, ref_out)
    out_chw = bg.slice_forward(G, x.permute(2, 0, 1).contiguous(), chw=True, clamp_input=True)
    assert torch.equal(out_chw.permute(1, 2, 0), ref_out)
    gg0, gr0 = bg.slice_backward(G, x.clamp(0, 1), g)
    gg1, gr1 = bg.slice_backward(G, x, g, clamp_input=True)
    inside = ((x >= 0) & (x <= 1)).float()
    assert torch.equal(gr1, gr0 * inside)
    noise_allclose("bilateral clamp_input grid grad", gg1, gg0, rtol=1e-4, atol=1e-5 * float(gg0.abs().max()))
    gg2, gr2 = bg.slice_backward(G, x.permute(2, 0, 1).contiguous(), g.permute(2, 0, 1).contiguous(), chw=True, clamp_input=True)
    assert torch.equal(gr2.permute(1, 2, 0), gr1)
    noise_allclose("bilateral chw grid grad", gg2, gg1, rtol=1e-4, atol=1e-5 * float(gg0.abs().max()))
    # accumulation into an existing gradient
    acc = torch.ones_like(G)
    bg.slice_backward(G, x, g, clamp_input=True, grad_grid=acc)
    noise_allclose("bilateral accumulate grid grad", acc - 1, gg1, rtol=1e-4, atol=2e-5 * float(gg0.abs().max()))
    with pytest.raises(LfsError):
        bg.slice_forward(G[:11], x)
    with pytest.raises(LfsError):
        bg.slice_forward(G, x[:1])      # h < 2: the uniform coordinate divides by (h - 1)


def test_tv_loss_and_module(lfs):
    from lichtfeld_studio_amd import bilateral_grid as bg
    rng = np.random.default_rng(9)
    grids = rng.standard_normal((5, 12, 8, 16, 16)).astype(np.float32)
    G = t(grids)
    tv = float(bg.tv_loss_forward(G))
    ref = float(ob.tv_forward(grids, np.float64))
    assert abs(tv - ref) < 1e-5 * ref
    gr = n(bg.tv_loss_backward(G, torch.tensor(0.37)))
    assert rel_l2(gr, ob.tv_backward(grids, 0.37, np.float64)) < 1e-6
    acc = torch.ones_like(G)
    bg.tv_loss_backward(G, 0.37, acc)
    assert np.allclose(n(acc) - 1, gr, rtol=1e-5, atol=1e-7)

    # the module: identity at init, autograd through apply() and tv_loss() equals the raw entry points
    m = bg.BilateralGrid(3, 16, 16, 8)
    img = torch.rand(3, 90, 160, device="cuda:0") * 1.2 - 0.1
    out = m.apply(img, 1)
    assert torch.allclose(out, img.clamp(0, 1), atol=1e-6) and float(m.tv_loss()) == 0.0
    with torch.no_grad():
        m.grids.add_(0.1 * torch.randn_like(m.grids))
    img.requires_grad_(True)
    loss = (m.apply(img[None], 2)[0] ** 2).sum() + 10.0 * m.tv_loss()
    loss.backward()
    g_img, g_grid = img.grad.clone(), m.grids.grad.clone()
    assert float(g_grid[0].abs().max()) > 0 and float(g_grid[2].abs().max()) > float(g_grid[0].abs().max())   # image 0: only the TV term
    # fused (no-autograd) path, CHW layout
    m.grids.grad = None
    loss_acc = torch.zeros(1, device="cuda:0")
    x = img.detach()
    y = m.apply_fused(x, 2, chw=True)
    gx = m.apply_fused_backward(x, 2, 2 * y, chw=True)
    m.tv_loss_fused(10.0, loss_acc)
    assert torch.allclose(gx, g_img, rtol=1e-4, atol=1e-6)
    noise_allclose("bilateral fused-path grid grad", m.grids.grad, g_grid, rtol=1e-4, atol=1e-5 * float(g_grid.abs().max()))
    assert abs(float(loss_acc) + float((y ** 2).sum()) - float(loss)) < 1e-4 * float(loss)


def test_gut_trainer_with_bilateral_grid_matches_autograd_composition(lfs):
    """3DGUT fused step + bilateral grid (clamp -> slice -> un-clamped L1 + SSIM loss -> slice backward -> rasterizer backward, TV regulariser,
    grid optimizer) against torch autograd over rasterize() + BilateralGrid.apply() + photometric_loss(): loss and first-step gradients."""
    from lichtfeld_studio_amd import bilateral_grid as bg, losses, scenes
    from lichtfeld_studio_amd.rasterizer import rasterize
    from lichtfeld_studio_amd.trainer import GutTrainer
    from gpu_util import rel_l2
    dev = torch.device("cuda:0")
    sc = scenes.syn_a(n=4000, sh_degree=1)
    tr = GutTrainer(sc, dev, iterations=300, loss="l1_ssim", use_bilateral_grid=True, tv_loss_weight=10.0)
    with torch.no_grad():
        tr.bilateral.grids.add_(0.05 * torch.randn(tr.bilateral.grids.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(3)))
    target = (scenes.target_image(sc.height, sc.width) * 0.8 + 0.1).to(dev)
    out = rasterize(tr.camera(0), tr.model, tr.bg, 1.0, False, False)
    shown = tr.bilateral.apply(out.image, 0)
    loss_ref = losses.photometric_loss(shown, target, 0.2) + 10.0 * tr.bilateral.tv_loss()
    loss_ref.backward()
    ref_grads = [p.grad.clone() for p in tr.model.parameters()]
    ref_grid = tr.bilateral.grids.grad.clone()
    for p in tr.model.parameters():
        p.grad = None
    tr.bilateral.grids.grad = torch.zeros_like(tr.bilateral.grids)
    grids_before = tr.bilateral.grids.detach().clone()
    loss = tr.train_step([target], views=[0])
    assert abs(float(loss) - float(loss_ref)) < 3e-6 * max(1.0, float(loss_ref))
    for name, g, r in zip(["means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opacities"], tr.bucket.views, ref_grads):
        assert rel_l2(n(g), n(r).reshape(n(g).shape)) < 2e-3, (name, rel_l2(n(g), n(r).reshape(n(g).shape)))
    assert float(ref_grid.abs().max()) > 0 and bool((tr.bilateral.grids.detach() != grids_before).any())
    losses_seen = [float(tr.train_step([target], views=[0])) for _ in range(40)]
    assert np.isfinite(losses_seen).all() and losses_seen[-1] < losses_seen[0]
