"""GPU: the fused L2 path (fused.py: explicit forward/backward over the fused element-wise kernels) against the
op-by-op autograd mirror (rasterizer.py), which tests/test_gpu_pipeline.py pins to the oracle, and against the
oracle directly."""
import numpy as np
import pytest
import torch

from gpu_util import n, noise_check, rel_l2
from test_gpu_pipeline import _oracle_step

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_activation_kernels_match_torch(lfs):
    from lichtfeld_studio_amd import fused
    g = torch.Generator().manual_seed(0)
    N = 10007
    rq, rs, ro = torch.randn(N, 4, generator=g).to(DEV), (torch.randn(N, 3, generator=g) - 3).to(DEV), (2 * torch.randn(N, generator=g)).to(DEV)
    rq[5] = 0.0  # degenerate quaternion: F.normalize clamps the norm at 1e-12
    q, s, o = fused.activations_fwd(rq, rs, ro)
    tq, ts, to = torch.nn.functional.normalize(rq, dim=-1), rs.exp(), torch.sigmoid(ro)
    assert torch.allclose(q, tq, atol=1e-6) and torch.allclose(s, ts, rtol=1e-6) and torch.allclose(o, to, atol=1e-6)
    vq, vs, vo = torch.randn(N, 4, generator=g).to(DEV), torch.randn(N, 3, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
    a = [x.clone().requires_grad_(True) for x in (rq, rs, ro)]
    (torch.nn.functional.normalize(a[0], dim=-1) * vq).sum().backward(); (a[1].exp() * vs).sum().backward(); (torch.sigmoid(a[2]) * vo).sum().backward()
    gq, gs, go = torch.empty_like(rq), torch.empty_like(rs), torch.empty_like(ro)
    fused.activations_bwd(rq, s, o, vq, vs, vo, gq, gs, go, accumulate=False)
    keep = torch.ones(N, dtype=torch.bool, device=DEV); keep[5] = False
    assert torch.allclose(gq[keep], a[0].grad[keep], atol=1e-5, rtol=1e-5) and torch.allclose(gs, a[1].grad, rtol=1e-5, atol=1e-7) and torch.allclose(go, a[2].grad, atol=1e-6)
    g2 = [x.clone() for x in (gq, gs, go)]
    fused.activations_bwd(rq, s, o, vq, vs, vo, *g2, accumulate=True)
    assert torch.allclose(g2[1], 2 * gs) and torch.allclose(g2[2], 2 * go)


@pytest.mark.parametrize("deg,K", [(0, 1), (1, 4), (3, 16), (2, 16)])
def test_sh_model_kernels_match_the_op_by_op_path(lfs, deg, K):
    from lichtfeld_studio_amd import fused, ops
    g = torch.Generator().manual_seed(deg)
    N = 5003
    means = torch.randn(N, 3, generator=g).to(DEV)
    sh0, shN = torch.randn(N, 1, 3, generator=g).to(DEV), torch.randn(N, K - 1, 3, generator=g).to(DEV)
    radii = torch.randint(0, 3, (1, N, 2), generator=g, dtype=torch.int32).to(DEV)
    vm = torch.eye(4); vm[:3, :3] = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]; vm[:3, 3] = torch.randn(3, generator=g)
    vm = vm[None].contiguous().to(DEV)
    colors = fused.sh_model_fwd(deg, means, vm, sh0, shN, radii)
    campos = torch.inverse(vm)[:, :3, 3]
    dirs = (means[None] - campos[:, None])[0].contiguous()
    mask = (radii > 0).all(-1)[0].contiguous()
    coeffs = torch.cat([sh0, shN], 1).contiguous()
    ref = torch.clamp_min(ops.spherical_harmonics_fwd(deg, dirs, coeffs, mask) + 0.5, 0.0)
    assert torch.allclose(colors, ref, atol=2e-6)
    v_colors = torch.randn(N, 3, generator=g).to(DEV)
    v_means0 = torch.randn(N, 3, generator=g).to(DEV)
    vc_ref, vd_ref = ops.spherical_harmonics_bwd(K, deg, dirs, coeffs, mask, (v_colors * (ref > 0)).contiguous(), True)
    v_sh0, v_shN, v_means = torch.empty_like(sh0), torch.empty_like(shN), v_means0.clone()
    fused.sh_model_bwd(deg, means, vm, sh0, shN, radii, colors, v_colors, v_sh0, v_shN, v_means, accumulate=False)
    assert torch.allclose(v_sh0, vc_ref[:, :1], atol=2e-6) and torch.allclose(v_shN, vc_ref[:, 1:], atol=2e-6)
    assert torch.allclose(v_means, v_means0 + vd_ref, atol=5e-5, rtol=1e-4), float((v_means - v_means0 - vd_ref).abs().max())
    fused.sh_model_bwd(deg, means, vm, sh0, shN, radii, colors, v_colors, v_sh0, v_shN, v_means, accumulate=True)
    assert torch.allclose(v_sh0, 2 * vc_ref[:, :1], atol=4e-6) and torch.allclose(v_shN, 2 * vc_ref[:, 1:], atol=4e-6)


def test_mse_kernel_matches_torch(lfs):
    from lichtfeld_studio_amd import fused
    g = torch.Generator().manual_seed(3)
    H, W = 117, 203
    render = (torch.rand(1, H, W, 3, generator=g) * 1.6 - 0.3).to(DEV)   # some values outside [0,1]
    target = torch.rand(3, H, W, generator=g).to(DEV)
    loss = torch.zeros(1, device=DEV)
    v = fused.mse_loss_fwd_bwd(render, target, 0.5, loss)
    r = render.clone().requires_grad_(True)
    ref = 0.5 * torch.nn.functional.mse_loss(torch.clamp(r[0].permute(2, 0, 1), 0, 1), target)
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-6 and torch.allclose(v, r.grad, atol=1e-9, rtol=1e-5)


def test_fused_step_gradients_match_autograd_path_and_oracle(lfs, oracle_mod):
    from lichtfeld_studio_amd import scenes
    from lichtfeld_studio_amd.fused import render_and_backward
    from lichtfeld_studio_amd.rasterizer import rasterize
    from lichtfeld_studio_amd.trainer import GutTrainer
    sc = scenes.syn_a(n=6000, sh_degree=2)
    tr = GutTrainer(sc, DEV, iterations=100)
    target = scenes.target_image(sc.height, sc.width).to(DEV)
    params = tr.model.parameters()
    grads = [torch.full_like(p, 7.0) for p in params]   # must be overwritten, not accumulated
    loss = torch.zeros(1, device=DEV)
    out = render_and_backward(tr.camera(0), tr.model, tr.bg, target, 1.0, grads, loss, accumulate=False)
    ref = rasterize(tr.camera(0), tr.model, tr.bg)
    ref_loss = torch.nn.functional.mse_loss(ref.image, target)
    ref_loss.backward()
    assert out.n_isects == ref.n_isects and abs(float(loss) - float(ref_loss)) < 1e-6
    # (campos = -R^T t in the fused kernel vs torch.inverse in the mirror: colours agree to an ulp, not bitwise)
    dimg = (torch.clamp(out.image_hwc[0].permute(2, 0, 1), 0, 1) - ref.image).abs()
    assert float(dimg.mean()) < 1e-7 and float((dimg > 1e-5).float().mean()) < 1e-3, (float(dimg.mean()), float(dimg.max()))
    names = ["means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opacities"]
    errs = {name: rel_l2(n(g), n(p.grad)) for name, g, p in zip(names, grads, params)}
    print("fused vs autograd rel-L2:", errs)
    assert max(errs.values()) < 1e-4, errs
    o_loss, _, o_g, o_I = _oracle_step(oracle_mod, sc, 0, target.cpu().numpy(), 2)
    assert out.n_isects == o_I and abs(float(loss) - o_loss) < 1e-6
    for name, g in zip(names, grads):
        assert rel_l2(n(g), o_g[name]) < 2e-3, (name, rel_l2(n(g), o_g[name]))
    # second view accumulates
    g1 = [g.clone() for g in grads]
    render_and_backward(tr.camera(0), tr.model, tr.bg, target, 1.0, grads, loss, accumulate=True)
    for name, a, b in zip(names, grads, g1):   # (second draw of the float-atomic sums added to the first: the difference IS their run-to-run noise)
        noise_check(f"accumulate=True doubles {name}", rel_l2(n(a), 2 * n(b)), 1e-4)


def test_fused_and_autograd_trainers_take_the_same_steps(lfs):
    from lichtfeld_studio_amd import scenes
    from lichtfeld_studio_amd.trainer import GutTrainer
    sc = scenes.syn_a(n=3000, sh_degree=1)
    a = GutTrainer(sc, DEV, iterations=200, fused_l2=True)
    b = GutTrainer(sc, DEV, iterations=200, fused_l2=False)
    target = torch.rand(3, sc.height, sc.width, generator=torch.Generator().manual_seed(1)).to(DEV) * 0.5
    for _ in range(5):
        la, lb = float(a.train_step([target], views=[0])), float(b.train_step([target], views=[0]))
        noise_check("fused vs autograd trainer loss", abs(la - lb), 1e-4 * max(1.0, abs(lb)))   # (parameters drift apart by atomics-order noise through Adam)
    for pa, pb in zip(a.model.parameters(), b.model.parameters()):
        # Adam normalises every gradient to ~lr-sized steps, so atomics-order noise on tiny gradients is visible: compare loosely
        assert float((pa - pb).abs().max()) < 5e-3 and rel_l2(n(pa), n(pb)) < 1e-4


def test_inline_shN_adam_is_bit_identical_to_separate_kernels(lfs):
    """lfs_sh_model_bwd_adam (SH backward + Adam on shN in one kernel, no shN gradient tensor) against lfs_sh_model_bwd followed by
    lfs_adam_step on shN, same inputs: shN, both moments, v_sh0 and v_means bit-identical, over several chained updates. (Whole
    training runs cannot be compared bit for bit: the rasterizer backward sums with float atomics.)"""
    from lichtfeld_studio_amd import fused, ops, scenes
    from lichtfeld_studio_amd.fused_adam import FusedAdam
    dev = torch.device(DEV)
    sc = scenes.syn_a(n=7001, sh_degree=3).to(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    N = sc.means.shape[0]
    radii = (torch.rand(1, N, 2, device=dev, generator=g) > 0.25).int() * 3          # a quarter of the Gaussians invisible
    vm = sc.viewmats[:1].contiguous()
    for deg in (3, 1):
        pa, pb = sc.shN.clone(), sc.shN.clone()
        oa, ob = FusedAdam([{"params": [pa], "lr": 2.5e-3 / 20}]), FusedAdam([{"params": [pb], "lr": 2.5e-3 / 20}])
        for it in range(4):
            colors = fused.sh_model_fwd(deg, sc.means, vm, sc.sh0, pa, radii)
            v_colors = torch.randn(N, 3, device=dev, generator=g)
            # separate kernels
            v_sh0_b, v_shN_b, v_means_b = torch.empty_like(sc.sh0), torch.empty_like(pb), torch.ones(N, 3, device=dev)
            fused.sh_model_bwd(deg, sc.means, vm, sc.sh0, pb, radii, colors, v_colors, v_sh0_b, v_shN_b, v_means_b, False)
            pb.grad = v_shN_b
            ob.step(2000 + it)
            # one kernel
            v_sh0_a, v_means_a = torch.empty_like(sc.sh0), torch.ones(N, 3, device=dev)
            fused.sh_model_bwd_adam(deg, sc.means, vm, sc.sh0, pa, radii, colors, v_colors, v_sh0_a, v_means_a, oa.prepare_inline(pa))
            pa.grad = torch.full_like(pa, float("nan"))      # must not be read
            oa.step(2000 + it)
            assert torch.equal(pa, pb) and torch.equal(v_sh0_a, v_sh0_b) and torch.equal(v_means_a, v_means_b), (deg, it)
            sa, sb = oa._state(pa), ob._state(pb)
            assert sa["step_count"] == sb["step_count"] == it + 1
            assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
        assert float((pa - sc.shN).abs().max()) > 0


def test_inline_shN_adam_trainer_path_trains(lfs):
    from lichtfeld_studio_amd import scenes
    from lichtfeld_studio_amd.trainer import GutTrainer
    dev = torch.device(DEV)
    sc = scenes.syn_a(n=6000, sh_degree=3)
    target = scenes.target_image(sc.height, sc.width).to(dev)
    a, b = GutTrainer(sc, dev, iterations=100), GutTrainer(sc, dev, iterations=100)
    b.inline_shN_adam = False
    a.iteration = b.iteration = 998
    la = [float(a.train_step([target], views=[0])) for _ in range(12)]
    lb = [float(b.train_step([target], views=[0])) for _ in range(12)]
    noise_check("inline shN Adam: 12 losses", float(np.max(np.abs(np.array(la) - lb) / np.abs(lb))), 1e-4)
    assert la[-1] < la[0]
    noise_check("inline shN Adam: shN after 12 steps", float((a.model.shN - b.model.shN).abs().max()), 2e-3)
    assert float((a.model.shN.detach() - sc.shN.to(dev)).abs().max()) > 0


def test_multi_view_sh_kernels_match_per_view_launches(lfs):
    """lfs_sh_model_fwd_views / _bwd_views (the owner side of SH-sharded data parallelism: all views in one launch, coefficients read once)
    against one lfs_sh_model_fwd / _bwd launch per view; and the inline-Adam form against the stored gradient + lfs_adam_step."""
    from lichtfeld_studio_amd import fused, scenes
    from lichtfeld_studio_amd.fused_adam import FusedAdam
    dev = torch.device(DEV)
    sc = scenes.syn_a(n=5003, sh_degree=3).to(dev)
    g = torch.Generator(device=dev).manual_seed(2)
    n, V, S = sc.means.shape[0], 5, 5100                     # S > n: padded rows, as for the last shard
    vms = torch.eye(4, device=dev).repeat(V, 1, 1)
    vms[:, :3, 3] = torch.randn(V, 3, device=dev, generator=g) * 2
    radii = torch.zeros(V, S, 2, dtype=torch.int32, device=dev)
    radii[:, :n] = (torch.rand(V, n, 2, device=dev, generator=g) > 0.3).int() * 4
    v_colors = torch.randn(V, S, 3, device=dev, generator=g)
    for deg in (3, 2, 0):
        colors = fused.sh_model_fwd_views(deg, sc.means, vms, sc.sh0, sc.shN, radii)
        ref_sh0, ref_shN, ref_means = torch.zeros_like(sc.sh0), torch.zeros_like(sc.shN), torch.ones(n, 3, device=dev)
        for v in range(V):
            c = fused.sh_model_fwd(deg, sc.means, vms[v:v + 1].contiguous(), sc.sh0, sc.shN, radii[v:v + 1, :n].contiguous())
            vis = (radii[v, :n] > 0).all(-1)
            assert torch.equal(colors[v, :n][vis], c[vis]), (deg, v)
            fused.sh_model_bwd(deg, sc.means, vms[v:v + 1].contiguous(), sc.sh0, sc.shN, radii[v:v + 1, :n].contiguous(), c, v_colors[v, :n].contiguous(),
                               ref_sh0, ref_shN, ref_means, v > 0)
        assert float(colors[:, n:].abs().max()) == 0
        v_sh0, v_shN, v_means = torch.empty_like(sc.sh0), torch.empty_like(sc.shN), torch.ones(n, 3, device=dev)
        fused.sh_model_bwd_views(deg, sc.means, vms, sc.sh0, sc.shN, radii, colors, v_colors, v_sh0, v_shN, v_means, False)
        assert torch.equal(v_sh0, ref_sh0) and torch.equal(v_shN, ref_shN)           # same products, same summation order over the views
        assert torch.allclose(v_means, ref_means, rtol=1e-5, atol=1e-6)              # dL/d(dirs): summed in registers vs in memory
        # accumulate: a second call doubles the coefficient gradients
        fused.sh_model_bwd_views(deg, sc.means, vms, sc.sh0, sc.shN, radii, colors, v_colors, v_sh0, v_shN, v_means, True)
        assert torch.allclose(v_shN, 2 * ref_shN, rtol=1e-6, atol=1e-7) and torch.allclose(v_sh0, 2 * ref_sh0, rtol=1e-6, atol=1e-7)
        if deg == 0:
            continue
        # inline Adam == stored gradient + optimizer step
        pa, pb = sc.shN.clone(), sc.shN.clone()
        oa, ob = FusedAdam([{"params": [pa], "lr": 1.25e-4}]), FusedAdam([{"params": [pb], "lr": 1.25e-4}])
        for it in range(2):
            ca = fused.sh_model_fwd_views(deg, sc.means, vms, sc.sh0, pa, radii)
            s0a, ma = torch.empty_like(sc.sh0), torch.zeros(n, 3, device=dev)
            fused.sh_model_bwd_views(deg, sc.means, vms, sc.sh0, pa, radii, ca, v_colors, s0a, None, ma, False, adam=oa.prepare_inline(pa))
            s0b, gb, mb = torch.empty_like(sc.sh0), torch.empty_like(pb), torch.zeros(n, 3, device=dev)
            fused.sh_model_bwd_views(deg, sc.means, vms, sc.sh0, pb, radii, ca, v_colors, s0b, gb, mb, False)
            pb.grad = gb
            ob.step(2000 + it)
            pa.grad = torch.full_like(pa, float("nan"))
            oa.step(2000 + it)
            assert torch.equal(pa, pb) and torch.equal(s0a, s0b) and torch.equal(ma, mb), (deg, it)
        assert float((pa - sc.shN).abs().max()) > 0


def test_mse_folded_into_backward_matches_separate_loss_kernel(lfs):
    """lfs_..._bwd_prepared_mse (dL/d(render) derived in the rasterizer backward's prologue) against lfs_mse_loss_fwd_bwd + the prepared
    backward: same per-pixel expressions -> gradients equal up to the backward's atomic summation order; loss equal to fp32 summation order.
    Includes pixels outside [0,1] (clamp mask), a background, and a ragged image size."""
    from lichtfeld_studio_amd import fused, scenes
    from lichtfeld_studio_amd.rasterizer import Camera, SplatModel
    dev = torch.device(DEV)
    sc = scenes._syn_box("SYN-M", 5, 5000, 203, 117, 160.0, 2, sh_degree=1).to(dev)
    sc.raw_scales += 1.8
    sc.sh0 *= 8.0                                             # over- and under-shooting colours
    model = SplatModel(sc.means, sc.sh0, sc.shN, sc.raw_scales, sc.raw_quats, sc.raw_opacities, 1)
    cam = Camera(sc.viewmats[:1].contiguous(), sc.Ks[:1].contiguous(), sc.width, sc.height)
    target = scenes.target_image(sc.height, sc.width).to(dev)
    bg = torch.tensor([0.2, 0.5, 0.9], device=dev)
    res = []
    for flag in (True, False, False):
        fused.FUSE_MSE_INTO_BACKWARD = flag
        grads = [torch.zeros_like(p) for p in model.parameters()]
        loss = torch.zeros(1, device=dev)
        out = fused.render_and_backward(cam, model, bg, target, 0.7, grads, loss, accumulate=False, loss="mse")
        res.append((float(loss), grads, out.image_hwc))
    fused.FUSE_MSE_INTO_BACKWARD = True
    (la, ga, ia), (lb, gb, ib), (lc, gc, _) = res
    assert torch.equal(ia, ib) and float((ia < 0).float().mean() + (ia > 1).float().mean()) > 0.003
    noise_check("fused-MSE vs separate loss value", abs(la - lb), 1e-5 * max(1.0, abs(lb)))
    assert lb > 0
    for name, a, b, c in zip(["means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opacities"], ga, gb, gc):
        noise = float((b - c).abs().max())                    # run-to-run noise of the float atomics
        assert float((a - b).abs().max()) <= max(5 * noise, 1e-4 * float(b.abs().max())), (name, float((a - b).abs().max()), noise)
        assert float(b.abs().max()) > 0


def test_all_inline_adam_step_is_bit_identical_to_the_separate_kernels(lfs):
    """fused.backward_adam_all (rasterizer backward -> SH backward with Adam on sh0 / shN -> finish + activation backward + Adam on the rest, no
    gradient tensors) against the same steps through raster_finish, lfs_sh_model_bwd_adam, lfs_activations_bwd and lfs_adam_step_multi. Both
    trainers run in the deterministic accumulation mode (lfs_set_debug_flags(16)), so the rasterizer hands both exactly the same sums and every
    parameter and every Adam moment has to agree BIT FOR BIT over several steps."""
    from lichtfeld_studio_amd import scenes
    from lichtfeld_studio_amd.trainer import GutTrainer
    sc = scenes.syn_a(n=7000, sh_degree=2)
    target = torch.rand(3, sc.height, sc.width, generator=torch.Generator().manual_seed(3)).to(DEV) * 0.7
    lib = lfs.load_library()
    try:
        lib.lfs_set_debug_flags(16)
        a = GutTrainer(sc, DEV, iterations=7000)
        b = GutTrainer(sc, DEV, iterations=7000)
        b.inline_all_adam = False
        a.iteration = b.iteration = 1500          # shN is being optimised
        for _ in range(4):
            la, lb = a.train_step([target], views=[0]), b.train_step([target], views=[0])
        torch.cuda.synchronize()
    finally:
        lib.lfs_set_debug_flags(0)
    noise_check("inline-all vs separate loss value", abs(float(la) - float(lb)), 1e-5 * float(lb))   # (the loss value is a float-atomic sum: last-bit order dependence)
    assert float(la) > 0
    for name, pa, pb in zip(["means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opacities"], a.model.parameters(), b.model.parameters()):
        assert torch.equal(pa, pb), (name, float((pa - pb).abs().max()))
        sa, sb = a.optimizer.state[id(pa)], b.optimizer.state[id(pb)]
        assert sa["step_count"] == sb["step_count"] == 4
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]), name


def test_fused_front_half_matches_the_separate_kernels(lfs):
    """Activations + projection in one kernel (FUSE_ACT_PROJ, the default) against activations_fwd + projection: bit-identical radii / means2d / depths -
    same operations, same compilation unit - so nothing but the launch count changes. (The round-2 experiment that also moved the rasterizer's record
    packing into the SH colour kernel was measured to gain nothing and removed: profiles/r02/fuse_front_ab.txt.)"""
    from lichtfeld_studio_amd import fused, scenes
    from lichtfeld_studio_amd.trainer import GutTrainer
    sc = scenes.syn_a(n=9000, sh_degree=3)
    target = scenes.target_image(sc.height, sc.width).to(DEV)
    res = {}
    try:
        for mode in (True, False):
            fused.FUSE_ACT_PROJ = mode
            tr = GutTrainer(sc, DEV, iterations=100)
            grads = [torch.zeros_like(p) for p in tr.model.parameters()]
            loss = torch.zeros(1, device=DEV)
            out = fused.render_and_backward(tr.camera(0), tr.model, tr.bg, target, 1.0, grads, loss, accumulate=False)
            res[mode] = (out, [g.clone() for g in grads], float(loss))
    finally:
        fused.FUSE_ACT_PROJ = True
    (c, gc, lc), (b, gb, lb) = res[True], res[False]
    # (the loss value is a float-atomic sum of per-wavefront partials: its last bit depends on their arrival order)
    assert c.n_isects == b.n_isects and torch.equal(c.radii, b.radii) and torch.equal(c.image_hwc, b.image_hwc)
    noise_check("fused front half loss value", abs(lc - lb), 1e-5 * abs(lb))


def test_batched_views_step_matches_the_view_by_view_step(lfs):
    """Several views per step on one rank (BASELINE config 4): fused.render_views_and_backward - projections of all views, ONE SH forward and ONE SH
    backward (with shN's Adam update inside) over all views - against the same step view by view (per-view SH kernels accumulating the shN
    gradient, shN in the optimizer launch). Deterministic rasterizer sums (debug bit 4), so what differs is only the order in which the per-view SH
    terms are added: loss identical, parameters and moments to a few ulp of the update."""
    import convergence_check as cc
    from lichtfeld_studio_amd.trainer import GutTrainer
    _, sc = cc.make_task(n=4000, size=96, n_views=6, sh_degree=2)
    g = torch.Generator().manual_seed(5)
    targets = [(torch.rand(3, sc.height, sc.width, generator=g) * 0.7).to(DEV) for _ in range(4)]
    lib = lfs.load_library()
    try:
        lib.lfs_set_debug_flags(16)
        a = GutTrainer(sc, DEV, iterations=7000, views_per_rank=4)
        b = GutTrainer(sc, DEV, iterations=7000, views_per_rank=4)
        b.batch_views = False
        for it0 in (500, 1500):                   # shN out of / in the optimizer (fused_adam.cpp:68-70)
            a.iteration = b.iteration = it0
            for _ in range(3):
                la, lb = a.train_step(targets), b.train_step(targets)
        torch.cuda.synchronize()
    finally:
        lib.lfs_set_debug_flags(0)
    assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(lb)) and float(la) > 0
    for name, pa, pb in zip(["means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opacities"], a.model.parameters(), b.model.parameters()):
        sa, sb = a.optimizer.state[id(pa)], b.optimizer.state[id(pb)]
        assert sa["step_count"] == sb["step_count"], name
        assert float((pa - pb).abs().max()) <= 2e-5 * float(pb.abs().max()), (name, float((pa - pb).abs().max()))
        assert float((sa["exp_avg"] - sb["exp_avg"]).abs().max()) <= 1e-4 * float(sb["exp_avg"].abs().max()) + 1e-12, name


def test_fused_finish_grads_matches_finish_plus_activation_backward(lfs):
    """lfs_gut_finish_grads (accumulator rows -> raw-parameter gradient tensors in one pass) against raster_finish + lfs_activations_bwd + the copy of
    dL/dmeans, through fused.render_and_backward with both losses, regularisers, and a second accumulating view. Deterministic rasterizer sums, so the
    first view has to agree bit for bit; the accumulated one up to the order in which dL/d(dirs) joins dL/dmeans."""
    import convergence_check as cc
    from lichtfeld_studio_amd import fused
    from lichtfeld_studio_amd.rasterizer import SplatModel
    from lichtfeld_studio_amd.trainer import GutTrainer
    _, sc = cc.make_task(n=5000, size=128, n_views=3, sh_degree=2)
    tr = GutTrainer(sc, DEV, iterations=100)
    g = torch.Generator().manual_seed(9)
    targets = [(torch.rand(3, sc.height, sc.width, generator=g) * 0.8).to(DEV) for _ in range(2)]
    lib = lfs.load_library()
    res = {}
    try:
        lib.lfs_set_debug_flags(16)
        for loss in ("mse", "l1_ssim"):
            for flag in (True, False):
                fused.FUSE_FINISH_GRADS = flag
                grads = [torch.zeros_like(p) for p in tr.model.parameters()]
                acc = torch.zeros(1, device=DEV)
                snaps = []
                for k in range(2):
                    fused.render_and_backward(tr.camera(k), tr.model, tr.bg, targets[k], 0.5, grads, acc, accumulate=k > 0, loss=loss,
                                              scale_reg=0.01 if k == 0 else 0.0, opacity_reg=0.02 if k == 0 else 0.0)
                    snaps.append([x.clone() for x in grads])
                torch.cuda.synchronize()
                res[(loss, flag)] = (snaps, float(acc))
    finally:
        fused.FUSE_FINISH_GRADS = True
        lib.lfs_set_debug_flags(0)
    for loss in ("mse", "l1_ssim"):
        (a, la), (b, lb) = res[(loss, True)], res[(loss, False)]
        assert la > 0
        noise_check(f"finish-kernel forms, loss value {loss}", abs(la - lb) / abs(lb), 4e-6)   # (the 256 partial sums of the loss are float atomics: ~1e-7)
        for name, x, y in zip(["means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opacities"], a[0], b[0]):
            assert torch.equal(x, y), (loss, name, float((x - y).abs().max()))
        for name, x, y in zip(["means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opacities"], a[1], b[1]):
            assert float((x - y).abs().max()) <= 1e-6 * float(y.abs().max()) + 1e-12, (loss, name, float((x - y).abs().max()))
