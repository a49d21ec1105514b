"""The C++ libtorch drop-in layer (csrc/torch_ops.cpp: the reference's `gsplat::` / `fast_gs::optimizer`
signatures over the C ABI), exercised through its pybind module: identical results to the Python
mirror (both call the same C entry points), reference error behaviour (c10::Error on CPU tensors)."""
import math

import numpy as np
import pytest
import torch

from gpu_util import atomic_noise_bar, n, noise_allclose, noise_check, rel_l2


def _mod():
    import lichtfeld_studio_amd  # noqa: F401
    try:
        from lichtfeld_studio_amd import _lfs_torch_ops as m
    except ImportError:
        pytest.skip("_lfs_torch_ops.so not built (python lichtfeld-studio_amd/build.py --torch-ops)")
    return m


def test_torch_ops_module_surface_and_cpu_rejection():
    m = _mod()
    for name in ["spherical_harmonics_fwd", "spherical_harmonics_bwd", "intersect_tile", "intersect_offset", "quats_to_rotmats", "relocation",
                 "add_noise", "projection_ut_3dgs_fused", "rasterize_to_pixels_from_world_3dgs_fwd", "rasterize_to_pixels_from_world_3dgs_bwd",
                 "adam_step_wrapper", "fastgs_forward_wrapper", "fastgs_backward_wrapper", "fusedssim", "fusedssim_backward",
                 "bilateral_slice_forward", "bilateral_slice_backward", "bilateral_tv_loss_forward", "bilateral_tv_loss_backward",
                 "read_colmap_cameras_and_images", "read_colmap_cameras_and_images_text", "read_transforms_cameras_and_images", "read_colmap_point_cloud",
                 "read_colmap_point_cloud_text", "save_ply", "load_ply"]:
        assert hasattr(m, name)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        m.quats_to_rotmats(torch.randn(3, 4))


@pytest.mark.gpu
def test_torch_ops_equal_python_mirror(lfs):
    m = _mod()
    from lichtfeld_studio_amd import ops, scenes
    dev = torch.device("cuda:0")
    sc = scenes.syn_a(n=5000, sh_degree=2).to(dev)
    means, quats = sc.means, torch.nn.functional.normalize(sc.raw_quats, dim=-1)
    scales, opac = sc.raw_scales.exp(), torch.sigmoid(sc.raw_opacities)
    W, H = sc.width, sc.height
    vm, K = sc.viewmats.contiguous(), sc.Ks.contiguous()
    a = m.projection_ut_3dgs_fused(means, quats, scales, opac, vm, None, K, W, H, 0.3, 0.01, 1e4, 0.0, True, 0, None, 4, None, None, None)
    b = ops.projection_ut_3dgs_fused(means, quats, scales, opac, vm, None, K, W, H, 0.3, 0.01, 1e4, 0.0, True, lfs.CameraModelType.PINHOLE)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    radii, m2, d = a[0], a[1], a[2]
    mask = (radii > 0).all(-1)[0]
    dirs = (means - torch.inverse(vm)[0, :3, 3]).contiguous()
    sh = torch.cat([sc.sh0, sc.shN], 1).contiguous()
    c1, c2 = m.spherical_harmonics_fwd(2, dirs, sh, mask), ops.spherical_harmonics_fwd(2, dirs, sh, mask)
    assert torch.equal(c1, c2)
    vc = torch.randn_like(c1)
    g1, g2 = m.spherical_harmonics_bwd(9, 2, dirs, sh, mask, vc, True), ops.spherical_harmonics_bwd(9, 2, dirs, sh, mask, vc, True)
    assert torch.equal(g1[0], g2[0]) and torch.equal(g1[1], g2[1])
    t1 = m.intersect_tile(m2, radii, d, None, None, 1, 16, W // 16, H // 16, True)
    t2 = ops.intersect_tile(m2, radii, d, None, None, 1, 16, W // 16, H // 16, True)
    for x, y in zip(t1, t2):
        assert torch.equal(x, y)
    offs = m.intersect_offset(t1[1], 1, W // 16, H // 16)
    assert torch.equal(offs, ops.intersect_offset(t2[1], 1, W // 16, H // 16))
    colors = torch.clamp_min(c1 + 0.5, 0)[None].contiguous()
    empty_bg = torch.empty(0, device=dev)   # the L2 passes an empty {0} tensor for "no background" (rasterizer.cpp:300-303)
    f1 = m.rasterize_to_pixels_from_world_3dgs_fwd(means, quats, scales, colors, opac[None].contiguous(), empty_bg, None, W, H, 16, vm, None, K, 0, None, 4,
                                                   None, None, None, offs, t1[2])
    f2 = ops.rasterize_to_pixels_from_world_3dgs_fwd(means, quats, scales, colors, opac[None].contiguous(), None, None, W, H, 16, vm, None, K,
                                                     lfs.CameraModelType.PINHOLE, None, lfs.ShutterType.GLOBAL, None, None, None, offs, t1[2])
    for x, y in zip(f1, f2):
        assert torch.equal(x, y)
    vr, va = torch.randn_like(f1[0]), torch.randn_like(f1[1])
    b1 = m.rasterize_to_pixels_from_world_3dgs_bwd(means, quats, scales, colors, opac[None].contiguous(), empty_bg, None, W, H, 16, vm, None, K, 0, None, 4,
                                                   None, None, None, offs, t1[2], f1[1], f1[2], vr, va)
    b2 = ops.rasterize_to_pixels_from_world_3dgs_bwd(means, quats, scales, colors, opac[None].contiguous(), None, None, W, H, 16, vm, None, K,
                                                     lfs.CameraModelType.PINHOLE, None, lfs.ShutterType.GLOBAL, None, None, None, offs, t1[2], f1[1], f1[2], vr, va)
    b3 = ops.rasterize_to_pixels_from_world_3dgs_bwd(means, quats, scales, colors, opac[None].contiguous(), None, None, W, H, 16, vm, None, K,
                                                     lfs.CameraModelType.PINHOLE, None, lfs.ShutterType.GLOBAL, None, None, None, offs, t1[2], f1[1], f1[2], vr, va)
    for i, (x, y, z) in enumerate(zip(b1, b2, b3)):     # float atomics: same math, order-dependent rounding -> bar relative to two draws of the same entry
        noise_check(f"libtorch wrapper vs ctypes bwd[{i}]", rel_l2(n(x), n(y)), atomic_noise_bar(z, y))
    assert torch.equal(m.quats_to_rotmats(quats), ops.quats_to_rotmats(quats))
    p = torch.randn(10007, device=dev); e1, e2, g = torch.zeros_like(p), torch.zeros_like(p), torch.randn_like(p)
    p2, f1_, f2_ = p.clone(), e1.clone(), e2.clone()
    m.adam_step_wrapper(p, e1, e2, g, 1e-3, 0.9, 0.999, 1e-15, 10.0, 1 / math.sqrt(1 - 0.999))
    ops.adam_step_wrapper(p2, f1_, f2_, g, 1e-3, 0.9, 0.999, 1e-15, 10.0, 1 / math.sqrt(1 - 0.999))
    assert torch.equal(p, p2) and torch.equal(e1, f1_) and torch.equal(e2, f2_)
    with pytest.raises(RuntimeError, match="Unsupported number of channels"):
        m.rasterize_to_pixels_from_world_3dgs_fwd(means, quats, scales, torch.rand(1, 5000, 7, device=dev), opac[None].contiguous(), None, None, W, H, 16,
                                                  vm, None, K, 0, None, 4, None, None, None, offs, t1[2])


@pytest.mark.gpu
def test_torch_fastgs_and_ssim_wrappers_equal_python_mirror(lfs):
    """fast_gs::rasterization::forward_wrapper / backward_wrapper and fusedssim / fusedssim_backward with the reference's
    argument lists (rasterization_api.h:27-75, ssim.cuh:11-30): same C entry points as fastgs.py / losses.py -> same forward
    bits; backward up to atomic ordering. The backward takes no sh_coefficients_0, as in the reference."""
    m = _mod()
    from lichtfeld_studio_amd import fastgs, losses
    from test_oracle_fastgs import _scene
    from gpu_util import t
    sc = _scene(N=4000, W=208, H=144, seed=11, deg=2)
    a = [t(sc[k]) for k in ("means", "scales_raw", "rot_raw", "opac_raw", "sh0", "sh_rest", "w2c")]
    cam = t(sc["cam_pos"])
    fr = (sc["active_sh_bases"], sc["W"], sc["H"], sc["fx"], sc["fy"], sc["cx"], sc["cy"], 0.01, 1e10)
    out = m.fastgs_forward_wrapper(*a, cam, *fr)
    assert len(out) == 11
    image, alpha, prim, tile, inst, bucket, n_vis, n_inst, n_buckets, s0, s1 = out
    s = fastgs.FastGSSettings(cam, *fr)
    image2, alpha2, pws, iws, n_inst2 = fastgs.forward_wrapper(*a, s)
    assert n_inst == n_inst2 and torch.equal(image, image2) and torch.equal(alpha, alpha2)
    assert image.shape == (3, sc["H"], sc["W"]) and alpha.shape == (1, sc["H"], sc["W"]) and prim.dtype == torch.uint8
    gi, ga = torch.randn_like(image), torch.randn_like(alpha)
    dens1, dens2 = torch.zeros(2, 4000, device="cuda:0"), torch.zeros(2, 4000, device="cuda:0")
    g1 = m.fastgs_backward_wrapper(dens1, gi, ga, image, alpha, a[0], a[1], a[2], a[5], prim, tile, inst, bucket, a[6], cam, *fr, n_vis, n_inst,
                                   n_buckets, s0, s1)
    g2 = fastgs.backward_wrapper(dens2, gi, ga, image2, alpha2, a[0], a[1], a[2], a[4], a[5], pws, iws, a[6], s, n_inst2)
    assert len(g1) == 7 and g1[6] is None     # grad_w2c undefined: w2c does not require grad (rasterization_api.cu backward)
    for i, (x, y) in enumerate(zip(g1[:6], g2[:6])):
        assert x.shape == y.shape or x.numel() == y.numel()
        noise_allclose(f"fastgs wrapper bwd[{i}]", x, y, rtol=1e-4, atol=1e-6 + 1e-5 * float(y.abs().max()))
    assert torch.equal(dens1[0], dens2[0])
    noise_allclose("fastgs wrapper densification_info[1]", dens1[1], dens2[1], rtol=1e-4, atol=1e-7)
    # no densification info requested: empty tensor, as the reference's `densification_info.size(0) > 0` test
    g3 = m.fastgs_backward_wrapper(torch.empty(0, device="cuda:0"), gi, ga, image, alpha, a[0], a[1], a[2], a[5], prim, tile, inst, bucket, a[6], cam,
                                   *fr, n_vis, n_inst, n_buckets, s0, s1)
    noise_allclose("fastgs wrapper bwd no-dens", g3[0], g1[0], rtol=1e-4, atol=1e-6 + 1e-5 * float(g1[0].abs().max()))
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        m.fastgs_forward_wrapper(*[x.cpu() for x in a], cam.cpu(), *fr)

    x1, x2 = torch.rand(2, 3, 67, 90, device="cuda:0"), torch.rand(2, 3, 67, 90, device="cuda:0")
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    r1, r2 = m.fusedssim(C1, C2, x1, x2, True), losses.fusedssim(C1, C2, x1, x2, True)
    for u, v in zip(r1, r2):
        assert torch.equal(u, v)
    assert m.fusedssim(C1, C2, x1, x2, False)[1].numel() == 0
    dmap = torch.randn_like(x1)
    assert torch.equal(m.fusedssim_backward(C1, C2, x1, x2, dmap, *r1[1:]), losses.fusedssim_backward(C1, C2, x1, x2, dmap, *r2[1:]))
    # gs::bilateral_grid::* (include/kernels/bilateral_grid.cuh)
    from lichtfeld_studio_amd import bilateral_grid as bg
    grid, rgb, go = torch.randn(12, 8, 16, 16, device="cuda:0"), torch.rand(120, 200, 3, device="cuda:0"), torch.randn(120, 200, 3, device="cuda:0")
    assert torch.equal(m.bilateral_slice_forward(grid, rgb), bg.slice_forward(grid, rgb))
    b1, b2 = m.bilateral_slice_backward(grid, rgb, go), bg.slice_backward(grid, rgb, go)
    assert torch.equal(b1[1], b2[1])
    noise_allclose("bilateral wrapper grid grad", b1[0], b2[0], rtol=1e-4, atol=1e-5 * float(b2[0].abs().max()))
    grids = torch.randn(3, 12, 8, 16, 16, device="cuda:0")
    assert torch.allclose(m.bilateral_tv_loss_forward(grids), bg.tv_loss_forward(grids), rtol=1e-5)
    assert torch.equal(m.bilateral_tv_loss_backward(grids, torch.tensor(0.5)), bg.tv_loss_backward(grids, 0.5))
    with pytest.raises(RuntimeError, match="Grid must be"):
        m.bilateral_slice_forward(grid[:11].contiguous(), rgb)


@pytest.mark.gpu
def test_drop_in_route_training_step_equals_the_python_mirror(lfs):
    """torch_ops_route: the op-by-op step (rasterizer.cpp:224-344 under autograd + six adam_step_wrapper launches) through the COMPILED C++ wrappers of
    csrc/torch_ops.cpp against the same step through the ctypes mirror (ops.py) - both end in the same C entry points, so with the deterministic
    rasterizer sums (debug bit 4) parameters and Adam moments agree BIT FOR BIT; and against the fused C++ step to rounding. This is the route
    `bench.py --path ops` times."""
    _mod()
    from lichtfeld_studio_amd import scenes, torch_ops_route
    from lichtfeld_studio_amd.trainer import GutTrainer
    dev = "cuda:0"
    sc = scenes.syn_a(n=6000, sh_degree=2)
    target = torch.rand(3, sc.height, sc.width, generator=torch.Generator().manual_seed(11)).to(dev) * 0.7
    lib = lfs.load_library()
    res = {}
    try:
        lib.lfs_set_debug_flags(16)
        for route in ("ctypes", "compiled", "fused"):
            if route == "compiled":
                torch_ops_route.install()
            tr = GutTrainer(sc, dev, iterations=7000, fused_l2=route == "fused", fused_adam=route == "fused")
            tr.iteration = 1500
            losses = [float(tr.train_step([target], views=[0])) for _ in range(3)]
            torch.cuda.synchronize()
            res[route] = (tr, losses)
            if route == "compiled":
                torch_ops_route.uninstall()
    finally:
        torch_ops_route.uninstall()
        lib.lfs_set_debug_flags(0)
    (a, la), (b, lb), (c, lc) = res["ctypes"], res["compiled"], res["fused"]
    assert la == lb and la[0] > 0
    for name, pa, pb, pc in zip(["means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opacities"], a.model.parameters(), b.model.parameters(), c.model.parameters()):
        assert torch.equal(pa, pb), (name, float((pa - pb).abs().max()))
        sa, sb = a.optimizer.state[id(pa)], b.optimizer.state[id(pb)]
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]), name
        # fused step: same arithmetic up to the SH direction (explicit dirs vs means - campos) and summation order; Adam normalises the step size
        upd = (pa.detach() - getattr(sc, {"raw_scales": "raw_scales", "raw_quats": "raw_quats", "raw_opacities": "raw_opacities"}.get(name, name)).to(dev)).abs().max()
        tol = 2e-3 * float(upd) + 1e-7   # (elements whose gradient is pure rounding noise may step the other way: allow 0.1 % of them, as tests/test_gpu_dp2.py does)
        assert float(((pa - pc).abs() > tol).float().mean()) < 1e-3, (name, float((pa - pc).abs().max()), tol)
    assert abs(la[-1] - lc[-1]) <= 1e-5 * abs(la[-1])


@pytest.mark.gpu
def test_cxx_gut_train_step_class_equals_the_python_driver(lfs):
    """lfs::GutTrainStep (include/lfs_gsplat_torch.hpp: the training step as one call for a libtorch C++ caller - what INTEGRATION.md patches into
    Trainer::train_step) against gut_step.GutStep.train_step: both enqueue lfs_gut_train_step, so with the deterministic rasterizer sums parameters, moments and
    loss agree BIT FOR BIT over several steps - including a first attempt that overflows its deliberately small workspace and is re-run."""
    m = _mod()
    from lichtfeld_studio_amd import scenes
    from lichtfeld_studio_amd.trainer import GutTrainer
    dev = "cuda:0"
    sc = scenes.syn_a(n=5000, sh_degree=2)
    target = torch.rand(3, sc.height, sc.width, generator=torch.Generator().manual_seed(13)).to(dev) * 0.7
    lib = lfs.load_library()
    names = ["means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opacities"]
    try:
        lib.lfs_set_debug_flags(16)
        a = GutTrainer(sc, dev, iterations=7000)      # Python driver (ctypes -> lfs_gut_train_step)
        b = GutTrainer(sc, dev, iterations=7000)      # same optimizer bookkeeping, the step itself through the C++ class
        a.iteration = b.iteration = 1500
        step = m.GutTrainStep(16, 1000)                # 1000 entries: the first attempt cannot fit
        loss_b = torch.zeros(1, device=dev)
        for _ in range(3):
            la = a.train_step([target], views=[0])
            b.iteration += 1
            ad = [b.optimizer.prepare_inline(getattr(b.model, k)) for k in names]
            params = [p.detach() for p in b.model.parameters()]
            n_isects = step.step(params, [d["exp_avg"] for d in ad], [d["exp_avg_sq"] for d in ad],
                                 [[d["lr"], d["beta1"], d["beta2"], d["eps"], d["bc1_rcp"], d["bc2_sqrt_rcp"]] for d in ad], b.model.get_active_sh_degree(),
                                 b.scene.viewmats[0], b.scene.Ks[0], sc.width, sc.height, b.bg, target, 1.0, loss_b, 0.0, 0.0)
            b.optimizer.step(b.iteration)      # (every group was updated inline: nothing left to do but the bookkeeping)
            b.scheduler.step()
        torch.cuda.synchronize()
    finally:
        lib.lfs_set_debug_flags(0)
    assert step.retries() >= 1 and n_isects == a.last_n_isects > 0
    noise_check("lfs::GutTrainStep loss value", abs(float(la) - float(loss_b)), 1e-5 * float(la))   # (float-atomic partial sums of the loss value: last-bit order dependence; parameters are exact)
    assert tuple(step.render().shape) == (sc.height, sc.width, 3) and tuple(step.radii().shape) == (5000, 2)
    for name, pa, pb in zip(names, a.model.parameters(), b.model.parameters()):
        assert torch.equal(pa, pb), (name, float((pa - pb).abs().max()))
        sa, sb = a.optimizer.state[id(pa)], b.optimizer.state[id(pb)]
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]), name


def _raster_inputs(lfs, m, seed, n=3000):
    """one small scene through the libtorch wrappers up to the tile lists -> (positional args of ..._fwd, W, H)"""
    from lichtfeld_studio_amd import scenes
    dev = torch.device("cuda:0")
    sc = scenes.syn_a(seed=seed, n=n, sh_degree=0).to(dev)
    means, quats = sc.means.contiguous(), torch.nn.functional.normalize(sc.raw_quats, dim=-1).contiguous()
    scales, opac = sc.raw_scales.exp().contiguous(), torch.sigmoid(sc.raw_opacities).contiguous()
    W, H = sc.width, sc.height
    vm, K = sc.viewmats.contiguous(), sc.Ks.contiguous()
    radii, m2, d, _, _ = m.projection_ut_3dgs_fused(means, quats, scales, opac, vm, None, K, W, H, 0.3, 0.01, 1e4, 0.0, False, 0, None, 4, None, None, None)
    _, ids, flat = m.intersect_tile(m2, radii, d, None, None, 1, 16, W // 16, H // 16, True)
    offs = m.intersect_offset(ids, 1, W // 16, H // 16)
    colors = torch.rand(1, n, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
    return [means, quats, scales, colors, opac[None].contiguous(), None, None, W, H, 16, vm, None, K, 0, None, 4, None, None, None, offs, flat]


@pytest.mark.gpu
def test_raster_staging_cache_survives_freed_and_reallocated_inputs(lfs):
    """Round-4 review: the forward -> backward staging cache of csrc/torch_ops.cpp was keyed on (address, version, numel) without holding the tensors: a caller that frees its
    inputs after the forward and allocates same-sized tensors (the caching allocator hands the same addresses back) with OTHER contents got a stale hit - the backward
    walked records packed from the old colours / opacities. The slot now holds its keyed tensors; whatever the allocator does, the backward must equal the
    self-contained ctypes path on the tensors it is GIVEN."""
    m = _mod()
    from lichtfeld_studio_amd import ops
    a = _raster_inputs(lfs, m, 3)
    a[0] = a[0].clone().requires_grad_(True)   # (the slot parks a forward only when a backward can follow: one differentiable operand requires a gradient)
    f = m.rasterize_to_pixels_from_world_3dgs_fwd(*a)
    vr, va = torch.randn_like(f[0]), torch.randn_like(f[1])
    ptrs = (a[3].data_ptr(), a[4].data_ptr())
    shape_c, shape_o = a[3].shape, a[4].shape
    a[3] = a[4] = None                      # the caller's temporaries die ...
    torch.cuda.synchronize()
    c2 = torch.rand(shape_c, device="cuda:0") * 0.5 + 0.25     # ... and same-sized ones with other contents are allocated
    o2 = torch.rand(shape_o, device="cuda:0") * 0.5 + 0.1
    a[3], a[4] = c2, o2
    print("addresses reused by the allocator:", (c2.data_ptr(), o2.data_ptr()) == ptrs, "(round 6: the slot only WATCHES the storages - they may be; the dead storage makes the key stale)")
    a = [x.detach() if isinstance(x, torch.Tensor) else x for x in a]
    f2 = ops.rasterize_to_pixels_from_world_3dgs_fwd(*a[:13], lfs.CameraModelType.PINHOLE, None, lfs.ShutterType.GLOBAL, None, None, None, a[19], a[20])   # (ctypes path: does not touch the slot)
    got = m.rasterize_to_pixels_from_world_3dgs_bwd(*a, f2[1], f2[2], vr, va)
    want = [ops.rasterize_to_pixels_from_world_3dgs_bwd(*a[:13], lfs.CameraModelType.PINHOLE, None, lfs.ShutterType.GLOBAL, None, None, None, a[19], a[20], f2[1], f2[2], vr, va)
            for _ in range(3)]
    for i, x in enumerate(got):
        noise_check(f"backward after re-allocated inputs [{i}]", rel_l2(n(x), n(want[0][i])), atomic_noise_bar(want[0][i], want[1][i], want[2][i]))


@pytest.mark.gpu
def test_raster_staging_cache_two_threads(lfs):
    """The viewer thread of the reference renders through the same wrappers while the training thread sits between its forward and its backward
    (rendering_pipeline.cpp:79; render_mutex_ covers post_backward / step only). Thread B hammers forwards of another scene; thread A's forward + backward pairs must
    keep producing the gradients of A's scene (hit or miss of the slot - both are correct), without a crash."""
    import threading
    m = _mod()
    from lichtfeld_studio_amd import ops
    a, b = _raster_inputs(lfs, m, 5), _raster_inputs(lfs, m, 6, n=2500)
    m.keep_raster_staging(True)   # plain tensors on both threads: park after every forward, so that the two threads really fight over the slot
    fa = m.rasterize_to_pixels_from_world_3dgs_fwd(*a)
    vr, va = torch.randn_like(fa[0]), torch.randn_like(fa[1])
    want = [ops.rasterize_to_pixels_from_world_3dgs_bwd(*a[:13], lfs.CameraModelType.PINHOLE, None, lfs.ShutterType.GLOBAL, None, None, None, a[19], a[20], fa[1], fa[2], vr, va)
            for _ in range(3)]
    bars = [atomic_noise_bar(want[0][i], want[1][i], want[2][i]) for i in range(5)]
    stop, errors = threading.Event(), []

    def viewer():
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                while not stop.is_set():
                    m.rasterize_to_pixels_from_world_3dgs_fwd(*b)
                s.synchronize()
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    th = threading.Thread(target=viewer)
    th.start()
    try:
        worst = [0.0] * 5
        for _ in range(60):
            f = m.rasterize_to_pixels_from_world_3dgs_fwd(*a)
            g = m.rasterize_to_pixels_from_world_3dgs_bwd(*a, f[1], f[2], vr, va)
            assert torch.equal(f[0], fa[0])
            for i, x in enumerate(g):
                worst[i] = max(worst[i], rel_l2(n(x), n(want[0][i])))
    finally:
        stop.set()
        th.join()
        m.keep_raster_staging(False)
    st = m.raster_staging_stats()
    print("staging slot under two threads:", st)
    assert st["hits"] + st["misses"] >= 60
    assert not errors, errors
    for i in range(5):
        noise_check(f"two-thread fwd/bwd [{i}]", worst[i], bars[i])


@pytest.mark.gpu
def test_raster_staging_slot_round6(lfs):
    """Round-5 review item 6 / ADVICE: (1) the thin-prism coefficients are part of the slot's key - two forwards that differ only in them, then the backward of the FIRST: a
    miss (and a forward + its own backward with thin-prism coefficients: a hit, which round 5 never allowed); (2) a forward no backward can follow (no differentiable
    operand requires a gradient) parks nothing and releases what an earlier forward parked; (3) the slot watches storages through weak references: an operand the caller
    freed after the forward makes the key stale (miss), and nothing of the caller's stays alive in the slot; (4) the explicit clear."""
    m = _mod()
    from lichtfeld_studio_amd import ops
    m.raster_staging_clear()
    a = _raster_inputs(lfs, m, 11)
    a[0] = a[0].clone().requires_grad_(True)          # means requires a gradient: the training step as the wrapper sees it from inside an autograd Function's forward
    dev = a[0].device
    thin1 = torch.tensor([[1e-3, -2e-3, 5e-4, 1e-3]], device=dev)
    thin2 = torch.tensor([[2e-3, 1e-3, -5e-4, 2e-3]], device=dev)
    radial = torch.tensor([[1e-2, -1e-3, 0.0, 0.0, 0.0, 0.0]], device=dev)
    def with_thin(t):
        b = list(a); b[16], b[18] = radial, t; return b
    s0 = m.raster_staging_stats()
    # (1a) forward + its own backward, thin-prism present: a hit
    f = m.rasterize_to_pixels_from_world_3dgs_fwd(*with_thin(thin1))
    vr, va = torch.randn_like(f[0]), torch.randn_like(f[1])
    assert m.raster_staging_stats()["parked_bytes"] > 0
    g_hit = m.rasterize_to_pixels_from_world_3dgs_bwd(*with_thin(thin1), f[1], f[2], vr, va)
    s1 = m.raster_staging_stats()
    assert s1["hits"] == s0["hits"] + 1 and s1["parked_bytes"] == 0, (s0, s1)
    # the prepared backward equals the self-contained one (ctypes path, which never touches the slot)
    b1 = with_thin(thin1)
    want = [ops.rasterize_to_pixels_from_world_3dgs_bwd(*[x.detach() if isinstance(x, torch.Tensor) else x for x in b1[:13]], lfs.CameraModelType.PINHOLE, None, lfs.ShutterType.GLOBAL,
                                                        radial, None, thin1, b1[19], b1[20], f[1], f[2], vr, va) for _ in range(3)]
    for i, x in enumerate(g_hit):
        noise_check(f"prepared backward with thin-prism coefficients [{i}]", rel_l2(n(x), n(want[0][i])), atomic_noise_bar(want[0][i], want[1][i], want[2][i]))
    # (1b) two forwards that differ only in the thin-prism coefficients, then the backward of the first: a miss
    m.rasterize_to_pixels_from_world_3dgs_fwd(*with_thin(thin1))
    m.rasterize_to_pixels_from_world_3dgs_fwd(*with_thin(thin2))
    g_miss = m.rasterize_to_pixels_from_world_3dgs_bwd(*with_thin(thin1), f[1], f[2], vr, va)
    s2 = m.raster_staging_stats()
    assert s2["misses"] == s1["misses"] + 1 and s2["hits"] == s1["hits"], (s1, s2)
    for i, x in enumerate(g_miss):
        noise_check(f"backward after a forward with OTHER thin-prism coefficients [{i}]", rel_l2(n(x), n(want[0][i])), atomic_noise_bar(want[0][i], want[1][i], want[2][i]))
    # (2) a forward of plain tensors (nothing requires a gradient): nothing parked, and what the previous forward parked is gone
    m.rasterize_to_pixels_from_world_3dgs_fwd(*a)
    assert m.raster_staging_stats()["parked_bytes"] > 0
    plain = list(a); plain[0] = a[0].detach()
    with torch.no_grad():
        m.rasterize_to_pixels_from_world_3dgs_fwd(*plain)
    s3 = m.raster_staging_stats()
    assert s3["parked_bytes"] == 0 and s3["skipped_forwards"] == s2["skipped_forwards"] + 1, (s2, s3)
    # (3) weak references: the caller frees an operand after the forward - the slot must not have kept it alive, and the backward with a same-sized replacement misses
    import gc, weakref
    c = a[3].clone()
    probe = weakref.ref(c)
    b = list(a); b[3] = c
    f3 = m.rasterize_to_pixels_from_world_3dgs_fwd(*b)
    del c; b[3] = None; gc.collect()
    assert probe() is None, "the staging slot kept the caller's colours alive"
    b[3] = a[3].clone()
    m.rasterize_to_pixels_from_world_3dgs_bwd(*b, f3[1], f3[2], torch.randn_like(f3[0]), torch.randn_like(f3[1]))
    s4 = m.raster_staging_stats()
    assert s4["misses"] == s3["misses"] + 1, (s3, s4)
    # (4) explicit clear
    m.rasterize_to_pixels_from_world_3dgs_fwd(*a)
    assert m.raster_staging_stats()["parked_bytes"] > 0
    m.raster_staging_clear()
    assert m.raster_staging_stats()["parked_bytes"] == 0
