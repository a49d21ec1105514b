"""The C++ libtorch drop-in layer (csrc/torch_ops.cpp: the reference's `gsplat::` / `fast_gs::optimizer`
signatures over the C ABI), exercised through its pybind module: identical results to the Python
mirror (both call the same C entry points), reference error behaviour (c10::Error on CPU tensors)."""
import math

import numpy as np
import pytest
import torch


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
                 "adam_step_wrapper"]:
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
    for x, y in zip(b1, b2):     # float atomics: same math, order-dependent rounding
        assert float((x - y).norm() / (y.norm() + 1e-20)) < 1e-5
    assert torch.equal(m.quats_to_rotmats(quats), ops.quats_to_rotmats(quats))
    p = torch.randn(10007, device=dev); e1, e2, g = torch.zeros_like(p), torch.zeros_like(p), torch.randn_like(p)
    p2, f1_, f2_ = p.clone(), e1.clone(), e2.clone()
    m.adam_step_wrapper(p, e1, e2, g, 1e-3, 0.9, 0.999, 1e-15, 10.0, 1 / math.sqrt(1 - 0.999))
    ops.adam_step_wrapper(p2, f1_, f2_, g, 1e-3, 0.9, 0.999, 1e-15, 10.0, 1 / math.sqrt(1 - 0.999))
    assert torch.equal(p, p2) and torch.equal(e1, f1_) and torch.equal(e2, f2_)
    with pytest.raises(RuntimeError, match="Unsupported number of channels"):
        m.rasterize_to_pixels_from_world_3dgs_fwd(means, quats, scales, torch.rand(1, 5000, 7, device=dev), opac[None].contiguous(), None, None, W, H, 16,
                                                  vm, None, K, 0, None, 4, None, None, None, offs, t1[2])
