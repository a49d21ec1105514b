"""CPU: internal consistency of the oracle for the ops the reference does not pin
(SURVEY.md §8c "parity unpinned"): finite differences of the forward restatement validate the
backward restatement (recipe of tests/test_numerical_gradients.cpp:33-81), float vs double, and
Adam vs torch.optim.Adam."""
import math

import numpy as np
import pytest
import torch


def _scene(rng, N=60, W=48, H=32):
    means = rng.standard_normal((N, 3)) * 0.6
    means[:, 2] = np.abs(means[:, 2]) + 3.0
    quats = rng.standard_normal((N, 4))
    scales = rng.random((N, 3)) * 0.15 + 0.05
    opac = rng.random(N) * 0.7 + 0.2
    colors = rng.random((1, N, 3))
    vm = np.eye(4)[None]
    K = np.array([[[40.0, 0, W / 2], [0, 40.0, H / 2], [0, 0, 1]]])
    return means, quats, scales, opac, colors, vm, K, W, H


def _render(o, means, quats, scales, opac, colors, vm, K, W, H, bg, fixed=None, dtype=np.float64):
    """forward with a FROZEN tile list (as in the backward pass); returns outputs + lists."""
    if fixed is None:
        radii, m2, d, _, _ = o.projection_ut_3dgs_fused(means, quats, scales, opac, vm, None, K, W, H, dtype=dtype)
        tw, th = (W + 15) // 16, (H + 15) // 16
        _, ids, flat = o.intersect_tile(m2.astype(np.float32), radii, d.astype(np.float32), 1, 16, tw, th, True)
        offs = o.intersect_offset(ids, 1, tw, th)
        fixed = (offs, flat)
    offs, flat = fixed
    rc, ra, li = o.rasterize_fwd(means, quats, scales, colors, opac[None], bg, None, W, H, 16, vm, None, K, 0, 4, None, None, None,
                                 offs, flat, dtype=dtype)
    return rc, ra, li, fixed


def test_raster_bwd_matches_finite_differences_double(oracle_mod):
    o = oracle_mod
    rng = np.random.default_rng(5)
    means, quats, scales, opac, colors, vm, K, W, H = _scene(rng)
    bg = np.array([[0.2, 0.1, 0.4]])
    rc, ra, li, fixed = _render(o, means, quats, scales, opac, colors, vm, K, W, H, bg)
    assert (ra > 0.05).mean() > 0.2  # the scene actually covers pixels
    v_rc, v_ra = rng.standard_normal(rc.shape), rng.standard_normal(ra.shape)
    g = o.rasterize_bwd(means, quats, scales, colors, opac[None], bg, None, W, H, 16, vm, None, K, 0, 4, None, None, None,
                        fixed[0], fixed[1], ra, li, v_rc, v_ra, dtype=np.float64)

    def loss(m, q, s, op, c):
        r, a, _, _ = _render(o, m, q, s, op, c, vm, K, W, H, bg, fixed)
        return float((r * v_rc).sum() + (a * v_ra).sum())

    eps = 1e-6
    checks = 0
    for name, arr, grad in [("means", means, g[0]), ("quats", quats, g[1]), ("scales", scales, g[2]), ("opac", opac, g[4][0]),
                            ("colors", colors, g[3])]:
        flat_idx = rng.choice(arr.size, size=min(12, arr.size), replace=False)
        for fi in flat_idx:
            idx = np.unravel_index(fi, arr.shape)
            args = dict(m=means, q=quats, s=scales, op=opac, c=colors)
            key = {"means": "m", "quats": "q", "scales": "s", "opac": "op", "colors": "c"}[name]
            p, m_ = arr.copy(), arr.copy()
            p[idx] += eps; m_[idx] -= eps
            fd = (loss(**{**args, key: p}) - loss(**{**args, key: m_})) / (2 * eps)
            an = grad[idx]
            # discontinuities (alpha threshold / early termination) are measure-zero but can hit a probe
            if abs(fd - an) <= 1e-3 * max(1.0, abs(an)) + 1e-4:
                checks += 1
    assert checks >= 55, f"only {checks}/60 finite-difference probes agree"


def test_sh_bwd_matches_finite_differences_double(oracle_mod):
    o = oracle_mod
    rng = np.random.default_rng(6)
    N = 20
    for deg in range(5):
        K = (deg + 1) ** 2
        dirs, coeffs = rng.standard_normal((N, 3)), rng.standard_normal((N, K, 3))
        v = rng.standard_normal((N, 3))
        vc, vd = o.spherical_harmonics_bwd(deg, dirs, coeffs, None, v, True, dtype=np.float64)
        f = lambda d, c: float((o.spherical_harmonics_fwd(deg, d, c, dtype=np.float64) * v).sum())
        eps = 1e-6
        for _ in range(10):
            i, a = rng.integers(N), rng.integers(3)
            p, m = dirs.copy(), dirs.copy(); p[i, a] += eps; m[i, a] -= eps
            assert abs((f(p, coeffs) - f(m, coeffs)) / (2 * eps) - vd[i, a]) < 1e-6 * max(1, abs(vd[i, a])) + 1e-7
            k = rng.integers(K)
            p, m = coeffs.copy(), coeffs.copy(); p[i, k, a] += eps; m[i, k, a] -= eps
            assert abs((f(dirs, p) - f(dirs, m)) / (2 * eps) - vc[i, k, a]) < 1e-6


def test_float_and_double_oracle_agree(oracle_mod):
    o = oracle_mod
    rng = np.random.default_rng(8)
    means, quats, scales, opac, colors, vm, K, W, H = _scene(rng, N=300, W=96, H=64)
    bg = None
    r64 = _render(o, means, quats, scales, opac, colors, vm, K, W, H, bg, dtype=np.float64)
    r32 = o.rasterize_fwd(means, quats, scales, colors, opac[None], bg, None, W, H, 16, vm, None, K, 0, 4, None, None, None,
                          r64[3][0], r64[3][1], dtype=np.float32)
    assert np.abs(r32[0] - r64[0]).mean() < 1e-6
    assert (r32[2] == r64[2]).mean() > 0.999


def test_adam_matches_torch_adam(oracle_mod):
    """adam_kernels.cuh:13-36 is torch::optim::Adam without amsgrad / weight decay."""
    rng = np.random.default_rng(9)
    p0 = rng.standard_normal(1000).astype(np.float32)
    tp = torch.nn.Parameter(torch.tensor(p0.copy(), dtype=torch.float64))
    opt = torch.optim.Adam([tp], lr=1e-2, betas=(0.9, 0.999), eps=1e-15)
    p, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    for t in range(1, 6):
        g = rng.standard_normal(1000).astype(np.float32)
        tp.grad = torch.tensor(g, dtype=torch.float64)
        opt.step()
        p, m, v = oracle_mod.adam_step(p, m, v, g, 1e-2, 0.9, 0.999, 1e-15, 1 / (1 - 0.9 ** t), 1 / math.sqrt(1 - 0.999 ** t))
        np.testing.assert_allclose(p, tp.detach().numpy(), rtol=2e-6, atol=2e-6)


def test_relocation_ranges(oracle_mod):
    """range sanity of tests/test_gsplat_ops.cpp:19-63: new opacity in (0, opacity], finite scales."""
    rng = np.random.default_rng(10)
    N = 500
    o = (rng.random(N) * 0.9 + 0.05).astype(np.float32)
    s = (rng.random((N, 3)) * 0.1 + 0.01).astype(np.float32)
    ratios = rng.integers(1, 11, N).astype(np.int32)
    binoms = np.zeros((51, 51), np.float32)
    for a in range(51):
        for b in range(a + 1):
            binoms[a, b] = math.comb(a, b)
    no, ns = oracle_mod.relocation(o, s, ratios, binoms, 51)
    assert np.all(no > 0) and np.all(no <= o + 1e-6) and np.all(np.isfinite(ns)) and np.all(ns > 0)
    one = ratios == 1
    np.testing.assert_allclose(no[one], o[one], rtol=1e-6)
    np.testing.assert_allclose(ns[one], s[one], rtol=1e-5)
