"""CPU: pins oracle/bilateral.py (the numpy restatement of the reference's bilateral-grid kernels - no reference test or
CPU implementation exists for them): identity grids reproduce the image, the analytic backward equals central differences
of the forward in float64, the TV backward is the derivative of the TV forward."""
import numpy as np

from oracle import bilateral as ob


def _case(seed, L=6, H=5, W=7, h=23, w=31):
    rng = np.random.default_rng(seed)
    grid = rng.standard_normal((12, L, H, W))
    rgb = rng.random((h, w, 3)) * 0.96 + 0.02
    return grid, rgb, rng


def test_identity_grid_is_identity_map():
    L, H, W = 8, 16, 16
    eye = np.eye(4)[:3].reshape(12)
    grid = np.broadcast_to(eye[:, None, None, None], (12, L, H, W)).copy()
    rgb = np.random.default_rng(0).random((40, 56, 3))
    out = ob.slice_forward(grid, rgb, np.float64)
    assert np.abs(out - rgb).max() < 1e-12


def test_slice_backward_is_gradient_of_forward():
    grid, rgb, rng = _case(1)
    go = rng.standard_normal(rgb.shape)
    gg, gr = ob.slice_backward(grid, rgb, go, np.float64)
    f = lambda g, c: float((ob.slice_forward(g, c, np.float64) * go).sum())
    eps = 1e-6
    for _ in range(40):     # d/dgrid
        idx = tuple(rng.integers(0, s) for s in grid.shape)
        gp, gm = grid.copy(), grid.copy()
        gp[idx] += eps; gm[idx] -= eps
        num = (f(gp, rgb) - f(gm, rgb)) / (2 * eps)
        assert abs(num - gg[idx]) <= 1e-6 * max(1.0, abs(num)), (idx, num, gg[idx])
    z = (0.299 * rgb[..., 0] + 0.587 * rgb[..., 1] + 0.114 * rgb[..., 2]) * (grid.shape[1] - 1)
    checked = 0
    for _ in range(200):    # d/drgb, away from the z-cell boundaries where the forward is only C0
        i, j, k = rng.integers(0, rgb.shape[0]), rng.integers(0, rgb.shape[1]), rng.integers(0, 3)
        if abs(z[i, j] - np.round(z[i, j])) < 1e-3:
            continue
        cp, cm = rgb.copy(), rgb.copy()
        cp[i, j, k] += eps; cm[i, j, k] -= eps
        num = (f(grid, cp) - f(grid, cm)) / (2 * eps)
        assert abs(num - gr[i, j, k]) <= 1e-5 * max(1.0, abs(num)), (i, j, k, num, gr[i, j, k])
        checked += 1
    assert checked > 100


def test_tv_backward_is_gradient_of_forward():
    rng = np.random.default_rng(2)
    grids = rng.standard_normal((3, 12, 4, 5, 6))
    g = ob.tv_backward(grids, 0.7, np.float64)
    eps = 1e-6
    for _ in range(40):
        idx = tuple(rng.integers(0, s) for s in grids.shape)
        gp, gm = grids.copy(), grids.copy()
        gp[idx] += eps; gm[idx] -= eps
        num = 0.7 * (ob.tv_forward(gp, np.float64) - ob.tv_forward(gm, np.float64)) / (2 * eps)
        assert abs(num - g[idx]) < 1e-8 + 1e-6 * abs(num)
    const = np.ones((2, 12, 3, 3, 3))
    assert ob.tv_forward(const, np.float64) == 0 and np.abs(ob.tv_backward(const, 1.0, np.float64)).max() == 0
