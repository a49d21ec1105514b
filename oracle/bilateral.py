"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): numpy restatement of the reference's bilateral-grid kernels.
  slice_forward  : src/training/kernels/bilateral_grid_forward.cu:13-94
  slice_backward : src/training/kernels/bilateral_grid_backward.cu:14-153
  tv_forward     : src/training/kernels/bilateral_grid_tv.cu:12-77
  tv_backward    : src/training/kernels/bilateral_grid_tv.cu:80-135
Pinned to the reference's own kernels: those four .cu files are run on the CPU under oracle/ref_emul/ (make -C oracle refk_loss), their outputs are committed
as tests/golden/refk_loss.npz, and tests/test_oracle_refk_loss_golden.py holds this restatement to them (the reference's own tests do not touch these kernels).
tests/test_oracle_bilateral.py adds: identity grid == identity map, analytic backward == central differences of the forward in float64 away from the z-cell
boundaries, TV backward == d(TV forward).
"""
import numpy as np

LUMA = (0.299, 0.587, 0.114)


def _taps(L, H, W, rgb, dtype):
    h, w, _ = rgb.shape
    f = dtype
    wi, hi = np.meshgrid(np.arange(w), np.arange(h))
    x = (wi.astype(f) / f(w - 1)) * f(W - 1)
    y = (hi.astype(f) / f(h - 1)) * f(H - 1)
    gz = f(LUMA[0]) * rgb[..., 0] + f(LUMA[1]) * rgb[..., 1] + f(LUMA[2]) * rgb[..., 2]
    z = gz * f(L - 1)
    x0, y0, z0 = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64), np.floor(z).astype(np.int64)
    x1, y1 = np.minimum(x0 + 1, W - 1), np.minimum(y0 + 1, H - 1)
    z1 = np.minimum(np.maximum(z0 + 1, 0), L - 1)
    z0 = np.minimum(np.maximum(z0, 0), L - 1)   # upper clamp: memory safety only (the reference would read out of bounds)
    fx, fy, fz = x - x0.astype(f), y - y0.astype(f), z - z0.astype(f)
    wts = [(1 - fx) * (1 - fy) * (1 - fz), fx * (1 - fy) * (1 - fz), (1 - fx) * fy * (1 - fz), fx * fy * (1 - fz),
           (1 - fx) * (1 - fy) * fz, fx * (1 - fy) * fz, (1 - fx) * fy * fz, fx * fy * fz]
    cx, cy, cz = [x0, x1, x0, x1, x0, x1, x0, x1], [y0, y0, y1, y1, y0, y0, y1, y1], [z0, z0, z0, z0, z1, z1, z1, z1]
    return wts, cx, cy, cz, (fx, fy, fz), z, z0, z1


def slice_forward(grid, rgb, dtype=np.float32):
    grid, rgb = grid.astype(dtype), rgb.astype(dtype)
    _, L, H, W = grid.shape
    wts, cx, cy, cz, _, _, _, _ = _taps(L, H, W, rgb, dtype)
    out = np.zeros_like(rgb)
    coeff = [rgb[..., 0], rgb[..., 1], rgb[..., 2], np.ones_like(rgb[..., 0])]
    for ci in range(12):
        val = np.zeros(rgb.shape[:2], dtype)
        for k in range(8):
            val = val + grid[ci][cz[k], cy[k], cx[k]] * wts[k]
        out[..., ci // 4] += val * coeff[ci % 4]
    return out


def slice_backward(grid, rgb, grad_out, dtype=np.float32):
    grid, rgb, grad_out = grid.astype(dtype), rgb.astype(dtype), grad_out.astype(dtype)
    _, L, H, W = grid.shape
    wts, cx, cy, cz, (fx, fy, fz), z, z0, z1 = _taps(L, H, W, rgb, dtype)
    coeff = [rgb[..., 0], rgb[..., 1], rgb[..., 2], np.ones_like(rgb[..., 0])]
    grad_grid = np.zeros_like(grid)
    v = [np.zeros(rgb.shape[:2], dtype) for _ in range(3)]
    tri = [np.zeros(rgb.shape[:2], dtype) for _ in range(8)]
    for ci in range(12):
        si, di = ci % 4, ci // 4
        gw = coeff[si] * grad_out[..., di]
        for k in range(8):
            np.add.at(grad_grid[ci], (cz[k], cy[k], cx[k]), wts[k] * gw)
            val = grid[ci][cz[k], cy[k], cx[k]]
            if si < 3:
                v[si] = v[si] + val * wts[k] * grad_out[..., di]
            tri[k] = tri[k] + val * gw
    dwdz = [-(1 - fx) * (1 - fy), -fx * (1 - fy), -(1 - fx) * fy, -fx * fy, (1 - fx) * (1 - fy), fx * (1 - fy), (1 - fx) * fy, fx * fy]
    gz = np.zeros(rgb.shape[:2], dtype)
    for k in range(8):
        gz = gz + dwdz[k] * dtype(L - 1) * tri[k]
    gz = gz * ((z0.astype(dtype) != z) & (z1.astype(dtype) != z)).astype(dtype)
    grad_rgb = np.stack([v[k] + dtype(LUMA[k]) * gz for k in range(3)], -1)
    return grad_grid, grad_rgb


def tv_forward(grids, dtype=np.float32):
    g = grids.astype(dtype)
    N, _, L, H, W = g.shape
    s = dtype(0)
    if W > 1:
        s += (np.diff(g, axis=4) ** 2).sum(dtype=dtype) / dtype(L * H * (W - 1))
    if H > 1:
        s += (np.diff(g, axis=3) ** 2).sum(dtype=dtype) / dtype(L * (H - 1) * W)
    if L > 1:
        s += (np.diff(g, axis=2) ** 2).sum(dtype=dtype) / dtype((L - 1) * H * W)
    return s / dtype(12 * N)


def tv_backward(grids, grad_output, dtype=np.float32):
    g = grids.astype(dtype)
    N, _, L, H, W = g.shape
    s = dtype(grad_output) / dtype(6 * N)
    out = np.zeros_like(g)
    for axis, n, scale in ((4, W, s / dtype(max(L * H * (W - 1), 1))), (3, H, s / dtype(max(L * (H - 1) * W, 1))), (2, L, s / dtype(max((L - 1) * H * W, 1)))):
        if n < 2:
            continue
        d = np.diff(g, axis=axis) * scale
        lo = [slice(None)] * 5
        hi = [slice(None)] * 5
        lo[axis], hi[axis] = slice(0, n - 1), slice(1, n)
        out[tuple(hi)] += d
        out[tuple(lo)] -= d
    return out
