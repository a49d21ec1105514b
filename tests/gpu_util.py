"""Helpers shared by the -m gpu parity tests."""
import numpy as np
import torch

DEV = "cuda:0"


def t(a, dtype=torch.float32):
    if a is None:
        return None
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV).to(dtype).contiguous()


def n(x):
    return None if x is None else x.detach().cpu().numpy()


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def make_gaussians(rng, N, spread=1.0, zmin=3.0, smin=0.01, smax=0.06):
    means = rng.standard_normal((N, 3)).astype(np.float32) * spread
    means[:, 2] = np.abs(means[:, 2]) + zmin
    quats = rng.standard_normal((N, 4)).astype(np.float32)
    scales = (rng.random((N, 3)) * (smax - smin) + smin).astype(np.float32)
    opac = (rng.random(N) * 0.8 + 0.1).astype(np.float32)
    return means, quats, scales, opac


def pinhole_K(f, W, H, C=1):
    return np.tile(np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], np.float32), (C, 1, 1))


def small_rotation_viewmat(rng, angle=0.05, shift=0.1):
    """world->camera close to identity (random small rotation + translation)."""
    ax = rng.standard_normal(3); ax /= np.linalg.norm(ax)
    a = angle
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(a) * Kx + (1 - np.cos(a)) * Kx @ Kx
    m = np.eye(4)
    m[:3, :3] = R
    m[:3, 3] = rng.standard_normal(3) * shift
    return m.astype(np.float32)


def rel_l2_rows(a, b, drop_frac=0.0):
    """relative L2 error over rows (one row per Gaussian: trailing dim <= 4 is the component axis)
    after dropping the `drop_frac` worst rows.  An alpha-threshold flip (alpha just below / above
    1/255 at ONE pixel, __expf vs exp) changes the gradient of exactly one Gaussian by a visible
    amount; everything else agrees to ~1e-5."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    comp = a.shape[-1] if (a.ndim > 1 and a.shape[-1] <= 4) else 1
    a2, b2 = a.reshape(-1, comp), b.reshape(-1, comp)
    e = ((a2 - b2) ** 2).sum(1)
    k = int(np.ceil(drop_frac * len(e)))
    if k:
        e = np.sort(e)[:-k]
    return float(np.sqrt(e.sum()) / (np.linalg.norm(b2) + 1e-30))
