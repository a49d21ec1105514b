"""Helpers shared by the -m gpu parity tests."""
import json
import os

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


_NOISE_LOG = os.environ.get("LFS_NOISE_LOG", "")


def noise_check(label, value, bar):
    """Assert `value < bar` for a NOISE-LIMITED comparison (one operand is a float-atomic sum whose own run-to-run rounding is the floor of the
    difference) and print / log value : bar.  LFS_NOISE_LOG=<file> appends one JSON line per check, so a batch of full-suite runs shows how far below
    its bar every such assertion sits (tools/r4_suite_runs.sh -> profiles/r04/noise_ratios.txt). A bar of this kind is never an absolute constant on
    the noise floor: callers pass max(floor, 5 x measured run-to-run noise)."""
    value, bar = float(value), float(bar)
    print(f"noise-limited {label}: {value:.3e} / bar {bar:.3e} = {value / bar:.3f}")
    if _NOISE_LOG:
        with open(_NOISE_LOG, "a") as f:
            f.write(json.dumps({"label": label, "value": value, "bar": bar, "ratio": value / bar}) + "\n")
    assert value < bar, (label, value, bar)


def noise_allclose(label, x, y, rtol, atol):
    """torch.allclose(x, y, rtol, atol) for a noise-limited elementwise comparison, reported as max |x-y| / (atol + rtol |y|) against 1."""
    x, y = x.detach().double().reshape(-1), y.detach().double().reshape(-1)
    assert x.numel() == y.numel() and torch.isfinite(x).all() and torch.isfinite(y).all(), label
    r = float(((x - y).abs() / (atol + rtol * y.abs())).max()) if x.numel() else 0.0
    noise_check(label, r, 1.0)


def atomic_noise_bar(*draws, floor=2e-5, k=8.0):
    """Bar for comparing something against ONE draw of a float-atomic result: max(floor, k x the largest relative L2 distance between the given draws of it)
    (>= 2 draws; ill-conditioned inputs - needle Gaussians - have a heavy-tailed noise, so callers pass three)."""
    d = [np.asarray(n(x) if torch.is_tensor(x) else x, np.float64) for x in draws]
    assert len(d) >= 2
    noise = max(float(np.linalg.norm(d[i] - d[j]) / (np.linalg.norm(d[j]) + 1e-30)) for i in range(len(d)) for j in range(i + 1, len(d)))
    return max(floor, k * noise)


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


def rows_check(a, ref, bar=2e-4, max_flips=4):
    """-> (relative L2 over all rows, k, relative L2 without the k worst rows); one row per Gaussian. k = the smallest number (<= max_flips) of
    rows that has to be set aside for the rest to agree to `bar` (k = max_flips if even that is not enough - the caller asserts on the third
    value). A "flip row": at one pixel of that Gaussian alpha is within an ulp of 1/255 or of the 0.999 cap (the geometry gradient is switched
    off above the cap, Bwd.cu:318) and the GPU's v_exp_f32 and the oracle's exp land on different sides. Flip rows are COUNTED - a handful out
    of up to 3 M - never dropped by fraction."""
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    a2, r2 = a.reshape(a.shape[0], -1), ref.reshape(ref.shape[0], -1)
    err = ((a2 - r2) ** 2).sum(1)
    norm = np.sqrt((r2 ** 2).sum()) + 1e-30
    total = float(np.sqrt(err.sum()) / norm)
    worst = np.sort(np.partition(err, -max_flips)[-max_flips:])[::-1] if len(err) > max_flips else np.sort(err)[::-1]
    s, k = err.sum(), 0
    while np.sqrt(max(s, 0.0)) / norm >= bar and k < min(max_flips, len(worst)):
        s -= worst[k]; k += 1
    return total, k, float(np.sqrt(max(s, 0.0)) / norm)
