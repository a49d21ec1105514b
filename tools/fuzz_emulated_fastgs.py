#!/usr/bin/env python3
"""Differential shape fuzzer for the fastgs (EWA) rasterizer - SURVEY.md §8f row 1 - on the EMULATED product library (tests/emul_util.py; no GPU): random small
problems (1 .. 400 primitives, images 1 x 1 .. 120 x 120, SH degree 0 - 3, near / far / lateral spreads that put part or all of the model outside the frustum,
tiny to image-filling scales) through lfs_fastgs_preprocess -> render -> backward, each case against the CPU oracle (oracle/oracle_fastgs.hpp) at the bars of
tests/test_gpu_fastgs.py: instance count identical, image / alpha mean |diff| <= 2e-6 with <= 0.1 % of the pixels (at least one allowed) beyond 1/255 + 1e-4, gradients
against the fp64 oracle <= 5e-4 relative L2 after setting aside at most 3 threshold-flip rows, visibility counts of the densification statistics identical, and the
forward with cell culling on == off bit for bit.

    python tools/fuzz_emulated_fastgs.py --seconds 300 --seed 0

Results of the runs of record: profiles/r04/fuzz_emulated.txt."""
import argparse
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import emul_util  # noqa: E402
from gpu_util import rows_check  # noqa: E402

STATS = {}


def _stat(name, value):
    a = STATS.setdefault(name, [0, 0.0])
    a[0] += 1
    a[1] = max(a[1], float(value))


def _t(a, dt=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dt).contiguous()


def one_case(rng, fastgs, lib, orc, idx):
    N = int(rng.choice([1, 2, 63, 64, 65, int(rng.integers(3, 400))]))
    W, H = int(rng.integers(1, 121)), int(rng.integers(1, 121))
    deg = int(rng.integers(0, 4))
    spread = float(rng.choice([0.3, 1.2, 4.0]))
    zmode = int(rng.choice([0, 0, 0, 1, 2]))              # in front / straddling the near plane / all behind the camera
    desc = dict(idx=idx, N=N, W=W, H=H, deg=deg, spread=spread, zmode=zmode)
    means = rng.standard_normal((N, 3)) * spread
    means[:, 2] = {0: np.abs(means[:, 2]) + float(rng.choice([0.5, 3.0])), 1: means[:, 2], 2: -np.abs(means[:, 2]) - 1.0}[zmode]
    smax = float(rng.choice([0.02, 0.25, 1.5]))
    scales_raw = np.log(rng.uniform(0.005, smax, (N, 3)))
    rot_raw = rng.standard_normal((N, 4))
    opac_raw = rng.standard_normal(N) * 1.5
    sh0 = rng.standard_normal((N, 1, 3)) * 0.5
    sh_rest = rng.standard_normal((N, 15, 3)) * 0.2
    ang = float(rng.uniform(-0.3, 0.3))
    R = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
    w2c = np.eye(4); w2c[:3, :3] = R; w2c[:3, 3] = rng.standard_normal(3) * 0.2
    cam_pos = -R.T @ w2c[:3, 3]
    f = float(rng.uniform(0.5, 1.5)) * max(W, H)
    fx, fy, cx, cy = f, f * float(rng.uniform(0.9, 1.1)), W / 2 + float(rng.uniform(-0.5, 0.5)), H / 2 + float(rng.uniform(-0.5, 0.5))
    Kb = (deg + 1) ** 2
    s = fastgs.FastGSSettings(_t(cam_pos), Kb, W, H, fx, fy, cx, cy, 0.01, 1e10)
    dev = [_t(x) for x in (means, scales_raw, rot_raw, opac_raw, sh0, sh_rest, w2c)]
    image, alpha, pws, iws, n_inst = fastgs.forward_wrapper(*dev, s)
    oa = (means, scales_raw, rot_raw, opac_raw, sh0, sh_rest, w2c, cam_pos, Kb, W, H, fx, fy, cx, cy)
    f32 = orc.fastgs_forward(*oa, dtype=np.float32)
    assert n_inst == len(f32["ids"]), (desc, "instances", n_inst, len(f32["ids"]))
    assert torch.isfinite(image).all() and float(alpha.min()) >= 0.0 and float(alpha.max()) <= 1.0 + 1e-6, (desc, "forward range")
    d, da = np.abs(image.numpy() - f32["image"]), np.abs(alpha.numpy()[0] - f32["alpha"])
    _stat("forward: mean |image diff| (bar 2e-6)", d.mean())
    _stat("forward: mean |alpha diff| (bar 2e-6)", da.mean())
    beyond = int((d.max(0) > 1 / 255 + 1e-4).sum())
    _stat("forward: pixels beyond 1/255 + 1e-4", beyond)
    assert d.mean() < 2e-6 and da.mean() < 2e-6, (desc, "forward vs oracle", float(d.mean()), float(da.mean()))
    assert beyond <= max(1, int(1e-3 * W * H)), (desc, "forward flips", beyond)
    lib.lfs_set_debug_flags(1)                               # cell culling off: the same image bit for bit
    try:
        image2, alpha2, _, _, n2 = fastgs.forward_wrapper(*dev, s)
    finally:
        lib.lfs_set_debug_flags(0)
    assert n2 == n_inst and torch.equal(image, image2) and torch.equal(alpha, alpha2), (desc, "culling on / off")
    gi, ga = rng.standard_normal((3, H, W)).astype(np.float32), rng.standard_normal((1, H, W)).astype(np.float32)
    dens = torch.zeros(2, N)
    g = fastgs.backward_wrapper(dens, _t(gi), _t(ga), image, alpha, *[_t(x) for x in (means, scales_raw, rot_raw, sh0, sh_rest)], pws, iws, _t(w2c), s, n_inst)
    f64 = orc.fastgs_forward(*oa, dtype=np.float64)
    og = orc.fastgs_backward(f64, *oa, gi, ga, dtype=np.float64)
    for name, a, b in zip(["means", "scales_raw", "rot_raw", "opac_raw", "sh0", "sh_rest"], g, og[:6]):
        a = a.numpy().reshape(b.shape)
        assert np.isfinite(a).all(), (desc, name)
        if np.abs(b).max() == 0:
            assert np.abs(a).max() == 0, (desc, name, "oracle gradient is zero")
            continue
        e, flips, rest = rows_check(a.reshape(N, -1), b.reshape(N, -1), bar=5e-4, max_flips=3)
        _stat(f"backward: {name} rel-L2 vs fp64 oracle without <= 3 flip rows (bar 5e-4)", rest)
        _stat(f"backward: {name} flip rows", flips)
        assert rest < 5e-4, (desc, name, e, flips, rest)
    assert np.array_equal(dens.numpy()[0], og[6][0].astype(np.float32)), (desc, "visibility counts")
    return desc, n_inst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    import oracle as orc
    orc.build(ref=False)
    with emul_util.installed() as lib:
        from lichtfeld_studio_amd import fastgs
        t0, n, inst, empty = time.time(), 0, 0, 0
        while time.time() - t0 < a.seconds:
            _, k = one_case(rng, fastgs, lib, orc, n)
            n += 1
            inst += k
            empty += k == 0
        print(f"fuzz_emulated_fastgs: {n} cases in {time.time() - t0:.0f} s (seed {a.seed}), {inst} instances in total, {empty} cases with none; no assertion failed")
        for k, (cnt, worst) in sorted(STATS.items()):
            print(f"  {k}: {cnt} comparisons, worst {worst:.3g}")


if __name__ == "__main__":
    main()
