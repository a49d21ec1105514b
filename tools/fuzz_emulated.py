#!/usr/bin/env python3
"""Shape fuzzer for the hot-path operator chain on the EMULATED product library (tests/emul_util.py; no GPU): random small problems - Gaussian counts 0 .. 400, images
from 1 x 1 to ~120 x 120, tile sizes 8 / 16 / 24 / 32 / 48, 1 - 2 cameras, 1 - 4 channels, SH degree 0 - 4, every camera model / shutter, optional masks and
backgrounds - through  projection_ut_3dgs_fused -> spherical_harmonics_fwd -> intersect_tile (sorted, with offsets) -> rasterize fwd -> rasterize bwd ->
spherical_harmonics_bwd -> adam_step.  Checked per case: outputs finite, tile lists sorted by (tile, depth), offsets consistent with the list, alpha in [0, 1],
culling on / off bit-identical in the forward, and - with the library built under AddressSanitizer - every global-memory access of every emulated lane in bounds:

    ASAN_LIB=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
    LFS_EMUL_SANITIZE=1 ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1 LD_PRELOAD=$ASAN_LIB python tools/fuzz_emulated.py --seconds 600

Results of the runs of record: profiles/r04/fuzz_emulated.txt."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import emul_util  # noqa: E402
from gpu_util import small_rotation_viewmat  # noqa: E402


def one_case(rng, lfs, ops, lib, idx):
    N = int(rng.choice([0, 1, 2, 63, 64, 65, int(rng.integers(3, 400))]))
    W, H = int(rng.integers(1, 121)), int(rng.integers(1, 121))
    ts = int(rng.choice([8, 16, 24, 32, 48]))
    Cn = int(rng.choice([1, 1, 2]))
    cdim = int(rng.choice([1, 2, 3, 3, 4]))
    model = int(rng.choice([0, 0, 0, 2]))                 # PINHOLE / FISHEYE
    shutter = int(rng.choice([4, 4, 4, 0, 1, 2, 3]))
    desc = dict(idx=idx, N=N, W=W, H=H, ts=ts, C=Cn, cdim=cdim, model=model, shutter=shutter)
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt).contiguous()
    means = rng.standard_normal((N, 3)).astype(np.float32) * float(rng.choice([0.3, 1.0, 3.0]))
    means[:, 2] = np.abs(means[:, 2]) + float(rng.choice([0.2, 1.5, 3.0]))
    quats = rng.standard_normal((N, 4)).astype(np.float32)
    smax = float(rng.choice([0.05, 0.3, 2.0]))
    scales = (rng.random((N, 3)) * smax + 1e-3).astype(np.float32)
    opac = (rng.random(N) * 0.98 + 0.01).astype(np.float32)
    vm0 = np.stack([small_rotation_viewmat(rng, 0.05 + 0.2 * c, 0.2) for c in range(Cn)]).astype(np.float32)
    vm1 = np.stack([small_rotation_viewmat(rng, 0.1, 0.3) for _ in range(Cn)]).astype(np.float32) if shutter != 4 else None
    f = float(rng.uniform(0.4, 1.5)) * max(W, H)
    K = np.tile(np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], np.float32), (Cn, 1, 1))
    rad = tan = thin = None
    if model == 2:
        rad = (rng.standard_normal((Cn, 4)) * 0.01).astype(np.float32)
    elif rng.random() < 0.3:
        rad = (rng.standard_normal((Cn, 4 if rng.random() < 0.5 else 6)) * 0.02).astype(np.float32)
        tan = (rng.standard_normal((Cn, 2)) * 0.002).astype(np.float32)
        if rng.random() < 0.5:
            thin = (rng.standard_normal((Cn, 4)) * 0.001).astype(np.float32)
    cm, st = lfs.CameraModelType(model), lfs.ShutterType(shutter)
    radii, m2, d, conics, _ = ops.projection_ut_3dgs_fused(t(means), t(quats), t(scales), t(opac), t(vm0), None if vm1 is None else t(vm1), t(K), W, H, 0.3, 0.01, 1e4, 0.0,
                                                           False, cm, None, st, None if rad is None else t(rad), None if tan is None else t(tan), None if thin is None else t(thin))
    for x in (m2, d, conics):
        assert torch.isfinite(x).all(), (desc, "projection")
    deg = int(rng.integers(0, 5))
    Kc = int(rng.choice([k for k in (1, 4, 9, 16, 25) if k >= (deg + 1) ** 2]))
    coeffs = (rng.standard_normal((N, Kc, 3)) * 0.3).astype(np.float32)
    dirs = rng.standard_normal((N, 3)).astype(np.float32)
    smask = (rng.random(N) < 0.8)
    col3 = ops.spherical_harmonics_fwd(deg, t(dirs), t(coeffs), t(smask, torch.bool) if rng.random() < 0.7 else None)
    assert torch.isfinite(col3).all(), (desc, "sh_fwd")
    tw, th = (W + ts - 1) // ts, (H + ts - 1) // ts
    tpg, ids, flat, offs = ops.intersect_tile(m2, radii, d, None, None, Cn, ts, tw, th, True, return_offsets=True)
    ids_n, flat_n, offs_n = ids.numpy(), flat.numpy(), offs.numpy().reshape(-1)
    assert int(tpg.sum()) == len(ids_n) == len(flat_n), (desc, "isect count")
    assert np.all(np.diff(ids_n) >= 0), (desc, "isect order")                      # (camera | tile | depth bits) ascending
    assert np.all(np.diff(offs_n) >= 0) and (len(offs_n) == 0 or offs_n[-1] <= len(ids_n)), (desc, "offsets")
    assert np.all((flat_n >= 0) & (flat_n < max(Cn * N, 1))), (desc, "flatten ids")
    colors = rng.random((Cn, N, cdim)).astype(np.float32)
    opacs = np.tile(opac[None], (Cn, 1)).astype(np.float32)
    bg = rng.random((Cn, cdim)).astype(np.float32) if rng.random() < 0.6 else None
    masks = (rng.random((Cn, th, tw)) > 0.25) if rng.random() < 0.3 else None
    args = (t(means), t(quats), t(scales), t(colors), t(opacs), None if bg is None else t(bg), None if masks is None else t(masks, torch.bool), W, H, ts, t(vm0),
            None if vm1 is None else t(vm1), t(K), cm, None, st, None if rad is None else t(rad), None if tan is None else t(tan), None if thin is None else t(thin), offs, flat)
    rc, ra, li = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)
    assert torch.isfinite(rc).all() and float(ra.min()) >= 0.0 and float(ra.max()) <= 1.0 + 1e-6, (desc, "raster fwd")
    lib.lfs_set_debug_flags(1)       # culling off: the cell lists are the tile lists - the same image, bit for bit
    try:
        rc2, ra2, li2 = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)
    finally:
        lib.lfs_set_debug_flags(0)
    assert torch.equal(rc, rc2) and torch.equal(ra, ra2) and torch.equal(li, li2), (desc, "culling on / off")
    v_rc, v_ra = t(rng.standard_normal(tuple(rc.shape)).astype(np.float32)), t(rng.standard_normal(tuple(ra.shape)).astype(np.float32))
    grads = ops.rasterize_to_pixels_from_world_3dgs_bwd(*args, ra, li, v_rc, v_ra)
    for g in grads:
        assert torch.isfinite(g).all(), (desc, "raster bwd")
    v_coeffs, v_dirs = ops.spherical_harmonics_bwd(Kc, deg, t(dirs), t(coeffs), None, t(rng.standard_normal((N, 3)).astype(np.float32)), True)
    assert torch.isfinite(v_coeffs).all() and torch.isfinite(v_dirs).all(), (desc, "sh_bwd")
    if N:
        p, m, v = t(means).clone(), torch.zeros(N, 3), torch.zeros(N, 3)
        ops.adam_step_wrapper(p, m, v, grads[0].reshape(-1, 3)[:N].contiguous(), 1e-3, 0.9, 0.999, 1e-15, 10.0, 31.6)
        assert torch.isfinite(p).all(), (desc, "adam")
    return desc, len(ids_n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    with emul_util.installed() as lib:
        import lichtfeld_studio_amd as lfs
        from lichtfeld_studio_amd import ops
        t0, n, isects, biggest = time.time(), 0, 0, 0
        while time.time() - t0 < a.seconds:
            desc, k = one_case(rng, lfs, ops, lib, n)
            n += 1
            isects += k
            biggest = max(biggest, k)
        print(f"fuzz_emulated: {n} cases in {time.time() - t0:.0f} s (seed {a.seed}), {isects} tile intersections walked in total, largest case {biggest}; "
              f"sanitizer {'ON' if os.environ.get('LFS_EMUL_SANITIZE') else 'off'}; no assertion failed")


if __name__ == "__main__":
    main()
