"""The world-space rasterizer's HIP kernels compiled as HOST code on top of a wavefront emulator (tests/emul/hip/hip_runtime.h: 64-lane
wavefronts of fibers, cross-lane operations in lock-step) and driven through the same C ABI and the same Python wrappers as on the GPU.
Checks the kernels' LOGIC without a GPU: the default kernels against the CPU oracle, and every alternative code path against the default
one - culling off, the two-pixels-per-lane kernels, the quadrant-row kernels: forward BIT-identical (they perform the same per-pixel
operation sequence; the emulator's arithmetic is host float, so this is equality between emulated runs, not with the GPU), backward equal
up to summation order. What it cannot check is ISA-level behaviour (DPP encodings, hazards, inline asm): tests/test_gpu_raster*.py.
Memory-safety run of the kernels (AddressSanitizer sees every global-memory access of the emulated lanes; tensors are exact-size heap blocks):
    LFS_EMUL_SANITIZE=1 ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 \
    LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so) python -m pytest tests/test_emulated_raster.py"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ to build the emulated rasterizer")
    out = str(tmp_path_factory.mktemp("emul") / "liblfs_raster_emul.so")
    cmd = [CLANG, "-x", "c++", "-std=c++17", "-O1", "-DLFS_EMULATE", "-fPIC", "-shared", "-ffp-contract=on", "-I" + os.path.join(HERE, "emul"),
           "-Wno-unused-value", "-Wno-unknown-attributes", os.path.join(ROOT, "lichtfeld-studio_amd", "csrc", "raster.hip"),
           os.path.join(HERE, "emul", "emul_stubs.cpp"), "-o", out]
    if os.environ.get("LFS_EMUL_SANITIZE"):  # memory-safety run of the kernels (see the module docstring)
        cmd[1:1] = ["-fsanitize=address", "-shared-libasan", "-fno-omit-frame-pointer", "-g"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lib = C.CDLL(out)
    lib.lfs_rasterize_workspace_bytes.restype = C.c_size_t
    import lichtfeld_studio_amd as lfs
    from lichtfeld_studio_amd import ops
    saved = {k: getattr(ops, k) for k in ("load_library", "require_gpu", "stream", "workspace")}
    ops.load_library = lambda: lib
    ops.require_gpu = lambda *a: None
    ops.stream = lambda: None
    ops.workspace = lambda nbytes, dev, tag: torch.zeros(max(int(nbytes), 256), dtype=torch.uint8)
    yield lfs, ops, lib
    for k, v in saved.items():
        setattr(ops, k, v)


def _t(a, dtype=torch.float32):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dtype).contiguous()


def _flag(lib, flag, fn):
    try:
        lib.lfs_set_debug_flags(flag)
        return fn()
    finally:
        lib.lfs_set_debug_flags(0)


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


CASES = {
    "rgb_ragged": dict(),
    "no_background_ts8": dict(bg=False, ts=8),
    "one_channel": dict(cdim=1),
    "four_channels_two_cameras": dict(cdim=4, C=2),
    "tile_masks_ts32": dict(masks=True, ts=32, W=72, H=40),
    "dense_early_termination": dict(spread=0.4, smin=0.05, smax=0.3),
    "rolling_shutter": dict(shutter=3),
    "global_shutter": dict(shutter=4),              # (the cases above run the ROLLING-shutter kernels on one pose, shutter 0; these two the global-shutter kernels: the rotated records, the symmetric accumulators)
    "global_shutter_dense": dict(shutter=4, spread=0.4, smin=0.05, smax=0.3),
}


@pytest.mark.parametrize("case", list(CASES))
def test_emulated_kernels_agree_with_the_oracle_and_with_each_other(emu, case):
    lfs, ops, lib = emu
    sys.path.insert(0, HERE)
    import oracle
    from gpu_util import make_gaussians, pinhole_K, small_rotation_viewmat
    kw = dict(N=260, W=52, H=44, ts=16, C=1, cdim=3, bg=True, masks=False, shutter=0, spread=1.0, smin=0.01, smax=0.06)
    kw.update(CASES[case])
    rng = np.random.default_rng(sum(map(ord, case)))
    N, W, H, ts, Cn, cdim = kw["N"], kw["W"], kw["H"], kw["ts"], kw["C"], kw["cdim"]
    means, quats, scales, opac = make_gaussians(rng, N, spread=kw["spread"], smin=kw["smin"], smax=kw["smax"])
    vm0 = np.stack([small_rotation_viewmat(rng, 0.05 + 0.1 * c, 0.1) for c in range(Cn)])
    vm1 = small_rotation_viewmat(rng, 0.12, 0.2)[None] if kw["shutter"] in (1, 2, 3) else None
    K = pinhole_K(0.8 * W, W, H, Cn)
    colors = rng.random((Cn, N, cdim)).astype(np.float32)
    opacs = np.tile(opac[None], (Cn, 1)).astype(np.float32)
    bgc = rng.random((Cn, cdim)).astype(np.float32) if kw["bg"] else None
    tw, th = (W + ts - 1) // ts, (H + ts - 1) // ts
    masks = (rng.random((Cn, th, tw)) > 0.3) if kw["masks"] else None
    radii, m2, d, _, _ = oracle.projection_ut_3dgs_fused(means, quats, scales, opac, vm0, vm1, K, W, H, camera_model=0, rs_type=kw["shutter"],
                                                         radial_coeffs=None, tangential_coeffs=None, thin_prism_coeffs=None)
    _, ids, flat = oracle.intersect_tile(m2, radii, d, Cn, ts, tw, th, True)
    offs = oracle.intersect_offset(ids, Cn, tw, th)
    assert len(flat) > 0
    args_o = (means, quats, scales, colors, opacs, bgc, masks, W, H, ts, vm0, vm1, K, 0, kw["shutter"], None, None, None, offs, flat)
    o_rc, o_ra, o_li = oracle.rasterize_fwd(*args_o)
    args = (_t(means), _t(quats), _t(scales), _t(colors), _t(opacs), _t(bgc), _t(masks, torch.bool), W, H, ts, _t(vm0), _t(vm1), _t(K),
            lfs.CameraModelType.PINHOLE, None, lfs.ShutterType(kw["shutter"]), None, None, None, _t(offs, torch.int32), _t(flat, torch.int32))
    fwd = lambda: ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)
    rc, ra, li = fwd()
    # 1. default kernels vs the oracle
    assert np.abs(rc.numpy() - o_rc).mean() < 2e-6 and np.abs(ra.numpy() - o_ra).mean() < 2e-6
    assert (li.numpy() == o_li).mean() > 0.995 and (o_ra > 0.05).mean() > 0.03
    v_rc, v_ra = rng.standard_normal(o_rc.shape).astype(np.float32), rng.standard_normal(o_ra.shape).astype(np.float32)
    og = oracle.rasterize_bwd(*args_o, o_ra, o_li, v_rc, v_ra)
    bwd = lambda: ops.rasterize_to_pixels_from_world_3dgs_bwd(*args, _t(o_ra), _t(o_li, torch.int32), _t(v_rc), _t(v_ra))
    g = bwd()
    for name, a, b in zip(["v_means", "v_quats", "v_scales", "v_colors", "v_opacities"], g, og):
        assert torch.isfinite(a).all() and _rel(a, torch.from_numpy(np.asarray(b))) < 2e-3, (name, _rel(a, torch.from_numpy(np.asarray(b))))
    # 2. the alternative paths vs the default kernels
    paths = [(1, "culling off")]
    for flag, what in paths:
        f_rc, f_ra, f_li = _flag(lib, flag, fwd)
        assert torch.equal(rc, f_rc) and torch.equal(ra, f_ra) and torch.equal(li, f_li), what
        fg = _flag(lib, flag, bwd)
        for name, a, b in zip(["v_means", "v_quats", "v_scales", "v_colors", "v_opacities"], fg, g):
            assert _rel(a, b) < 2e-5, (what, name, _rel(a, b))
