"""The spherical-harmonics kernels (csrc/sh.hip) as host code on the wavefront emulator (tests/emul): forward / backward against the CPU oracle for
every degree, and the LFS_SH_DPP_SUM build (the 16-lane sums as DPP row operations instead of ds_bpermute) BIT-identical to the
default build - the claim the switch rests on, checked through the real kernel code."""
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


def _build(out, defines):
    cmd = [CLANG, "-x", "c++", "-std=c++17", "-O1", "-DLFS_EMULATE", *defines, "-fPIC", "-shared", "-ffp-contract=off", "-I" + os.path.join(HERE, "emul"),
           "-Wno-unused-value", "-Wno-unknown-attributes", os.path.join(ROOT, "lichtfeld-studio_amd", "csrc", "sh.hip"), os.path.join(HERE, "emul", "emul_stubs.cpp"),
           "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return C.CDLL(out)


@pytest.fixture(scope="module")
def libs(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ to build the emulated kernels")
    d = tmp_path_factory.mktemp("emul_sh")
    import lichtfeld_studio_amd  # noqa: F401
    from lichtfeld_studio_amd import ops
    saved = {k: getattr(ops, k) for k in ("load_library", "require_gpu", "stream")}
    both = {"bpermute": _build(str(d / "a.so"), []), "dpp": _build(str(d / "b.so"), ["-DLFS_SH_DPP_SUM"])}
    ops.require_gpu = lambda *a: None
    ops.stream = lambda: None
    yield ops, both
    for k, v in saved.items():
        setattr(ops, k, v)


def _t(a, dtype=torch.float32):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dtype).contiguous()


@pytest.mark.parametrize("degree,K", [(0, 1), (1, 4), (2, 9), (3, 16), (4, 25), (1, 16)])
def test_emulated_sh_against_oracle_and_dpp_sums_bit_identical(libs, degree, K):
    ops, both = libs
    sys.path.insert(0, HERE)
    import oracle
    rng = np.random.default_rng(40 + degree + K)
    N = 333
    dirs = (rng.standard_normal((N, 3)) * rng.uniform(0.2, 6, (N, 1))).astype(np.float32)
    coeffs = rng.standard_normal((N, K, 3)).astype(np.float32)
    masks = rng.random(N) > 0.3
    vcol = rng.standard_normal((N, 3)).astype(np.float32)
    res = {}
    for name, lib in both.items():
        ops.load_library = lambda lib=lib: lib
        fwd = ops.spherical_harmonics_fwd(degree, _t(dirs), _t(coeffs), _t(masks, torch.bool))
        vc, vd = ops.spherical_harmonics_bwd(K, degree, _t(dirs), _t(coeffs), _t(masks, torch.bool), _t(vcol), True)
        res[name] = (fwd, vc, vd)
    o = oracle.spherical_harmonics_fwd(degree, dirs, coeffs, masks)
    ovc, ovd = oracle.spherical_harmonics_bwd(degree, dirs, coeffs, masks, vcol, True)
    fwd, vc, vd = res["bpermute"]
    np.testing.assert_allclose(fwd.numpy(), o, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(vc.numpy(), ovc, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(vd.numpy(), ovd, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(ovd).max()))
    for a, b in zip(res["bpermute"], res["dpp"]):
        assert torch.equal(a, b)
