"""GPU parity: K10-K13 (relocation, add_noise, quats_to_rotmats, fused Adam) vs the oracle."""
import math

import numpy as np
import pytest
import torch

from gpu_util import n, t

pytestmark = pytest.mark.gpu


def _binoms():
    b = np.zeros((51, 51), np.float32)   # mcmc.cpp:459-472
    for a in range(51):
        for k in range(a + 1):
            b[a, k] = math.comb(a, k)
    return b


def test_quats_to_rotmats(lfs, oracle_mod):
    from lichtfeld_studio_amd import ops
    q = np.random.default_rng(0).standard_normal((9999, 4)).astype(np.float32)
    R = n(ops.quats_to_rotmats(t(q)))
    np.testing.assert_allclose(R, oracle_mod.quats_to_rotmats(q), rtol=0, atol=1e-6)
    np.testing.assert_allclose(R @ R.transpose(0, 2, 1), np.tile(np.eye(3, dtype=np.float32), (9999, 1, 1)), atol=1e-5)


def test_relocation(lfs, oracle_mod):
    from lichtfeld_studio_amd import ops
    rng = np.random.default_rng(1)
    N = 20000
    o = (rng.random(N) * 0.98 + 0.01).astype(np.float32)
    s = (rng.random((N, 3)) * 0.2 + 1e-3).astype(np.float32)
    ratios = rng.integers(1, 52, N).astype(np.int32)
    no, ns = oracle_mod.relocation(o, s, ratios, _binoms(), 51)
    gno, gns = ops.relocation(t(o), t(s), t(ratios, torch.int32), t(_binoms()), 51)
    np.testing.assert_allclose(n(gno), no, rtol=1e-5, atol=1e-7)
    # large ratios sum alternating binomial terms (catastrophic cancellation in fp32 in the reference as well):
    # compare where the denominator is well conditioned, and require agreement of the bulk elsewhere
    small = ratios <= 12
    np.testing.assert_allclose(n(gns)[small], ns[small], rtol=2e-3)
    ok = np.isfinite(ns).all(-1)
    assert np.median(np.abs(n(gns)[ok] - ns[ok]) / (np.abs(ns[ok]) + 1e-12)) < 1e-3


def test_add_noise(lfs, oracle_mod):
    from lichtfeld_studio_amd import ops
    rng = np.random.default_rng(2)
    N = 20000
    ro = (rng.standard_normal(N) * 3).astype(np.float32)
    rs = (np.log(0.02) + 0.5 * rng.standard_normal((N, 3))).astype(np.float32)
    rq = rng.standard_normal((N, 4)).astype(np.float32)
    nz = rng.standard_normal((N, 3)).astype(np.float32)
    m = rng.standard_normal((N, 3)).astype(np.float32)
    lr = 1.6e-4 * 5e5
    want = oracle_mod.add_noise(ro, rs, rq, nz, m, lr)
    got = t(m).clone()
    assert ops.add_noise(t(ro), t(rs), t(rq), t(nz), got, lr) is None   # in place, returns void
    delta = np.abs(want - m).max()
    assert delta > 1e-3
    np.testing.assert_allclose(n(got), want, rtol=1e-5, atol=1e-6 * max(1.0, delta))


@pytest.mark.parametrize("numel,offset", [(1, 0), (1023, 0), (100003, 0), (4096, 1), (59 * 20000, 0)])
def test_adam_bit_exact(lfs, oracle_mod, numel, offset):
    """Same IEEE operations in the same order as adam_kernels.cuh:13-36 (the file is compiled with
    -ffp-contract=off): bit-exact vs the fp32 oracle, including un-aligned views."""
    from lichtfeld_studio_amd import ops
    rng = np.random.default_rng(numel)
    mk = lambda s=1.0: (rng.standard_normal(numel + offset) * s).astype(np.float32)
    p, m, v, g = mk(), mk(0.1), np.abs(mk(0.01)), mk()
    args = (1.6e-4, 0.9, 0.999, 1e-15, 1 / (1 - 0.9 ** 3), 1 / math.sqrt(1 - 0.999 ** 3))
    wp, wm, wv = oracle_mod.adam_step(p[offset:], m[offset:], v[offset:], g[offset:], *args)
    gp, gm, gv, gg = t(p), t(m), t(v), t(g)
    ops.adam_step_wrapper(gp[offset:], gm[offset:], gv[offset:], gg[offset:], *args)
    assert np.array_equal(n(gp)[offset:], wp) and np.array_equal(n(gm)[offset:], wm) and np.array_equal(n(gv)[offset:], wv)
    if offset:
        assert np.array_equal(n(gp)[:offset], p[:offset])   # nothing before the view was touched


def test_adam_multi_equals_per_tensor(lfs):
    from lichtfeld_studio_amd import ops
    torch.manual_seed(0)
    shapes = [(5000, 3), (5000, 1, 3), (5000, 15, 3), (5000, 3), (5000, 4), (5000,)]
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 5e-3, 1e-3, 5e-2]
    mk = lambda: [torch.randn(s, device="cuda:0") for s in shapes]
    p, m, v, g = mk(), [x.abs() * 0.1 for x in mk()], [x.abs() * 0.01 for x in mk()], mk()
    p2, m2, v2 = [x.clone() for x in p], [x.clone() for x in m], [x.clone() for x in v]
    bc = (1 / (1 - 0.9 ** 5), 1 / math.sqrt(1 - 0.999 ** 5))
    ops.adam_step_multi([(p[i], m[i], v[i], g[i], lrs[i], 0.9, 0.999, 1e-15, *bc) for i in range(6)])
    for i in range(6):
        ops.adam_step_wrapper(p2[i], m2[i], v2[i], g[i], lrs[i], 0.9, 0.999, 1e-15, *bc)
    for a, b in zip(p + m + v, p2 + m2 + v2):
        assert torch.equal(a, b)
