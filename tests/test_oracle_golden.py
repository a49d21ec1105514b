"""CPU: the oracle (oracle/oracle_ops.hpp) against the golden vectors generated from the
reference's own CPU code (tests/torch_impl.cpp -> oracle/make_golden.py), and, when oracle/_ref
is present, against that code directly.  Tolerances are the reference's own
(tests/test_garden_data.cpp:430 SH 1e-4; :566-568 tile intersection exact)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_sh_fwd_matches_reference_golden(oracle_mod):
    g = np.load(os.path.join(GOLD, "sh_fwd.npz"))
    for deg in range(5):
        got = oracle_mod.spherical_harmonics_fwd(deg, g["dirs"], g[f"coeffs{deg}"])
        np.testing.assert_allclose(got, g[f"colors{deg}"], rtol=1e-4, atol=1e-4)
        # in practice the restatement is much closer than the reference's own tolerance
        assert np.abs(got - g[f"colors{deg}"]).max() < 2e-5


def test_quat_to_rotmat_matches_reference_golden(oracle_mod):
    g = np.load(os.path.join(GOLD, "quat_to_rotmat.npz"))
    np.testing.assert_allclose(oracle_mod.quats_to_rotmats(g["quats"]), g["rotmats"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", ["isect_c3_n1000_40x60.npz", "isect_c1_n5000_200x120.npz"])
def test_intersect_tile_exact_vs_reference_golden(oracle_mod, name):
    g = np.load(os.path.join(GOLD, name))
    C = g["means2d"].shape[0]
    ts, tw, th = int(g["tile_size"]), int(g["tile_width"]), int(g["tile_height"])
    tpg, ids, flat = oracle_mod.intersect_tile(g["means2d"], g["radii"], g["depths"], C, ts, tw, th, True)
    assert np.array_equal(tpg, g["tiles_per_gauss"])
    assert np.array_equal(ids, g["isect_ids"]) and np.array_equal(flat, g["flatten_ids"])
    _, uids, uflat = oracle_mod.intersect_tile(g["means2d"], g["radii"], g["depths"], C, ts, tw, th, False)
    assert np.array_equal(uids, g["isect_ids_unsorted"]) and np.array_equal(uflat, g["flatten_ids_unsorted"])
    # offsets: lower bound of every (camera, tile) in the sorted key list
    offs = oracle_mod.intersect_offset(ids, C, tw, th).reshape(-1)
    tile_n_bits = int(np.floor(np.log2(tw * th))) + 1
    hi = ids >> 32
    flat_tile = (hi >> tile_n_bits) * (tw * th) + (hi & ((1 << tile_n_bits) - 1))
    expect = np.searchsorted(flat_tile, np.arange(C * tw * th), side="left")
    assert np.array_equal(offs, expect)


def test_ut_projection_close_to_reference_ewa_for_small_gaussians(oracle_mod):
    """Indicative only: the reference's CPU projection is EWA, not the unscented transform; for
    small Gaussians the two agree to a fraction of a pixel (means2d) and depths are identical."""
    g = np.load(os.path.join(GOLD, "ewa_projection.npz"))
    radii, m2, d, con, _ = oracle_mod.projection_ut_3dgs_fused(
        g["means"], g["quats"], g["scales"], None, g["viewmat"][None], None, g["K"][None], 256, 256, far_plane=1e10)
    vis = (radii[0] > 0).all(-1) & (g["radii"] > 0).all(-1)
    assert vis.sum() > 1000
    assert np.abs(m2[0][vis] - g["means2d"][vis]).max() < 0.05
    np.testing.assert_allclose(d[0][vis], g["depths"][vis], rtol=1e-5, atol=1e-5)
    assert np.abs(radii[0][vis] - g["radii"][vis]).max() <= 1
    rel = np.abs(con[0][vis] - g["conics"][vis]) / (np.abs(g["conics"][vis]).max(-1, keepdims=True) + 1e-6)
    assert np.median(rel) < 2e-3


def test_against_reference_build_when_present(oracle_mod):
    """oracle/_ref (the reference's torch_impl.cpp compiled in place) — fresh random inputs."""
    if oracle_mod.ref_lib() is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box and no prebuilt .so)")
    rng = np.random.default_rng(123)
    dirs = rng.standard_normal((300, 3)).astype(np.float32)
    coeffs = rng.standard_normal((300, 16, 3)).astype(np.float32)
    np.testing.assert_allclose(oracle_mod.spherical_harmonics_fwd(3, dirs, coeffs), oracle_mod.ref_spherical_harmonics(3, dirs, coeffs), atol=2e-5)
    m = (rng.standard_normal((2, 700, 2)) * 50).astype(np.float32)
    r = rng.integers(0, 30, (2, 700, 2)).astype(np.int32)
    d = rng.random((2, 700)).astype(np.float32)
    a = oracle_mod.intersect_tile(m, r, d, 2, 16, 7, 5, True)   # T = 35: not a power of two
    b = oracle_mod.ref_isect_tiles(m, r, d, 16, 7, 5, True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
