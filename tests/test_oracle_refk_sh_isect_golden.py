"""CPU: the oracle's spherical harmonics (oracle_ops.hpp; forward + vjp) and tile intersection (count, key layout, stable depth sort, offsets) against golden
vectors generated from the REFERENCE'S OWN KERNELS run on the CPU (tests/golden/refk_sh_isect.npz, oracle/make_golden_refk_sh_isect.py:
gsplat/SphericalHarmonicsCUDA.cu and IntersectTile.cu under oracle/ref_emul/). Round 1 pinned these two stages to the reference's CPU code tests/torch_impl.cpp;
this pins them to the device kernels themselves, which completes K1 - K13 of SURVEY.md §8a. Integers bit-exact; SH colours / gradients 2e-6 relative to the
tensor maximum (IEEE fp32 on both sides, different association in the polynomial evaluation)."""
import os

import numpy as np
import pytest

import oracle
import refk_sh_isect_util as U

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", U.GOLD))


def close(got, want, rel, what):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, what
    nan = np.isnan(want)               # the zero direction (row 5): x / |x| is NaN in the reference kernel, and has to be in whatever is compared with it
    assert np.array_equal(np.isnan(got), nan), what
    got, want = got[~nan], want[~nan]
    assert np.abs(got - want).max() <= rel * max(np.abs(want).max(), 1e-30), (what, np.abs(got - want).max(), np.abs(want).max())


@pytest.mark.parametrize("name", sorted(U.SH_CASES))
def test_oracle_sh_matches_the_reference_kernels(oracle_mod, name):
    c, g = U.SH_CASES[name], lambda k: GOLD[f"sh/{name}/{k}"]
    dirs, coeffs, masks, v = U.sh_inputs(c)
    col = oracle_mod.spherical_harmonics_fwd(c["degree"], dirs, coeffs, masks)
    close(np.where(masks[:, None], col, 0), g("colors"), 2e-6, "colors")
    close(oracle_mod.spherical_harmonics_fwd(c["degree"], dirs, coeffs, None), g("colors_unmasked"), 2e-6, "colors without mask")
    v_coeffs, v_dirs = oracle_mod.spherical_harmonics_bwd(c["degree"], dirs, coeffs, masks, v, True)
    close(v_coeffs, g("v_coeffs"), 2e-6, "v_coeffs")
    assert np.abs(g("v_coeffs")[~masks]).max() == 0 and np.nanmax(np.abs(g("v_coeffs")[:, (c["degree"] + 1) ** 2:]), initial=0) == 0     # masked rows / unused degrees
    close(v_dirs, g("v_dirs"), 5e-6, "v_dirs")


@pytest.mark.parametrize("name", sorted(U.ISECT_CASES))
def test_oracle_intersection_matches_the_reference_kernels(oracle_mod, name):
    c, g = U.ISECT_CASES[name], lambda k: GOLD[f"isect/{name}/{k}"]
    m, r, d = U.isect_inputs(c)
    tw, th = (c["W"] + c["tile"] - 1) // c["tile"], (c["H"] + c["tile"] - 1) // c["tile"]
    tpg, ids, flat = oracle_mod.intersect_tile(m, r, d, c["C"], c["tile"], tw, th, True)
    assert np.array_equal(tpg, g("tiles_per_gauss")) and np.array_equal(ids, g("isect_ids")) and np.array_equal(flat, g("flatten_ids"))
    _, ids_u, flat_u = oracle_mod.intersect_tile(m, r, d, c["C"], c["tile"], tw, th, False)
    assert np.array_equal(ids_u, g("isect_ids_unsorted")) and np.array_equal(flat_u, g("flatten_ids_unsorted"))
    assert np.array_equal(oracle_mod.intersect_offset(ids, c["C"], tw, th), g("offsets"))
    if len(ids):   # the file itself: sorted by key, ties in flatten order (a stable sort), offsets delimit the tiles
        assert (np.diff(g("isect_ids")) >= 0).all()
        same = np.diff(g("isect_ids")) == 0
        assert (np.diff(g("flatten_ids"))[same] > 0).all()


@pytest.mark.skipif(not oracle.have_ref("libref_kernels.so"), reason="oracle/_ref/libref_kernels.so not built (needs /root/reference)")
def test_golden_file_regenerates_from_the_reference_kernels():
    from oracle import make_golden_refk_sh_isect as mg
    for name, c in U.SH_CASES.items():
        for k, v in mg.run_sh(c).items():
            assert np.array_equal(v, GOLD[f"sh/{name}/{k}"], equal_nan=True), (name, k)
    for name, c in U.ISECT_CASES.items():
        for k, v in mg.run_isect(c).items():
            assert np.array_equal(v, GOLD[f"isect/{name}/{k}"]), (name, k)
