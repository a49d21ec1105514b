"""GPU parity: K3-K6 tile intersection — bit-exact (rtol = atol = 0, as tests/test_garden_data.cpp:566-568)
against the oracle and against the golden vectors produced by the reference's own CPU code."""
import os

import numpy as np
import pytest
import torch

from gpu_util import n, t

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _check(oracle, ops, m, r, d, C, ts, tw, th):
    for sort in (True, False):
        o_tpg, o_ids, o_flat = oracle.intersect_tile(m, r, d, C, ts, tw, th, sort)
        res = ops.intersect_tile(t(m), t(r, torch.int32), t(d), None, None, C, ts, tw, th, sort, return_offsets=True)
        assert np.array_equal(n(res[0]), o_tpg)
        assert res[1].dtype == torch.int64 and res[2].dtype == torch.int32
        assert np.array_equal(n(res[1]), o_ids), f"isect_ids differ (sort={sort})"
        assert np.array_equal(n(res[2]), o_flat), f"flatten_ids differ (sort={sort})"
        if sort:
            o_off = oracle.intersect_offset(o_ids, C, tw, th)
            assert np.array_equal(n(res[3]), o_off)
            assert np.array_equal(n(ops.intersect_offset(res[1], C, tw, th)), o_off)
    return len(o_ids)


@pytest.mark.parametrize("name", ["isect_c3_n1000_40x60.npz", "isect_c1_n5000_200x120.npz"])
def test_intersect_matches_reference_golden(lfs, name):
    from lichtfeld_studio_amd import ops
    g = np.load(os.path.join(GOLD, name))
    C = g["means2d"].shape[0]
    ts, tw, th = int(g["tile_size"]), int(g["tile_width"]), int(g["tile_height"])
    res = ops.intersect_tile(t(g["means2d"]), t(g["radii"], torch.int32), t(g["depths"]), None, None, C, ts, tw, th, True)
    assert np.array_equal(n(res[0]), g["tiles_per_gauss"])
    assert np.array_equal(n(res[1]), g["isect_ids"]) and np.array_equal(n(res[2]), g["flatten_ids"])
    res = ops.intersect_tile(t(g["means2d"]), t(g["radii"], torch.int32), t(g["depths"]), None, None, C, ts, tw, th, False)
    assert np.array_equal(n(res[1]), g["isect_ids_unsorted"]) and np.array_equal(n(res[2]), g["flatten_ids_unsorted"])


@pytest.mark.parametrize("C,N,W,H,ts", [(1, 20000, 640, 360, 16), (3, 4000, 200, 120, 16), (2, 3000, 130, 70, 8), (1, 5000, 512, 512, 32), (1, 1, 16, 16, 16)])
def test_intersect_random(lfs, oracle_mod, C, N, W, H, ts):
    from lichtfeld_studio_amd import ops
    rng = np.random.default_rng(C * 100 + N)
    tw, th = (W + ts - 1) // ts, (H + ts - 1) // ts
    m = np.stack([rng.uniform(-0.2 * W, 1.2 * W, (C, N)), rng.uniform(-0.2 * H, 1.2 * H, (C, N))], -1).astype(np.float32)
    r = rng.integers(0, 3 * ts, (C, N, 2)).astype(np.int32)
    r[rng.random((C, N)) < 0.2] = 0                       # culled Gaussians
    d = rng.uniform(0.01, 50, (C, N)).astype(np.float32)
    assert _check(oracle_mod, ops, m, r, d, C, ts, tw, th) > 0


def test_intersect_depth_ties_negative_depths_and_huge_footprints(lfs, oracle_mod):
    """ties in the 32-bit depth key must keep flatten-id order; negative depths sort after positive
    ones (raw bit pattern); footprints covering the whole image make buckets larger than 4096 and
    16384 entries, exercising all three per-tile sort size classes."""
    from lichtfeld_studio_amd import ops
    rng = np.random.default_rng(77)
    C, N, W, H, ts = 1, 30000, 64, 48, 16
    tw, th = 4, 3
    m = np.stack([rng.uniform(0, W, (C, N)), rng.uniform(0, H, (C, N))], -1).astype(np.float32)
    r = rng.integers(1, 8, (C, N, 2)).astype(np.int32)
    r[0, :20000] = 200                                   # 20000 Gaussians cover every tile -> buckets > 16384
    d = rng.choice(np.array([0.5, 1.0, 1.0, 2.0, -1.0, 3.5], np.float32), (C, N))   # massive ties + negatives
    total = _check(oracle_mod, ops, m, r, d, C, ts, tw, th)
    assert total > 12 * 20000
    # medium buckets (4097..16384)
    r[0, :20000] = 0; r[0, :9000] = 200
    _check(oracle_mod, ops, m, r, d, C, ts, tw, th)


def test_intersect_empty_and_all_culled(lfs):
    from lichtfeld_studio_amd import ops
    z = torch.zeros
    res = ops.intersect_tile(z((1, 0, 2), device="cuda:0"), z((1, 0, 2), dtype=torch.int32, device="cuda:0"), z((1, 0), device="cuda:0"),
                             None, None, 1, 16, 4, 4, True, return_offsets=True)
    assert res[1].numel() == 0 and res[2].numel() == 0 and int(res[3].abs().sum()) == 0
    m = torch.rand((2, 100, 2), device="cuda:0") * 64
    res = ops.intersect_tile(m, z((2, 100, 2), dtype=torch.int32, device="cuda:0"), torch.rand((2, 100), device="cuda:0"),
                             None, None, 2, 16, 4, 4, True, return_offsets=True)
    assert int(res[0].sum()) == 0 and res[1].numel() == 0 and int(res[3].abs().sum()) == 0
    off = ops.intersect_offset(res[1], 2, 4, 4)
    assert off.shape == (2, 4, 4) and int(off.abs().sum()) == 0     # IntersectTile.cu:268-271


def test_intersect_full_size_properties(lfs):
    """BASELINE config-2 sizes (1M Gaussians, 1080p): size-independent properties — keys sorted, stable
    w.r.t. flatten id on ties, offsets are the lower bounds, histogram of values == tiles_per_gauss."""
    from lichtfeld_studio_amd import ops
    g = torch.Generator(device="cuda:0").manual_seed(3)
    N, W, H, ts = 1_000_000, 1920, 1080, 16
    tw, th = 120, 68
    m = torch.rand((1, N, 2), device="cuda:0", generator=g) * torch.tensor([W * 1.1, H * 1.1], device="cuda:0") - 50
    r = torch.randint(0, 20, (1, N, 2), device="cuda:0", generator=g, dtype=torch.int32)
    d = torch.rand((1, N), device="cuda:0", generator=g) * 10 + 1
    tpg, ids, flat, off = ops.intersect_tile(m.contiguous(), r, d, None, None, 1, ts, tw, th, True, return_offsets=True)
    I = ids.numel()
    assert I == int(tpg.sum()) and I > 2_000_000
    assert bool((ids[1:] >= ids[:-1]).all())
    tie = ids[1:] == ids[:-1]
    assert bool((flat[1:][tie] > flat[:-1][tie]).all())
    assert torch.equal(torch.bincount(flat.long(), minlength=N).int(), tpg.view(-1))
    tile_of = (ids >> 32) & ((1 << 13) - 1)
    expect = torch.searchsorted(tile_of, torch.arange(tw * th, device="cuda:0"))
    assert torch.equal(off.view(-1).long(), expect)
    assert torch.equal(ops.intersect_offset(ids, 1, tw, th), off)
    # depth bits of every entry are the depth of its Gaussian
    assert torch.equal((ids & 0xFFFFFFFF).int(), d.view(-1)[flat.long()].view(torch.int32))


def test_two_pass_scatter_equals_one_pass_over_random_shapes(lfs):
    """The binned two-pass scatter (default) against the one-pass kernel (debug bit 5) over random problem shapes - cameras, image sizes down to one
    tile and up to thousands of tiles per row, tile sizes, dense and sparse lists, a reused workspace (LFS_ISECT_COUNTERS_ZERO) and the overlap form
    (counts written to pinned host memory): every output identical. (Each path against the oracle: the tests above and tests/test_emulated_intersect.py.)"""
    from lichtfeld_studio_amd import ops
    lib = lfs.load_library()
    g = np.random.default_rng(7)
    for trial in range(24):
        C = int(g.integers(1, 4))
        N = int(g.choice([1, 37, 1000, 20000, 150000]))
        ts = int(g.choice([8, 16, 32]))
        W, H = int(g.integers(ts, 2400)), int(g.integers(ts, 1400))
        if trial % 6 == 5:
            W, H = ts * int(g.integers(1, 4)), int(g.integers(4000, 9000))   # tall and narrow: more than 512 tile rows at tile 8 -> the fallback
        tw, th = (W + ts - 1) // ts, (H + ts - 1) // ts
        rmax = int(g.choice([3, 20, 120]))
        m = np.stack([g.uniform(-0.05 * W, 1.05 * W, (C, N)), g.uniform(-0.05 * H, 1.05 * H, (C, N))], -1).astype(np.float32)
        r = g.integers(0, rmax + 1, (C, N, 2)).astype(np.int32)
        d = (np.round(g.uniform(0.2, 30.0, (C, N)) * 8) / 8).astype(np.float32)
        tm, tr, td = t(m), t(r, torch.int32), t(d)
        outs = []
        for flags, overlap in ((0, None), (0, lambda: 1), (32, None)):
            lib.lfs_set_debug_flags(flags)
            try:
                res = ops.intersect_tile(tm, tr, td, None, None, C, ts, tw, th, True, return_offsets=True, overlap=overlap)
            finally:
                lib.lfs_set_debug_flags(0)
            outs.append([n(x) for x in res[:4]])
        for o in outs[1:]:
            for a, b in zip(outs[0], o):
                assert a.shape == b.shape and np.array_equal(a, b), (trial, C, N, W, H, ts)
        ids = outs[0][1]
        assert (np.diff(ids) >= 0).all()                       # sorted by (camera | tile | depth)
        assert int(outs[0][0].sum()) == ids.shape[0]
