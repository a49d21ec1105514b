"""The integer stage (csrc/intersect.hip: count, scan, both scatter paths, the per-tile sort classes) compiled as HOST code on the wavefront
emulator (tests/emul) and driven through ops.intersect_tile: tiles_per_gauss, isect_ids, flatten_ids and the offsets must be BIT-identical to
the CPU oracle (the restatement of gsplat/IntersectTile.cu), for the two-pass binned scatter (default when a scratch array is passed), the
one-pass scatter (debug bit 5), the fallbacks of the two-pass kernels (staging overflow, chunks across many rows, too many rows) and with the
workspace counters left zero by the previous call (LFS_ISECT_COUNTERS_ZERO)."""
import ctypes as C
import os
import subprocess
import zlib

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
CSRC = os.path.join(ROOT, "lichtfeld-studio_amd", "csrc")


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ to build the emulated kernels")
    out = str(tmp_path_factory.mktemp("emul") / "liblfs_isect_emul.so")
    cmd = [CLANG, "-x", "c++", "-std=c++17", "-O1", "-DLFS_EMULATE", "-fPIC", "-shared", "-I" + os.path.join(HERE, "emul"), "-Wno-unused-value",
           "-Wno-unknown-attributes", os.path.join(CSRC, "intersect.hip"), os.path.join(CSRC, "raster.hip"), os.path.join(HERE, "emul", "emul_stubs.cpp"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lib = C.CDLL(out)
    lib.lfs_intersect_tile_workspace_bytes.restype = C.c_size_t
    from lichtfeld_studio_amd import ops
    saved = {k: getattr(ops, k) for k in ("load_library", "require_gpu", "stream", "workspace")}
    store = {}

    def workspace(nbytes, dev, tag):   # persistent, like the product's
        buf = store.get(tag)
        if buf is None or buf.numel() < nbytes:
            buf = torch.full((max(int(nbytes), 256),), 0xA5, dtype=torch.uint8)   # garbage: the first call must not rely on a clean workspace
            store[tag] = buf
        return buf
    ops.load_library = lambda: lib
    ops.require_gpu = lambda *a: None
    ops.stream = lambda: None
    ops.workspace = workspace
    yield ops, lib
    for k, v in saved.items():
        setattr(ops, k, v)


def _inputs(seed, C_, N, W, H, rmax, dead=0.2, ties=True):
    g = np.random.default_rng(seed)
    m = np.stack([g.uniform(-0.1 * W, 1.1 * W, (C_, N)), g.uniform(-0.1 * H, 1.1 * H, (C_, N))], -1).astype(np.float32)
    r = g.integers(1, rmax + 1, (C_, N, 2)).astype(np.int32)
    r[g.random((C_, N)) < dead] = 0                     # culled by the projection
    r[g.random((C_, N)) < 0.05, 0] = 0                  # one radius zero: culled as well
    d = g.uniform(0.5, 20.0, (C_, N)).astype(np.float32)
    if ties:
        d = np.round(d * 4) / 4                          # many equal depths: the order inside a tile falls back to the flatten id
    return m, r, d.astype(np.float32)


CASES = {
    # name: (C, N, W, H, tile, max radius)
    "syn_a_like": (1, 3000, 256, 256, 16, 24),
    "two_cameras_ragged": (2, 1500, 200, 136, 16, 30),
    "dense_big_tiles": (1, 6000, 64, 64, 16, 48),           # 16 tiles of thousands of entries: the 1025-4096 / 4097-16384 sort classes, staging overflow
    "staging_overflow": (1, 2500, 64, 64, 16, 100),         # > 8192 intersections from one workgroup's 1024 Gaussians: direct stores in pass 1
    "wide_sparse_many_rows_per_chunk": (1, 1500, 4800, 160, 16, 12),   # one chunk spans 10 rows x 300 tiles > TILES_SPAN: plain scatter inside pass 2
    "too_many_rows": (1, 800, 16, 8320, 16, 20),            # 520 tile rows > ROWS_MAX: the one-pass kernel
    "tile_8": (1, 2000, 120, 72, 8, 10),
    "more_tiles_than_a_scan_slice": (1, 700, 1600, 1600, 16, 20),   # T = 10 000 > 8192: two slices of the scan kernel
    "empty": (1, 300, 64, 64, 16, 0),
}


@pytest.mark.parametrize("case", list(CASES))
def test_emulated_intersection_is_bit_exact(emu, oracle_mod, case):
    ops, lib = emu
    C_, N, W, H, ts, rmax = CASES[case]
    tw, th = (W + ts - 1) // ts, (H + ts - 1) // ts
    if rmax == 0:
        m, r, d = _inputs(3, C_, N, W, H, 1)
        r[:] = 0
    else:
        m, r, d = _inputs(zlib.crc32(case.encode()) % 1000, C_, N, W, H, rmax)
    tpg_o, ids_o, flat_o = oracle_mod.intersect_tile(m, r, d, C_, ts, tw, th, True)
    offs_o = oracle_mod.intersect_offset(ids_o, C_, tw, th)
    tm, tr, td = torch.from_numpy(m), torch.from_numpy(r), torch.from_numpy(d)
    runs = []
    for flags in (0, 0, 32):   # two-pass on a dirty workspace, two-pass on the workspace the first call left behind, one-pass
        lib.lfs_set_debug_flags(flags)
        try:
            runs.append(ops.intersect_tile(tm, tr, td, None, None, C_, ts, tw, th, True, return_offsets=True))
        finally:
            lib.lfs_set_debug_flags(0)
    for tpg, ids, flat, offs in runs:
        assert np.array_equal(tpg.numpy(), tpg_o)
        assert ids.shape[0] == ids_o.shape[0]
        assert np.array_equal(ids.numpy(), ids_o) and np.array_equal(flat.numpy(), flat_o)
        assert np.array_equal(offs.numpy(), offs_o)
    # the unsorted path (reference emission order)
    tpg, ids, flat = ops.intersect_tile(tm, tr, td, None, None, C_, ts, tw, th, False)
    tpg_u, ids_u, flat_u = oracle_mod.intersect_tile(m, r, d, C_, ts, tw, th, False)
    assert np.array_equal(ids.numpy(), ids_u) and np.array_equal(flat.numpy(), flat_u)
