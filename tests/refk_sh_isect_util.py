"""Inputs shared by oracle/make_golden_refk_sh_isect.py (the reference's SH and tile-intersection KERNELS run on the CPU under oracle/ref_emul/) and the tests that
compare the oracle (CPU) and the HIP kernels (MI355X) with tests/golden/refk_sh_isect.npz: integer-hash inputs (refstrategy_util.hashed), outputs only on file."""
import numpy as np

from refstrategy_util import hashed

GOLD = "refk_sh_isect.npz"
SH_CASES = {f"deg{d}_of_{K}": dict(N=257, K=K, degree=d) for d, K in ((0, 1), (1, 4), (2, 9), (3, 16), (4, 25), (2, 16))}   # (2, 16): fewer degrees used than stored
ISECT_CASES = {
    # name: C, N, W, H, tile, largest radius. The first is the shape of the reference's own exact-equality recipe (tests/test_garden_data.cpp:531-569)
    "c3_n1000_40x60": dict(C=3, N=1000, W=40, H=60, tile=16, rmax=40),
    "c1_n3000_200x136_ties": dict(C=1, N=3000, W=200, H=136, tile=16, rmax=30),
    "c2_n800_tile8": dict(C=2, N=800, W=120, H=72, tile=8, rmax=12),
    "c1_n300_empty": dict(C=1, N=300, W=64, H=64, tile=16, rmax=0),
    "c1_n500_power_of_two_tiles": dict(C=1, N=500, W=128, H=128, tile=16, rmax=20),          # 64 tiles: floor(log2) + 1 = 7 key bits
}


def sh_inputs(c):
    N, K = c["N"], c["K"]
    dirs = (hashed((N, 3), 31) * 4.0).astype(np.float32)                 # not normalised: the kernel normalises
    dirs[5] = 0                                                          # a zero direction
    coeffs = hashed((N, K, 3), 32).astype(np.float32)
    masks = hashed((N,), 33) > -0.3
    v_colors = hashed((N, 3), 34).astype(np.float32)
    return dirs, coeffs, masks, v_colors


def isect_inputs(c):
    Cn, N, W, H = c["C"], c["N"], c["W"], c["H"]
    m = np.stack([(hashed((Cn, N), 41) + 0.5) * 1.2 * W - 0.1 * W, (hashed((Cn, N), 42) + 0.5) * 1.2 * H - 0.1 * H], -1).astype(np.float32)
    r = np.floor((hashed((Cn, N, 2), 43) + 0.5) * (c["rmax"] + 1)).astype(np.int32)
    r[hashed((Cn, N), 44) < -0.3] = 0                                    # culled by the projection
    d = ((hashed((Cn, N), 45) + 0.5) * 20 + 0.5).astype(np.float32)
    if "ties" in str(c.get("name", "")) or c["N"] == 3000:
        d = (np.round(d * 4) / 4).astype(np.float32)                     # equal depths: the order inside a tile falls back to the flatten id (stable sort)
    return m, r, d
