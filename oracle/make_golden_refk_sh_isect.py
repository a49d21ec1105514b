"""Generate tests/golden/refk_sh_isect.npz from the REFERENCE'S OWN spherical-harmonics and tile-intersection KERNELS run on the CPU (oracle/_ref/libref_kernels.so:
gsplat/SphericalHarmonicsCUDA.cu and IntersectTile.cu kernel parts compiled in place under oracle/ref_emul/, the host sequence of Intersect.cpp restated with cub's
radix sort as the stable sort CUB documents; `make -C oracle refk`). Inputs: tests/refk_sh_isect_util.py. Run in the build container:
    python oracle/make_golden_refk_sh_isect.py
With this every kernel of SURVEY.md §8a (K1 - K13) is pinned to its own reference kernel; the oracle (tests/test_oracle_refk_sh_isect_golden.py) and the HIP kernels
(tests/test_gpu_refk_golden.py) are compared with the file."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import refk_sh_isect_util as U  # noqa: E402


def run_sh(c):
    dirs, coeffs, masks, v = U.sh_inputs(c)
    colors = oracle.refk_sh_fwd(c["degree"], dirs, coeffs, masks)
    v_coeffs, v_dirs = oracle.refk_sh_bwd(c["degree"], dirs, coeffs, masks, v, True)
    unmasked = oracle.refk_sh_fwd(c["degree"], dirs, coeffs, None)
    return dict(colors=colors, colors_unmasked=unmasked, v_coeffs=v_coeffs, v_dirs=v_dirs)


def run_isect(c):
    m, r, d = U.isect_inputs(c)
    tw, th = (c["W"] + c["tile"] - 1) // c["tile"], (c["H"] + c["tile"] - 1) // c["tile"]
    tpg, ids, flat = oracle.refk_intersect_tile(m, r, d, c["tile"], tw, th, True)
    _, ids_u, flat_u = oracle.refk_intersect_tile(m, r, d, c["tile"], tw, th, False)
    return dict(tiles_per_gauss=tpg, isect_ids=ids, flatten_ids=flat, isect_ids_unsorted=ids_u, flatten_ids_unsorted=flat_u,
                offsets=oracle.refk_intersect_offset(ids, c["C"], tw, th))


if __name__ == "__main__":
    assert oracle.refk_lib() is not None
    out = {}
    for name, c in U.SH_CASES.items():
        for k, v in run_sh(c).items():
            out[f"sh/{name}/{k}"] = v
    for name, c in U.ISECT_CASES.items():
        r = run_isect(c)
        print(name, "n_isects", len(r["isect_ids"]))
        for k, v in r.items():
            out[f"isect/{name}/{k}"] = v
    path = os.path.join(ROOT, "tests", "golden", U.GOLD)
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")
