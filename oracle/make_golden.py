"""Generate tests/golden/*.npz from the REFERENCE's own CPU code (tests/torch_impl.cpp, built in
place into oracle/_ref by `make -C oracle ref`). Run in the build container (where /root/reference
exists):   python oracle/make_golden.py
The vectors pin the parts of the hot path the reference's tests still cover:
spherical harmonics forward, quat->rotmat, tile intersection (exact), plus the reference's EWA
pinhole projection as an indicative cross-check for the UT projection."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

out = os.path.join(ROOT, "tests", "golden")
os.makedirs(out, exist_ok=True)
assert oracle.ref_lib() is not None, "build oracle/_ref first (make -C oracle ref)"

rng = np.random.default_rng(42)

# --- spherical harmonics (tests/torch_impl.cpp:221-322; test_garden_data.cpp:378-431 tolerances 1e-4)
N = 512
dirs = rng.standard_normal((N, 3)).astype(np.float32) * rng.uniform(0.1, 5.0, (N, 1)).astype(np.float32)
sh = {}
for deg in range(5):
    K = (deg + 1) ** 2
    coeffs = rng.standard_normal((N, K, 3)).astype(np.float32)
    sh[f"coeffs{deg}"] = coeffs
    sh[f"colors{deg}"] = oracle.ref_spherical_harmonics(deg, dirs, coeffs)
np.savez_compressed(os.path.join(out, "sh_fwd.npz"), dirs=dirs, **sh)

# --- quat -> rotmat (tests/torch_impl.cpp:8-36)
q = rng.standard_normal((256, 4)).astype(np.float32)
np.savez_compressed(os.path.join(out, "quat_to_rotmat.npz"), quats=q, rotmats=oracle.ref_quat_to_rotmat(q))

# --- tile intersection, the reference's exact-equality recipe (tests/test_garden_data.cpp:531-569):
#     C=3, N=1000, 40x60 image, tile 16; means2d ~ N(0,1)*40, radii ~ randint[0,40), depths ~ U(0,1)
def isect_case(C, N, W, H, ts, scale, rmax, seed):
    r = np.random.default_rng(seed)
    m = (r.standard_normal((C, N, 2)) * scale).astype(np.float32)
    rad = r.integers(0, rmax, (C, N, 2)).astype(np.int32)
    d = r.random((C, N)).astype(np.float32)
    tw, th = (W + ts - 1) // ts, (H + ts - 1) // ts
    tpg, ids, flat = oracle.ref_isect_tiles(m, rad, d, ts, tw, th, True)
    utpg, uids, uflat = oracle.ref_isect_tiles(m, rad, d, ts, tw, th, False)
    return dict(means2d=m, radii=rad, depths=d, tile_size=ts, tile_width=tw, tile_height=th,
                tiles_per_gauss=tpg, isect_ids=ids, flatten_ids=flat, isect_ids_unsorted=uids, flatten_ids_unsorted=uflat)

# NB: torch_impl uses tile_n_bits = ceil(log2 T) while the kernel uses floor(log2 T)+1; they agree unless T is a
# power of two. 40x60 -> T=12 (4 bits both); C=1 cases are insensitive anyway.
np.savez_compressed(os.path.join(out, "isect_c3_n1000_40x60.npz"), **isect_case(3, 1000, 40, 60, 16, 40.0, 40, 42))
np.savez_compressed(os.path.join(out, "isect_c1_n5000_200x120.npz"), **isect_case(1, 5000, 200, 120, 16, 90.0, 25, 7))

# --- EWA pinhole projection of the reference (indicative only for the UT kernel)
N = 2000
means = rng.standard_normal((N, 3)).astype(np.float32); means[:, 2] = np.abs(means[:, 2]) + 3
quats = rng.standard_normal((N, 4)).astype(np.float32)
scales = (rng.random((N, 3)) * 0.02 + 0.005).astype(np.float32)
vm = np.eye(4, dtype=np.float32); K = np.array([[200, 0, 128], [0, 200, 128], [0, 0, 1]], np.float32)
radii, m2, d, con = oracle.ref_fully_fused_projection(means, quats, scales, vm, K, 256, 256)
np.savez_compressed(os.path.join(out, "ewa_projection.npz"), means=means, quats=quats, scales=scales, viewmat=vm, K=K,
                    radii=radii, means2d=m2, depths=d, conics=con)
print("golden vectors written to", out, os.listdir(out))
