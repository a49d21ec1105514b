"""Generate tests/golden/refk_fastgs.npz from the REFERENCE's own fastgs rasterizer run on the CPU (oracle/_ref/libref_fastgs.so: forward.cu, backward.cu
and the kernels in fastgs/rasterization/include compiled in place as host code under oracle/ref_emul/, `make -C oracle refk_fastgs`). Run in the build
container, where /root/reference exists:   python oracle/make_golden_refk_fastgs.py
The file pins SURVEY.md §8f row 1 (fast_gs::rasterization::forward / backward: preprocess, depth + tile sorts, instance creation with exact tile culling,
blend, bucketed blend backward, preprocess backward incl. densification_info) to the reference itself: the oracle restatement (oracle_fastgs.hpp) and
the HIP kernels are both compared with it (tests/test_oracle_refk_golden.py, tests/test_gpu_refk_golden.py)."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

assert oracle.refk_fastgs_lib() is not None, "build oracle/_ref/libref_fastgs.so first (make -C oracle refk_fastgs)"

CASES = {   # name: scene parameters (the generator below) + whether densification_info is requested
    "deg3_128x96": dict(N=1500, W=128, H=96, seed=0, deg=3, spread=1.2, dens=True),
    "deg0_ragged_203x117": dict(N=1200, W=203, H=117, seed=1, deg=0, spread=1.2, dens=False),
    "deg1_dense_long_lists_64x64": dict(N=1500, W=64, H=64, seed=2, deg=1, spread=0.4, dens=True),     # up to ~1200 instances per tile: many buckets, early termination
    "deg2_sparse_176x144": dict(N=1200, W=176, H=144, seed=3, deg=2, spread=2.5, dens=False),
}


def scene(N, W, H, seed, deg, spread, **_):
    rng = np.random.default_rng(seed)
    means = rng.standard_normal((N, 3)) * spread
    means[:, 2] = np.abs(means[:, 2]) + 3.0
    scales_raw = np.log(rng.uniform(0.03, 0.25, (N, 3)))
    rot_raw = rng.standard_normal((N, 4))
    opac_raw = rng.standard_normal(N) * 1.5
    sh0 = rng.standard_normal((N, 1, 3)) * 0.5
    sh_rest = rng.standard_normal((N, 15, 3)) * 0.2       # storage for degree 3, `deg` of them active
    ang = 0.1
    R = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
    w2c = np.eye(4); w2c[:3, :3] = R; w2c[:3, 3] = [0.1, -0.05, 0.2]
    cam_pos = -R.T @ w2c[:3, 3]
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    return dict(means=f32(means), scales_raw=f32(scales_raw), rot_raw=f32(rot_raw), opac_raw=f32(opac_raw), sh0=f32(sh0), sh_rest=f32(sh_rest), w2c=f32(w2c),
                cam_pos=f32(cam_pos), active_sh_bases=(deg + 1) ** 2, W=W, H=H, fx=70.0, fy=75.0, cx=W / 2 + 0.3, cy=H / 2 - 0.2)


out = {}
for name, cfg in CASES.items():
    sc = scene(**cfg)
    rng = np.random.default_rng(500 + cfg["seed"])
    g_image = rng.standard_normal((3, sc["H"], sc["W"])).astype(np.float32)
    g_alpha = rng.standard_normal((1, sc["H"], sc["W"])).astype(np.float32)
    dens0 = rng.uniform(0, 2, (2, cfg["N"])).astype(np.float32) if cfg["dens"] else None      # accumulated into, not overwritten
    r = oracle.refk_fastgs_fwd_bwd(sc["means"], sc["scales_raw"], sc["rot_raw"], sc["opac_raw"], sc["sh0"], sc["sh_rest"], sc["w2c"], sc["cam_pos"],
                                   sc["active_sh_bases"], sc["W"], sc["H"], sc["fx"], sc["fy"], sc["cx"], sc["cy"], g_image, g_alpha, densification_info=dens0)
    print(name, "visible / instances / buckets", r["counts"], "alpha max", float(r["alpha"].max()))
    assert r["counts"][1] > 1000 and r["alpha"].max() > 0.3, "degenerate case"
    for k, v in sc.items():
        out[f"{name}/{k}"] = np.asarray(v)
    out[f"{name}/g_image"], out[f"{name}/g_alpha"] = g_image, g_alpha
    if dens0 is not None:
        out[f"{name}/densification_info_in"] = dens0
    for k, v in r.items():
        out[f"{name}/out_{k}"] = v
path = os.path.join(ROOT, "tests", "golden", "refk_fastgs.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path) // 1024, "KiB")
