"""Scenes shared by oracle/make_golden_ref_raster.py (the REFERENCE's rasterize() + autograd run on the CPU) and tests/test_gpu_raster_reference.py (the product's
render path on the MI355X): inputs from integer hashes (refstrategy_util.hashed: exact in float64, identical float32 on both sides), so the golden file holds
outputs only."""
import numpy as np

from refstrategy_util import hashed

GOLD = "ref_raster.npz"
CASES = {
    # name: N, shN coefficients K1, max / active SH degree, W, H, focal, pose seed (0 = identity), background, alpha gradient, scale range (log-uniform)
    "deg1_96x64_background": dict(N=800, K1=3, sh_degree=1, active=1, W=96, H=64, focal=80.0, pose=0, bg=(0.1, 0.2, 0.3), v_alpha=False, scales=(0.02, 0.15)),
    "deg3_active2_107x75_posed_no_background_alpha_gradient": dict(N=700, K1=15, sh_degree=3, active=2, W=107, H=75, focal=95.0, pose=5, bg=None, v_alpha=True,
                                                                  scales=(0.02, 0.12)),
    "deg3_active0_64x64_long_lists": dict(N=500, K1=15, sh_degree=3, active=0, W=64, H=64, focal=60.0, pose=9, bg=(0.0, 0.0, 0.0), v_alpha=False, scales=(0.1, 0.5)),
}


def scene(c):
    N, K1 = c["N"], c["K1"]
    f = lambda shape, salt, scale=1.0, shift=0.0: (hashed(shape, salt) * scale + shift).astype(np.float32)
    means = f((N, 3), 11, 5.0)
    means[:, 2] = np.abs(means[:, 2]) + np.float32(3.0)                       # in front of the identity camera
    lo, hi = np.log(c["scales"][0]), np.log(c["scales"][1])
    s = dict(means=means, sh0=f((N, 1, 3), 12, 1.2), shN=f((N, K1, 3), 13, 0.5), scaling=f((N, 3), 14, hi - lo, (hi + lo) / 2), rotation=f((N, 4), 15, 2.0),
             opacity=f((N,), 16, 6.0, 0.5))
    if c["pose"]:
        q = hashed((4,), 1000 + c["pose"]) + np.array([1.5, 0, 0, 0])       # a moderate rotation about a hashed axis
        w, x, y, z = q / np.linalg.norm(q)
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        T = hashed((3,), 2000 + c["pose"]) * np.array([1.0, 1.0, 0.5])
        # keep the cloud in front of the posed camera: world points = R^T (p_cam - T)
        s["means"] = ((means.astype(np.float64) - T) @ R).astype(np.float32)
    else:
        R, T = np.eye(3), np.zeros(3)
    s["R"], s["T"] = R.astype(np.float32), T.astype(np.float32)
    s["v_image"] = f((3, c["H"], c["W"]), 17, 2.0)
    s["v_alpha"] = f((1, c["H"], c["W"]), 18, 1.0) if c["v_alpha"] else None
    return s


def target_image(c):
    """the ground-truth image of the "train/..." entries (the trainer's loss against it)"""
    return (hashed((3, c["H"], c["W"]), 21) + 0.5).astype(np.float32)


BILATERAL = dict(n_images=3, gW=6, gH=5, gL=4, image_idx=1, tv_weight=10.0)      # config 5's appearance model on a small grid; tv_loss_weight 10 as in the json


def bilateral_delta():
    b = BILATERAL
    return (hashed((b["n_images"], 12, b["gL"], b["gH"], b["gW"]), 22) * 0.4).astype(np.float32)
