"""Generate tests/golden/refk_loss.npz from the REFERENCE's own fused-SSIM and bilateral-grid kernels run on the CPU (oracle/_ref/libref_loss.so:
src/training/kernels/ssim.cu and bilateral_grid_{forward,backward,tv}.cu, kernel parts compiled in place as host code under oracle/ref_emul/,
`make -C oracle refk_loss`). Run in the build container, where /root/reference exists:   python oracle/make_golden_refk_loss.py
The file pins SURVEY.md §8f row 2 to the reference itself: the torch / numpy restatements used as oracle for these kernels (tests/ssim_reference.py,
oracle/bilateral.py) and the HIP kernels are compared with it (tests/test_oracle_refk_loss_golden.py, tests/test_gpu_refk_golden.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

assert oracle.refk_loss_lib() is not None, "build oracle/_ref/libref_loss.so first (make -C oracle refk_loss)"
C1, C2 = 0.01 ** 2, 0.03 ** 2
out = {}
for name, (B, H, W, seed) in {"ssim_ragged_1x3x45x37": (1, 45, 37, 0), "ssim_batch_2x3x24x40": (2, 24, 40, 1), "ssim_small_1x3x9x21": (1, 9, 21, 2)}.items():
    rng = np.random.default_rng(seed)
    a = rng.random((B, 3, H, W)).astype(np.float32)
    b = np.clip(a + 0.15 * rng.standard_normal((B, 3, H, W)), 0, 1).astype(np.float32)
    dmap = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    m, d1, d2, d3 = oracle.refk_fusedssim(C1, C2, a, b, True)
    g = oracle.refk_fusedssim_backward(C1, C2, a, b, dmap, d1, d2, d3)
    print(name, "mean ssim", float(m.mean()))
    out.update({f"{name}/img1": a, f"{name}/img2": b, f"{name}/dL_dmap": dmap, f"{name}/ssim_map": m, f"{name}/dm_dmu1": d1, f"{name}/dm_dsigma1_sq": d2,
                f"{name}/dm_dsigma12": d3, f"{name}/dL_dimg1": g})
for name, (L, H, W, h, w, seed) in {"slice_8x16x16_41x53": (8, 16, 16, 41, 53, 3), "slice_4x5x7_23x31": (4, 5, 7, 23, 31, 4)}.items():
    rng = np.random.default_rng(seed)
    grid = (np.eye(4)[:3].reshape(12)[:, None, None, None] + 0.3 * rng.standard_normal((12, L, H, W))).astype(np.float32)
    rgb = (rng.random((h, w, 3)) * 0.96 + 0.02).astype(np.float32)
    go = rng.standard_normal((h, w, 3)).astype(np.float32)
    o_, gg, gr = oracle.refk_bilateral_slice(grid, rgb, go)
    out.update({f"{name}/grid": grid, f"{name}/rgb": rgb, f"{name}/grad_output": go, f"{name}/output": o_, f"{name}/grad_grid": gg, f"{name}/grad_rgb": gr})
rng = np.random.default_rng(5)
grids = rng.standard_normal((3, 12, 6, 9, 11)).astype(np.float32)
loss, gg = oracle.refk_bilateral_tv(grids, 0.7)
print("tv loss", loss)
out.update({"tv_3x12x6x9x11/grids": grids, "tv_3x12x6x9x11/grad_output": np.float32(0.7), "tv_3x12x6x9x11/tv_loss": np.float32(loss), "tv_3x12x6x9x11/grad_grids": gg})
# ---- the host code around the kernels (oracle/_ref/libref_loss_host.so: fused_ssim.cuh's autograd wrapper, Trainer::compute_photometric_loss's three lines,
# BilateralGrid, WarmupExponentialLR compiled in place against CPU libtorch over the kernels above; `make -C oracle reflosshost`)
assert oracle.ref_loss_host_lib() is not None, "build oracle/_ref/libref_loss_host.so first (make -C oracle reflosshost)"
for name, (H, W, seed) in {"host/photometric_40x52": (40, 52, 6), "host/photometric_small_9x40": (9, 40, 7)}.items():   # <= 10 rows: the "valid" crop is skipped
    rng = np.random.default_rng(seed)
    gt = rng.random((3, H, W)).astype(np.float32)
    img = np.clip(gt + 0.1 * rng.standard_normal((3, H, W)), 0, 1).astype(np.float32)
    loss, g = oracle.ref_photometric_loss(img, gt, 0.2)
    sv, sg = oracle.ref_fused_ssim(img, gt, "valid")
    ss, ssg = oracle.ref_fused_ssim(img, gt, "same")
    print(name, "loss", float(loss), "ssim valid / same", float(sv), float(ss))
    out.update({f"{name}/rendered": img, f"{name}/gt": gt, f"{name}/loss": loss, f"{name}/grad": g, f"{name}/ssim_valid": sv, f"{name}/ssim_valid_grad": sg,
                f"{name}/ssim_same": ss, f"{name}/ssim_same_grad": ssg})
rng = np.random.default_rng(8)
delta = (0.2 * rng.standard_normal((3, 12, 4, 5, 6))).astype(np.float32)
rgb = (rng.random((3, 23, 31)) * 1.4 - 0.2).astype(np.float32)                       # outside [0, 1] in places: apply() clamps first
v_out = rng.standard_normal((3, 23, 31)).astype(np.float32)
r = oracle.ref_bilateral_grid(3, 6, 5, 4, 1, delta, rgb, v_out, 10.0)
out.update({"host/bilateral/delta": delta, "host/bilateral/rgb": rgb, "host/bilateral/v_out": v_out, "host/bilateral/tv_weight": np.float32(10.0)})
out.update({f"host/bilateral/{k}": v for k, v in r.items()})
out["host/warmup/args"] = np.array([2e-3, 0.01 ** (1.0 / 300), 40, 0.01, 120], np.float64)   # lr0, gamma, warmup_steps, warmup_start_factor, n
out["host/warmup/lrs"] = oracle.ref_warmup_schedule(2e-3, 0.01 ** (1.0 / 300), 40, 0.01, 120)
path = os.path.join(ROOT, "tests", "golden", "refk_loss.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path) // 1024, "KiB")
