"""Generate tests/golden/ref_fast_raster.npz by running the REFERENCE'S OWN default (fastgs) training render path end to end on the CPU (oracle/_ref/libref_fast_raster.so:
fast_rasterizer.cpp, fast_rasterizer_autograd.cpp, rasterization_api.cu, forward.cu / backward.cu with the kernel headers, camera.cpp, SplatData's activations -
compiled in place, `make -C oracle reffast`; nothing restated) on scenes of tests/refraster_util.py: image, alpha, the gradients of the six raw parameter tensors
for the loss sum(image * v_image) + sum(alpha * v_alpha), and densification_info accumulated by the backward. Run in the build container:
    python oracle/make_golden_ref_fast_raster.py
tests/test_gpu_raster_reference.py holds the product's fastgs.fast_rasterize + backward to it (the composition around the kernels)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import refraster_util as U  # noqa: E402

GOLD = "ref_fast_raster.npz"
CASES = ["deg1_96x64_background", "deg3_active2_107x75_posed_no_background_alpha_gradient"]


def run(name):
    c = U.CASES[name]
    s = U.scene(c)
    bg = np.array(c["bg"] if c["bg"] is not None else (0.3, 0.5, 0.2), np.float32)     # fast_rasterize always composites a background
    v_alpha = s["v_alpha"] if s["v_alpha"] is not None else U.hashed((1, c["H"], c["W"]), 19).astype(np.float32)
    dens = (U.hashed((2, c["N"]), 20) + 0.5).astype(np.float32)
    r = oracle.ref_fast_render_backward(s["means"], s["sh0"], s["shN"], s["scaling"], s["rotation"], s["opacity"], c["sh_degree"], c["active"], s["R"], s["T"], c["focal"],
                                        c["focal"] * 1.05, c["W"] / 2 + 0.5, c["H"] / 2 - 0.25, c["W"], c["H"], bg, s["v_image"], v_alpha, dens)
    r["bg"], r["v_alpha"], r["densification_info_in"] = bg, v_alpha, dens
    return r


def run_train(name):
    """the trainer's photometric loss on this path (black background, as the trainer's default) and its gradients: "train/..." entries"""
    c = U.CASES[name]
    s = U.scene(c)
    dens = (U.hashed((2, c["N"]), 20) + 0.5).astype(np.float32)
    r = oracle.ref_fast_train_loss_backward(s["means"], s["sh0"], s["shN"], s["scaling"], s["rotation"], s["opacity"], c["sh_degree"], c["active"], s["R"], s["T"], c["focal"],
                                            c["focal"] * 1.05, c["W"] / 2 + 0.5, c["H"] / 2 - 0.25, c["W"], c["H"], np.zeros(3, np.float32), U.target_image(c), 0.2, dens)
    return r


if __name__ == "__main__":
    assert oracle.ref_fast_raster_lib() is not None, "build oracle/_ref/libref_fast_raster.so first (make -C oracle reffast)"
    out = {}
    for name in CASES:
        r = run(name)
        print(f"{name}: mean alpha {r['alpha'].mean():.3f}, |g_means| max {np.abs(r['g_means']).max():.3g}")
        for k, v in r.items():
            out[f"{name}/{k}"] = v
        for k, v in run_train(name).items():
            out[f"{name}/train/{k}"] = v
    path = os.path.join(ROOT, "tests", "golden", GOLD)
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")
