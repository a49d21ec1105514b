"""Generate tests/golden/ref_raster.npz by running the REFERENCE'S OWN training-time render path end to end on the CPU (oracle/_ref/libref_raster_full.so:
rasterizer.cpp, rasterizer_autograd.cpp, camera.cpp, SplatData's activations AND the whole gsplat library - every .cu with its launch functions, every .cpp operator -
compiled in place against libtorch under oracle/ref_emul/; `make -C oracle refgsplat`). (oracle/_ref/libref_raster.so, `make refraster`, is the same render path
over the restated launch sequences of ref_raster_shim.cpp with the reference's tests/torch_impl.cpp for SH and intersection: it reproduces this file to 5e-6.) on the scenes of tests/refraster_util.py: image, alpha, radii, the camera matrices the Camera class
derives, and the gradients of the six raw parameter tensors for the loss sum(image * v_image) [+ sum(alpha * v_alpha)]. Run in the build container:
    python oracle/make_golden_ref_raster.py
tests/test_gpu_raster_reference.py holds the product's rasterize() + backward to it - the composition of SURVEY.md §8 rows a1-a6 as the reference composes them."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import refraster_util as U  # noqa: E402
from refstrategy_util import hashed as _h  # noqa: E402,F401


def run(c, full=False):
    s = U.scene(c)
    return oracle.ref_render_backward(s["means"], s["sh0"], s["shN"], s["scaling"], s["rotation"], s["opacity"], c["sh_degree"], c["active"], s["R"], s["T"], c["focal"],
                                      c["focal"] * 1.05, c["W"] / 2 + 0.5, c["H"] / 2 - 0.25, c["W"], c["H"], c["bg"], s["v_image"], s["v_alpha"], full=full)


TRAIN = dict(lambda_dssim=0.2, scale_reg=0.01, opacity_reg=0.01)      # eval/mcmc_optimization_params.json


def run_train(c, bilateral=False):
    """the trainer's loss (L1 + D-SSIM + regularisers [, through the bilateral grid, + TV]) and its gradients for a hashed target image: "train/..." entries"""
    s = U.scene(c)
    gt = U.target_image(c)
    b = dict(U.BILATERAL, delta=U.bilateral_delta()) if bilateral else None
    return oracle.ref_train_loss_backward(s["means"], s["sh0"], s["shN"], s["scaling"], s["rotation"], s["opacity"], c["sh_degree"], c["active"], s["R"], s["T"], c["focal"],
                                          c["focal"] * 1.05, c["W"] / 2 + 0.5, c["H"] / 2 - 0.25, c["W"], c["H"], c["bg"], gt, bilateral=b, **TRAIN)


if __name__ == "__main__":
    assert oracle.ref_raster_lib(full=True) is not None, "build oracle/_ref/libref_raster_full.so first (make -C oracle refgsplat)"
    out = {}
    for name, c in U.CASES.items():
        r = run(c, full=True)      # the reference's whole gsplat library behind its render path: nothing of the operator layer restated
        print(f"{name}: visible {(r['radii'] > 0).sum()} of {c['N']}, mean alpha {r['alpha'].mean():.3f}, |g_means| max {np.abs(r['g_means']).max():.3g}")
        for k, v in r.items():
            out[f"{name}/{k}"] = v
        tr = run_train(c)
        print(f"   train loss {float(tr['loss']):.6f}")
        for k, v in tr.items():
            if k != "image":
                out[f"{name}/train/{k}"] = v
        if name == "deg1_96x64_background":          # config 5's step: the rendered image goes through the bilateral grid before the loss, + TV
            for k, v in run_train(c, bilateral=True).items():
                if k != "image":
                    out[f"{name}/train_bilateral/{k}"] = v
    path = os.path.join(ROOT, "tests", "golden", U.GOLD)
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")
