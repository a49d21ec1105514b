"""Run by tests/test_gpu_raster.py in a subprocess per library (LFS_GSPLAT_LIB): the 3DGUT and the EWA backward on fixed inputs, deterministic accumulation
(debug bit 4) for the former -> an .npz of the gradients. Not a test on its own."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main(out_path):
    import lichtfeld_studio_amd as lfs
    from gpu_util import make_gaussians, pinhole_K, small_rotation_viewmat, t
    from lichtfeld_studio_amd import ops, scenes
    from lichtfeld_studio_amd.trainer import GutTrainer
    import oracle
    rng = np.random.default_rng(123)
    N, W, H = 12000, 256, 192
    means, quats, scales, opac = make_gaussians(rng, N, smin=0.01, smax=0.08)
    vm0 = np.stack([small_rotation_viewmat(rng, 0.05, 0.1)])
    K = pinhole_K(0.8 * W, W, H, 1)
    colors = rng.random((1, N, 3)).astype(np.float32)
    radii, m2, d, _, _ = oracle.projection_ut_3dgs_fused(means, quats / np.linalg.norm(quats, axis=-1, keepdims=True), scales, opac, vm0, None, K, W, H)
    _, ids, flat = oracle.intersect_tile(m2, radii, d, 1, 16, W // 16, H // 16, True)
    offs = oracle.intersect_offset(ids, 1, W // 16, H // 16)
    args = (t(means), t(quats), t(scales), t(colors), t(opac[None]), t(rng.random((1, 3)).astype(np.float32)), None, W, H, 16, t(vm0), None, t(K),
            lfs.CameraModelType.PINHOLE, None, lfs.ShutterType.GLOBAL, None, None, None, t(offs, torch.int32), t(flat, torch.int32))
    rc, ra, li = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)
    g = torch.Generator().manual_seed(9)
    v_rc, v_ra = torch.randn(rc.shape, generator=g).to(rc.device), torch.randn(ra.shape, generator=g).to(ra.device)
    lib = lfs.load_library()
    out = {"library": np.frombuffer(lib.lfs_version(), dtype=np.uint8)}
    lib.lfs_set_debug_flags(16)
    try:
        for name, x in zip(["v_means", "v_quats", "v_scales", "v_colors", "v_opacities"], ops.rasterize_to_pixels_from_world_3dgs_bwd(*args, ra, li, v_rc, v_ra)):
            out["gut_" + name] = x.cpu().numpy()
    finally:
        lib.lfs_set_debug_flags(0)
    # the EWA (fastgs) blend backward: its nine sums go through wave_sum9_atomic_lds (float atomics: two runs for the caller's noise bar)
    sc = scenes.syn_a(n=6000, sh_degree=1)
    target = scenes.target_image(sc.height, sc.width).to("cuda:0")
    for rep in range(2):
        tr = GutTrainer(sc, torch.device("cuda:0"), iterations=100, rasterizer="fastgs")
        tr.train_step([target], views=[0])
        for name, gv in zip(["means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opacities"], tr.bucket.views):
            out[f"fastgs{rep}_{name}"] = gv.detach().cpu().numpy().copy()
    torch.cuda.synchronize()
    np.savez(out_path, **out)


if __name__ == "__main__":
    main(sys.argv[1])
