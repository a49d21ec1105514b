import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import lichtfeld_studio_amd as lfs
from lichtfeld_studio_amd import ops, capi
from gpu_util import make_gaussians, pinhole_K, small_rotation_viewmat, t, n
rng = np.random.default_rng(5)
N, W, H, ts = 64, 16, 16, 16
means, quats, scales, opac = make_gaussians(rng, N, spread=0.05, smin=0.02, smax=0.06)
means[0] = [50, 50, 5]  # dummy, far off screen
vm0 = np.eye(4, dtype=np.float32)[None]
K = pinhole_K(0.8 * 64, W, H, 1)
colors = rng.random((1, N, 3)).astype(np.float32)
lib = capi.load_library()
lib.lfs_set_debug_flags(1)
order = (np.argsort(means[1:, 2], kind="stable") + 1).astype(np.int32)[:9]
outs = []
for shift in range(5):
    flat = np.concatenate([np.zeros(shift, np.int32), order])
    offs = np.zeros((1, 1, 1), np.int32)
    args = (t(means), t(quats), t(scales), t(colors), t(opac[None]), None, None, W, H, ts, t(vm0), None, t(K), lfs.CameraModelType.PINHOLE, None,
            lfs.ShutterType.GLOBAL, None, None, None, t(offs, torch.int32), t(flat, torch.int32))
    r, a, l = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)
    outs.append(r.clone())
    print(shift, "equal to shift0", torch.equal(r, outs[0]), float((r - outs[0]).abs().max()), float(r.max()))
