"""Developer script (CPU only): wave-evaluation counts and lane utilisation of the rasterizer's kernels on a WINDOW of SYN-B (same
Gaussian statistics as the benchmark: the full scene, a 160x96-pixel crop of the 1080p view), run on the wavefront emulator of tests/emul.
    python tools/emul_eval_counts.py [width height]"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lichtfeld_studio_amd as lfs  # noqa: E402
from lichtfeld_studio_amd import ops, scenes  # noqa: E402
import oracle  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (160, 96)
out = os.path.join(tempfile.mkdtemp(), "liblfs_raster_emul.so")
subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-x", "c++", "-std=c++17", "-O2", "-DLFS_EMULATE", "-fPIC", "-shared", "-ffp-contract=on",
                       "-I" + os.path.join(ROOT, "tests", "emul"), "-Wno-unused-value", "-Wno-unknown-attributes",
                       os.path.join(ROOT, "lichtfeld-studio_amd", "csrc", "raster.hip"), os.path.join(ROOT, "tests", "emul", "emul_stubs.cpp"), "-o", out])
lib = C.CDLL(out)
lib.lfs_rasterize_workspace_bytes.restype = C.c_size_t
ops.load_library = lambda: lib
ops.require_gpu = lambda *a: None
ops.stream = lambda: None
ops.workspace = lambda nbytes, dev, tag: torch.zeros(max(int(nbytes), 256), dtype=torch.uint8)
counters = (C.c_ulonglong * 8).in_dll(lib, "lfs_emul_counters")

sc = scenes.syn_b(n=1_000_000, n_views=4)
quats = torch.nn.functional.normalize(sc.raw_quats, dim=-1).numpy(); scales = sc.raw_scales.exp().numpy(); opac = torch.sigmoid(sc.raw_opacities).numpy()
means = sc.means.numpy()
vm = sc.viewmats[:1].numpy().copy(); K = sc.Ks[:1].numpy().copy()
x0, y0 = (sc.width - W) // 2 + 200, (sc.height - H) // 2 + 100        # an off-centre window of the full view
K[0, 0, 2] -= x0; K[0, 1, 2] -= y0
radii, m2, d, _, _ = oracle.projection_ut_3dgs_fused(means, quats, scales, opac, vm, None, K, W, H, camera_model=0, rs_type=0, radial_coeffs=None,
                                                     tangential_coeffs=None, thin_prism_coeffs=None)
keep = np.nonzero((radii[0] > 0).all(axis=1))[0]
means, quats, scales, opac = means[keep], quats[keep], scales[keep], opac[keep]
radii, m2, d = radii[:, keep], m2[:, keep], d[:, keep]
tw, th = (W + 15) // 16, (H + 15) // 16
_, ids, flat = oracle.intersect_tile(m2, radii, d, 1, 16, tw, th, True)
offs = oracle.intersect_offset(ids, 1, tw, th)
N = len(keep)
print(f"window {W}x{H}: {N} Gaussians, {len(flat)} tile entries ({len(flat) / (tw * th):.0f} per tile; SYN-B full view: 546)")
rng = np.random.default_rng(0)
t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt)
colors = rng.random((1, N, 3)).astype(np.float32)
args = (t(means), t(quats), t(scales), t(colors), t(opac[None]), None, None, W, H, 16, t(vm), None, t(K), lfs.CameraModelType.PINHOLE, None,
        lfs.ShutterType.GLOBAL, None, None, None, t(offs, torch.int32), t(flat, torch.int32))
lib.lfs_set_debug_flags(0)
for i in range(8): counters[i] = 0
rc, ra, li = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)
ops.rasterize_to_pixels_from_world_3dgs_bwd(*args, ra, li, torch.randn_like(rc), torch.randn_like(ra))
c = list(counters)
print(f"8x8 cells: fwd evaluations {c[0]} (composited {c[1]}), bwd evaluations {c[2]} (accumulated {c[3]}) = {c[2] / len(flat):.3f} per tile entry")
print(f"accumulated bwd evaluations: {c[4] / c[3]:.1f} of 64 lanes live ({100 * c[4] / (64 * c[3]):.0f} %); 8x4 half cells with a live lane: {c[5] / c[3]:.2f} of 2; "
      f"4x4 quarters: {c[6] / c[3]:.2f} of 4")
