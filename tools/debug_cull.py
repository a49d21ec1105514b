import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import lichtfeld_studio_amd as lfs
from lichtfeld_studio_amd import ops, capi
from gpu_util import make_gaussians, pinhole_K, small_rotation_viewmat, t, n
rng = np.random.default_rng(40)
N, W, H, ts = 2000, 64, 64, 16
means, quats, scales, opac = make_gaussians(rng, N, spread=1.0)
vm0 = np.eye(4, dtype=np.float32)[None]
K = pinhole_K(0.8 * W, W, H, 1)
colors = rng.random((1, N, 3)).astype(np.float32)
tw, th = W // ts, H // ts
depth = means[:, 2]
order = np.argsort(depth, kind="stable").astype(np.int32)
n_t = tw * th
flat = np.tile(order, n_t).astype(np.int32)
offs = (np.arange(n_t, dtype=np.int32) * N).reshape(1, th, tw)
args = (t(means), t(quats), t(scales), t(colors), t(opac[None]), None, None, W, H, ts, t(vm0), None, t(K), lfs.CameraModelType.PINHOLE, None,
        lfs.ShutterType.GLOBAL, None, None, None, t(offs, torch.int32), t(flat, torch.int32))
lib = capi.load_library()
lib.lfs_set_debug_flags(1)
r0, a0, l0 = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)
ws0 = capi.workspace(1, torch.device("cuda:0"), "raster").clone()
lib.lfs_set_debug_flags(0)
r1, a1, l1 = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)
ws1 = capi.workspace(1, torch.device("cuda:0"), "raster").clone()
print("equal", torch.equal(r0, r1), float((r0 - r1).abs().max()), "alpha", float(a0.max()), float(a1.max()))
# layout
al = lambda v: (v + 255) & ~255
CamDevSize = None
total = ws1.numel()
cells = n_t * 4
# work backwards: cell_list = al(8*4*I), cell_count = al(4*cells), cull = al(32*N), acc = al(64*N), recs = al(64*N)
I = flat.size
o_list = total - al(8 * 4 * I) if False else None
sizes = [al(64 * N), al(64 * N), al(32 * N), al(4 * cells), al(8 * 4 * I)]
lib.lfs_rasterize_workspace_bytes.restype = __import__("ctypes").c_size_t
import ctypes as C
nbytes = lib.lfs_rasterize_workspace_bytes(C.c_uint32(1), C.c_uint32(N), C.c_uint32(3), C.c_uint32(W), C.c_uint32(H), C.c_uint32(ts), C.c_int64(I))
cam_sz = nbytes - sum(sizes)
print("nbytes", nbytes, "cam", cam_sz)
o = cam_sz
recs = ws1[o:o + 64 * N].view(torch.float32).reshape(N, 16).cpu().numpy(); o += sizes[0] + sizes[1]
cull = ws1[o:o + 32 * N].view(torch.float32).reshape(N, 8).cpu().numpy(); o += sizes[2]
cnt1 = ws1[o:o + 4 * cells].view(torch.int32).cpu().numpy()
cnt0 = ws0[o:o + 4 * cells].view(torch.int32).cpu().numpy()
print("counts on", cnt1.reshape(n_t, 4)[:4], "off", cnt0.reshape(n_t, 4)[:4])
print("cull rec sample", cull[:3])
# numpy expectation of the cull record
def rot(q):
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([1-2*(y*y+z*z),2*(x*y-w*z),2*(x*z+w*y),2*(x*y+w*z),1-2*(x*x+z*z),2*(y*z-w*x),2*(x*z-w*y),2*(y*z+w*x),1-2*(x*x+y*y)],1).reshape(-1,3,3)
R = rot(quats.astype(np.float64)); A = R * scales[:, None, :]; Sig = A @ np.transpose(A, (0, 2, 1))
r2 = np.maximum(0, 2 * np.log(255 * opac)) * 1.02 + 0.02
inv = 1 / means[:, 2]; k = r2 * inv ** 2
exp = np.stack([means[:, 0] * inv, means[:, 1] * inv, k * Sig[:, 0, 0], k * Sig[:, 1, 1], k * Sig[:, 0, 2], k * Sig[:, 1, 2], k * Sig[:, 2, 2]], 1)
print("expected", exp[:3])
