import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import lichtfeld_studio_amd as lfs
from lichtfeld_studio_amd import ops, capi
from gpu_util import make_gaussians, pinhole_K, small_rotation_viewmat, t, n
rng = np.random.default_rng(40)
N, W, H, ts = 6000, 200, 136, 16
means, quats, scales, opac = make_gaussians(rng, N, spread=1.5, smin=0.01, smax=0.05)
vm0 = small_rotation_viewmat(rng, 0.2, 0.3)[None]
K = pinhole_K(0.8 * W, W, H, 1)
colors = rng.random((1, N, 3)).astype(np.float32)
tw, th = (W + ts - 1) // ts, (H + ts - 1) // ts
depth = (means @ vm0[0, :3, :3].T + vm0[0, :3, 3])[:, 2]
order = np.argsort(depth, kind="stable").astype(np.int32)
n_t = tw * th
per_tile = 400
flat = np.concatenate([np.sort(rng.choice(N, per_tile, replace=False)) for _ in range(n_t)]).astype(np.int32)
flat = order[flat.reshape(n_t, per_tile)].reshape(-1)
offs = (np.arange(n_t, dtype=np.int32) * per_tile).reshape(1, th, tw)
args = (t(means), t(quats), t(scales), t(colors), t(opac[None]), None, None, W, H, ts, t(vm0), None, t(K), lfs.CameraModelType.PINHOLE, None,
        lfs.ShutterType.GLOBAL, None, None, None, t(offs, torch.int32), t(flat, torch.int32))
lib = capi.load_library()
lib.lfs_set_debug_flags(1)
r0, a0, l0 = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)
lib.lfs_set_debug_flags(0)
r1, a1, l1 = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)
d = (r0 - r1).abs().amax(-1)[0]
print("differing pixels", int((d > 0).sum()), "of", d.numel(), "max", float(d.max()), "last ids differ", int((l0 != l1).sum()))
ys, xs = torch.nonzero(d > 0, as_tuple=True)
for y, x in list(zip(ys.tolist(), xs.tolist()))[:10]:
    print(y, x, float(d[y, x]), r0[0, y, x].tolist(), r1[0, y, x].tolist(), int(l0[0, y, x]), int(l1[0, y, x]))
# ---- which entries are missing for pixel (y,x)?
import ctypes as C
lib.lfs_set_debug_flags(1); ops.rasterize_to_pixels_from_world_3dgs_fwd(*args); ws0 = capi.workspace(1, torch.device("cuda:0"), "raster").clone()
lib.lfs_set_debug_flags(0); ops.rasterize_to_pixels_from_world_3dgs_fwd(*args); ws1 = capi.workspace(1, torch.device("cuda:0"), "raster").clone()
al = lambda v: (v + 255) & ~255
I = flat.size; cells = n_t * 4
sizes = [al(64 * N), al(64 * N), al(32 * N), al(4 * cells), al(8 * 4 * I)]
lib.lfs_rasterize_workspace_bytes.restype = C.c_size_t
nbytes = lib.lfs_rasterize_workspace_bytes(C.c_uint32(1), C.c_uint32(N), C.c_uint32(3), C.c_uint32(W), C.c_uint32(H), C.c_uint32(ts), C.c_int64(I))
o = nbytes - sum(sizes) + sizes[0] + sizes[1] + sizes[2]
def lists(ws):
    cnt = ws[o:o + 4 * cells].view(torch.int32).cpu().numpy()
    cl = ws[o + sizes[3]: o + sizes[3] + 8 * 4 * I].view(torch.int32).reshape(-1, 2).cpu().numpy()
    return cnt, cl
cnt0, cl0 = lists(ws0); cnt1, cl1 = lists(ws1)
y, x = ys[0].item(), xs[0].item()
tile = (y // 16) * tw + (x // 16); wl = ((y % 16) // 8) * 2 + ((x % 16) // 8)
st = tile * per_tile
seg = 4 * st + wl * per_tile
e0 = cl0[seg: seg + cnt0[tile * 4 + wl]]; e1 = cl1[seg: seg + cnt1[tile * 4 + wl]]
print("pixel", y, x, "tile", tile, "wl", wl, "counts", len(e0), len(e1))
missing = sorted(set(e0[:, 0].tolist()) - set(e1[:, 0].tolist()))
Rc = vm0[0, :3, :3].astype(np.float64); tc = vm0[0, :3, 3].astype(np.float64)
f = 0.8 * W
def rot(q):
    q = q / np.linalg.norm(q, axis=1, keepdims=True); w, x_, y_, z = q.T
    return np.stack([1-2*(y_*y_+z*z),2*(x_*y_-w*z),2*(x_*z+w*y_),2*(x_*y_+w*z),1-2*(x_*x_+z*z),2*(y_*z-w*x_),2*(x_*z-w*y_),2*(y_*z+w*x_),1-2*(x_*x_+y_*y_)],1).reshape(-1,3,3)
R = rot(quats.astype(np.float64)); s = scales.astype(np.float64)
M = (1 / s)[:, :, None] * np.transpose(R, (0, 2, 1))
oo = -Rc.T @ tc
dc = np.array([(x + 0.5 - W / 2) / f, (y + 0.5 - H / 2) / f, 1.0]); dc /= np.linalg.norm(dc); d = Rc.T @ dc
npass = 0
for g in missing:
    gro = M[g] @ (oo - means[g]); grd = M[g] @ d; grd /= np.linalg.norm(grd)
    a = opac[g] * np.exp(-0.5 * (np.cross(grd, gro) ** 2).sum())
    if a >= 1 / 255 * 0.9: npass += 1; print("missing gaussian", g, "alpha", a)
print("missing", len(missing), "passing among them", npass)
# order check: is e1 a subsequence of e0 ?
pos = {g: i for i, g in enumerate(e0[:, 0].tolist())}
p = [pos[g] for g in e1[:, 0].tolist()]
print("subsequence order ok", all(p[i] < p[i + 1] for i in range(len(p) - 1)), "idx match", all(e0[pos[g], 1] == ii for g, ii in e1.tolist()))
# determinism
lib.lfs_set_debug_flags(1); ra_ = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)[0].clone(); rb_ = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)[0].clone()
print("unculled deterministic", torch.equal(ra_, rb_), torch.equal(ra_, r0))
lib.lfs_set_debug_flags(0); rc_ = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)[0].clone(); rd_ = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)[0].clone()
print("culled deterministic", torch.equal(rc_, rd_), torch.equal(rc_, r1))
# which entries pass at this pixel (fp64), in e0 order
Tt = 1.0; acc = np.zeros(3)
for g, ii in e0.tolist():
    gro = M[g] @ (oo - means[g]); grd = M[g] @ d; grd /= np.linalg.norm(grd)
    a = min(0.999, opac[g] * np.exp(-0.5 * (np.cross(grd, gro) ** 2).sum()))
    if a < 1 / 255: continue
    print("contrib", g, ii, "alpha", a, "T", Tt, "in culled list", g in set(e1[:, 0].tolist()))
    acc += colors[0, g] * a * Tt; Tt *= 1 - a
print("fp64 result", acc, "gpu unculled", r0[0, y, x].tolist(), "culled", r1[0, y, x].tolist())
