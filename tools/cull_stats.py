"""Developer script: how much of the tile lists survives the per-cell culling on SYN-B."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lichtfeld_studio_amd as lfs
from lichtfeld_studio_amd import ops, capi, scenes
from lichtfeld_studio_amd.rasterizer import Camera, SplatModel, rasterize
dev = torch.device("cuda:0")
sc = scenes.syn_b(n=1_000_000, n_views=4).to(dev)
model = SplatModel(sc.means, sc.sh0, sc.shN, sc.raw_scales, sc.raw_quats, sc.raw_opacities, 3)
cam = Camera(sc.viewmats[:1].contiguous(), sc.Ks[:1].contiguous(), sc.width, sc.height)
with torch.no_grad():
    out = rasterize(cam, model, torch.zeros(3, device=dev))
torch.cuda.synchronize()
ws = capi.workspace(1, dev, "raster")
N, W, H, ts = sc.N, sc.width, sc.height, 16
I = out.n_isects
tw, th = (W + 15) // 16, (H + 15) // 16
cells = tw * th * 4
al = lambda v: (v + 255) & ~255
sizes = [al(64 * N), al(64 * N), al(32 * N), al(4 * cells), al(8 * 4 * I)]
lib = capi.load_library()
nbytes = lib.lfs_rasterize_workspace_bytes(C.c_uint32(1), C.c_uint32(N), C.c_uint32(3), C.c_uint32(W), C.c_uint32(H), C.c_uint32(ts), C.c_int64(I))
o = nbytes - sum(sizes) + sizes[0] + sizes[1] + sizes[2]
cnt = ws[o:o + 4 * cells].view(torch.int32).cpu().numpy().astype(np.int64)
print("I", I, "4I", 4 * I, "sum cell_count", cnt.sum(), "ratio", cnt.sum() / (4 * I), "max", cnt.max(), "mean", cnt.mean())
# how far does each cell walk (last contributor index)? use last_ids via raw op
quats = torch.nn.functional.normalize(sc.raw_quats, dim=-1); scl = sc.raw_scales.exp(); opac = torch.sigmoid(sc.raw_opacities)[None]
radii, m2, d, _, _ = ops.projection_ut_3dgs_fused(sc.means, quats, scl, opac[0], cam.world_view_transform, None, cam.K, W, H, 0.3, 0.01, 1e4, 0.0, False, lfs.CameraModelType.PINHOLE)
_, ids, flat, offs = ops.intersect_tile(m2, radii, d, None, None, 1, 16, tw, th, True, return_offsets=True)
colors = torch.rand(1, N, 3, device=dev)
fa = (sc.means, quats, scl, colors, opac, None, None, W, H, 16, cam.world_view_transform, None, cam.K, lfs.CameraModelType.PINHOLE, None, lfs.ShutterType.GLOBAL, None, None, None, offs, flat)
rc, ra, li = ops.rasterize_to_pixels_from_world_3dgs_fwd(*fa)
torch.cuda.synchronize()
ws = capi.workspace(1, dev, "raster")
cnt = ws[o:o + 4 * cells].view(torch.int32).cpu().numpy().astype(np.int64)
cl = ws[o + sizes[3]: o + sizes[3] + 8 * 4 * I].view(torch.int32).reshape(-1, 2)
li_c = li[0].reshape(th * 16 if False else H, W)
# per cell max last id
Hp, Wp = th * 16, tw * 16
lip = torch.full((Hp, Wp), -1, dtype=torch.int32, device=dev); lip[:H, :W] = li[0]
cellmax = lip.reshape(th, 2, 8, tw, 2, 8).permute(0, 3, 1, 4, 2, 5).reshape(th * tw, 4, 64).amax(-1).cpu().numpy()  # [tile, wl]
offs_np = offs.reshape(-1).cpu().numpy().astype(np.int64)
ends = np.append(offs_np[1:], I)
cl_np = cl.cpu().numpy()
walked = 0; total = 0
for tidx in range(th * tw):
    st, en = offs_np[tidx], ends[tidx]
    for wl in range(4):
        c = cnt[tidx * 4 + wl]
        seg = 4 * st + wl * (en - st)
        e = cl_np[seg: seg + c, 1]
        walked += int(np.searchsorted(e, cellmax[tidx, wl], side="right"))
        total += c
print("entries up to the last contributor per cell (bwd walk, ~fwd walk):", walked, "of culled", total, "ratio", walked / max(total, 1))
alpha = ra[0, :, :, 0]
print("pixels saturated (T<1e-3):", float((alpha > 0.999).float().mean()))
