"""Developer script: run every op on the GPU, compare with the oracle, print max deviations.
(tests/ holds the asserted versions; this prints raw numbers used to set tolerances.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
import lichtfeld_studio_amd as lfs
from lichtfeld_studio_amd import ops, scenes

dev = torch.device("cuda:0")
t = lambda a, dt=torch.float32: None if a is None else torch.as_tensor(np.ascontiguousarray(a)).to(dev).to(dt)
n = lambda x: x.detach().cpu().numpy()

def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))

def run(sh_degree=3, N=10000, W=256, H=256, bg=True, seed=42):
    print(f"==== N={N} {W}x{H} sh_degree={sh_degree} bg={bg}")
    sc = scenes.syn_a(seed=seed, n=N, sh_degree=sh_degree)
    means = sc.means.numpy(); quats = sc.raw_quats.numpy()
    scales = sc.raw_scales.exp().numpy(); opac = torch.sigmoid(sc.raw_opacities).numpy()
    sh = torch.cat([sc.sh0, sc.shN], 1).numpy()
    vm = sc.viewmats.numpy(); K = sc.Ks.numpy().copy()
    K[0, 0, 2] = W / 2; K[0, 1, 2] = H / 2
    # ---- projection
    o_r, o_m2, o_d, o_c, _ = oracle.projection_ut_3dgs_fused(means, quats, scales, opac, vm, None, K, W, H)
    g = ops.projection_ut_3dgs_fused(t(means), t(quats), t(scales), t(opac), t(vm), None, t(K), W, H, 0.3, 0.01, 1e4, 0.0, False, lfs.CameraModelType.PINHOLE)
    g_r, g_m2, g_d, g_c = [n(x) for x in g[:4]]
    vis = (o_r > 0).all(-1)
    print("proj: radii mismatch", int((g_r != o_r).sum()), "of", o_r.size, "visible", int(vis.sum()),
          "means2d", np.abs(g_m2 - o_m2).max(), "depths", np.abs(g_d - o_d).max(), "conics", np.abs(g_c - o_c).max())
    # ---- SH
    dirs = means - np.linalg.inv(vm[0])[:3, 3]
    Kc = sh.shape[1]
    o_col = oracle.spherical_harmonics_fwd(sh_degree, dirs, sh, vis[0])
    g_col = n(ops.spherical_harmonics_fwd(sh_degree, t(dirs), t(sh), t(vis[0], torch.bool)))
    print("sh fwd:", np.abs(g_col - o_col).max())
    vcol = np.random.default_rng(1).standard_normal(o_col.shape).astype(np.float32)
    o_vc, o_vd = oracle.spherical_harmonics_bwd(sh_degree, dirs, sh, vis[0], vcol, True)
    g_vc, g_vd = ops.spherical_harmonics_bwd(Kc, sh_degree, t(dirs), t(sh), t(vis[0], torch.bool), t(vcol), True)
    print("sh bwd: v_coeffs", np.abs(n(g_vc) - o_vc).max(), "v_dirs", np.abs(n(g_vd) - o_vd).max(), "scale", np.abs(o_vd).max())
    # ---- intersect (feed the oracle's projection to both)
    ts = 16; tw = (W + ts - 1) // ts; th = (H + ts - 1) // ts
    o_tpg, o_ids, o_flat = oracle.intersect_tile(o_m2, o_r, o_d, 1, ts, tw, th, True)
    g_tpg, g_ids, g_flat, g_off = ops.intersect_tile(t(o_m2), t(o_r, torch.int32), t(o_d), None, None, 1, ts, tw, th, True, return_offsets=True)
    print("isect sorted: n", len(o_ids), g_ids.numel(), "tpg", bool((n(g_tpg) == o_tpg).all()),
          "ids", bool(g_ids.numel() == len(o_ids) and (n(g_ids) == o_ids).all()), "flat", bool(g_ids.numel() == len(o_ids) and (n(g_flat) == o_flat).all()))
    o_off = oracle.intersect_offset(o_ids, 1, tw, th)
    g_off2 = ops.intersect_offset(g_ids, 1, tw, th)
    print("offsets: free", bool((n(g_off) == o_off).all()), "standalone", bool((n(g_off2) == o_off).all()))
    u_tpg, u_ids, u_flat = oracle.intersect_tile(o_m2, o_r, o_d, 1, ts, tw, th, False)
    gu = ops.intersect_tile(t(o_m2), t(o_r, torch.int32), t(o_d), None, None, 1, ts, tw, th, False)
    print("isect unsorted: ids", bool((n(gu[1]) == u_ids).all()), "flat", bool((n(gu[2]) == u_flat).all()))
    # ---- raster fwd
    colors = np.maximum(o_col + 0.5, 0)[None].astype(np.float32)
    colors[~np.isfinite(colors)] = 0
    bgc = np.array([[0.1, 0.2, 0.3]], np.float32) if bg else None
    t0 = time.time()
    o_rc, o_ra, o_li = oracle.rasterize_fwd(means, quats, scales, colors, opac[None], bgc, None, W, H, ts, vm, None, K, 0, 4, None, None, None, o_off, o_flat)
    t1 = time.time()
    g_rc, g_ra, g_li = ops.rasterize_to_pixels_from_world_3dgs_fwd(t(means), t(quats), t(scales), t(colors), t(opac[None]), t(bgc), None, W, H, ts,
        t(vm), None, t(K), lfs.CameraModelType.PINHOLE, None, lfs.ShutterType.GLOBAL, None, None, None, t(o_off, torch.int32), t(o_flat, torch.int32))
    d = np.abs(n(g_rc) - o_rc)
    print(f"raster fwd: oracle {t1-t0:.2f}s  rgb max {d.max():.3e} mean {d.mean():.3e} frac>1e-3 {(d > 1e-3).mean():.2e}  alpha max {np.abs(n(g_ra) - o_ra).max():.3e}  last_ids equal {(n(g_li) == o_li).mean():.6f}")
    # ---- raster bwd (feed oracle fwd outputs to both)
    rng = np.random.default_rng(2)
    vrc = rng.standard_normal(o_rc.shape).astype(np.float32); vra = rng.standard_normal(o_ra.shape).astype(np.float32)
    og = oracle.rasterize_bwd(means, quats, scales, colors, opac[None], bgc, None, W, H, ts, vm, None, K, 0, 4, None, None, None, o_off, o_flat, o_ra, o_li, vrc, vra)
    og64 = oracle.rasterize_bwd(means, quats, scales, colors, opac[None], bgc, None, W, H, ts, vm, None, K, 0, 4, None, None, None, o_off, o_flat, o_ra, o_li, vrc, vra, dtype=np.float64)
    gg = ops.rasterize_to_pixels_from_world_3dgs_bwd(t(means), t(quats), t(scales), t(colors), t(opac[None]), t(bgc), None, W, H, ts,
        t(vm), None, t(K), lfs.CameraModelType.PINHOLE, None, lfs.ShutterType.GLOBAL, None, None, None, t(o_off, torch.int32), t(o_flat, torch.int32),
        t(o_ra), t(o_li, torch.int32), t(vrc), t(vra))
    for name, a, b, c in zip(["v_means", "v_quats", "v_scales", "v_colors", "v_opac"], gg, og, og64):
        print(f"raster bwd {name}: rel-L2 hip-vs-f32 {rel(n(a), b):.3e}  hip-vs-f64 {rel(n(a), c):.3e}  f32-vs-f64 {rel(b, c):.3e}  max|.| {np.abs(b).max():.3e}")

run(sh_degree=3)
run(sh_degree=0, bg=False)
run(sh_degree=1, N=3000, W=200, H=120, seed=7)

# ---- small ops
rng = np.random.default_rng(3)
q = rng.standard_normal((5000, 4)).astype(np.float32)
print("quats_to_rotmats", np.abs(n(ops.quats_to_rotmats(t(q))) - oracle.quats_to_rotmats(q)).max())
o = (rng.random(5000) * 0.9 + 0.05).astype(np.float32); s = (rng.random((5000, 3)) * 0.1).astype(np.float32)
ratios = rng.integers(1, 52, 5000).astype(np.int32)
binoms = np.zeros((51, 51), np.float32)
import math
for a in range(51):
    for b in range(a + 1): binoms[a, b] = math.comb(a, b)
ono, ons = oracle.relocation(o, s, ratios, binoms, 51)
gno, gns = ops.relocation(t(o), t(s), t(ratios, torch.int32), t(binoms), 51)
print("relocation: opac", np.abs(n(gno) - ono).max(), "scales rel", np.nanmax(np.abs(n(gns) - ons) / (np.abs(ons) + 1e-12)))
ro = rng.standard_normal(5000).astype(np.float32) * 2; rs = (np.log(0.02) + 0.4 * rng.standard_normal((5000, 3))).astype(np.float32)
nz = rng.standard_normal((5000, 3)).astype(np.float32); m = rng.standard_normal((5000, 3)).astype(np.float32)
om = oracle.add_noise(ro, rs, q, nz, m, 1e-3 * 5e5)
gm = t(m).clone(); ops.add_noise(t(ro), t(rs), t(q), t(nz), gm, 1e-3 * 5e5)
print("add_noise", np.abs(n(gm) - om).max(), "delta scale", np.abs(om - m).max())
p = rng.standard_normal(100003).astype(np.float32); m1 = rng.standard_normal(100003).astype(np.float32) * 0.1
v1 = (rng.random(100003) * 0.01).astype(np.float32); gr = rng.standard_normal(100003).astype(np.float32)
op_, om_, ov_ = oracle.adam_step(p, m1, v1, gr, 1e-3, 0.9, 0.999, 1e-15, 1 / (1 - 0.9 ** 7), 1 / math.sqrt(1 - 0.999 ** 7))
gp, gm1, gv1 = t(p).clone(), t(m1).clone(), t(v1).clone()
ops.adam_step_wrapper(gp, gm1, gv1, t(gr), 1e-3, 0.9, 0.999, 1e-15, 1 / (1 - 0.9 ** 7), 1 / math.sqrt(1 - 0.999 ** 7))
print("adam: exact param", bool((n(gp) == op_).all()), "m", bool((n(gm1) == om_).all()), "v", bool((n(gv1) == ov_).all()), "max", np.abs(n(gp) - op_).max())
print("DONE")
