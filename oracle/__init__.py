"""ORACLE — TEST INFRASTRUCTURE ONLY.

numpy/ctypes front-end of oracle/liboracle.so (our CPU restatement of the
reference's gsplat ops + fastgs Adam, see oracle_ops.hpp) and, when present, of
oracle/_ref/*.so: the reference's own code compiled in place and run on the CPU
(tests/torch_impl.cpp; its device kernels, fastgs rasterizer and loss kernels
under ref_emul/; its COLMAP reader, splat_data.cpp host functions, strategy layer
and FusedAdam against CPU libtorch under ref_stub/).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product (lichtfeld-studio_amd/)
never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

PINHOLE, ORTHO, FISHEYE = 0, 1, 2
ROLLING_TOP_TO_BOTTOM, ROLLING_LEFT_TO_RIGHT, ROLLING_BOTTOM_TO_TOP, ROLLING_RIGHT_TO_LEFT, GLOBAL = range(5)


def build(ref: bool = True) -> None:
    """Compile liboracle.so (always) and _ref (only when /root/reference exists)."""
    subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, capture_output=True)
    if ref and os.path.exists("/root/reference/tests/torch_impl.cpp"):
        # oracle/_ref: the reference's own code compiled in place (see the Makefile header for what each target is). One parallel make - the libtorch-based
        # targets cost a minute or two of header parsing each when built from scratch. The four the core parity tests use must build; the others only serve the
        # *_reference tests, which skip without them, so their failure does not fail build().
        core = ["ref", "refk", "refk_fastgs", "refk_loss"]
        more = ["refcolmap", "refsplatio", "refstrategy", "refraster", "reflosshost", "refply", "refgsplat", "reffast", "reflink"]
        subprocess.run(["make", "-C", _HERE, "-k", "-j8", *core, *more], check=False, capture_output=True)
        subprocess.run(["make", "-C", _HERE, *core], check=True, capture_output=True)     # (up to date unless the parallel run failed: then this reports it)


def have_ref(name: str) -> bool:
    """Is oracle/_ref/<name> built? A file check only - NOTHING is dlopen'ed. The tests' import-time `skipif` marks use this, so that collecting
    the suite (also for a `-m gpu` run, which imports every module) never loads a checker library into the process."""
    return os.path.exists(os.path.join(_HERE, "_ref", name))


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        _LIB = C.CDLL(path)
        _LIB.orc_intersect_tile_count.restype = C.c_int64
    return _LIB


def ref_lib():
    """The reference's own CPU implementation, or None when it was not built."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libtorch_impl_ref.so")
        if not os.path.exists(path):
            return None
        import torch  # noqa: F401  (libtorch must be loaded first)
        _REF = C.CDLL(path)
        _REF.ref_isect_tiles.restype = C.c_int64
        _REF.ref_cpu_stage_pipeline.restype = C.c_int64
    return _REF


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _dt(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32", C.c_float
    if dtype == np.float64:
        return "f64", C.c_double
    raise TypeError(dtype)


def _c(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


def _ut(ut, dtype):
    if ut is None:
        ut = (0.1, 2.0, 0.0, 0.1, 1.0)
    return np.asarray(ut, dtype=dtype)


# ----------------------------------------------------------------------------
def quats_to_rotmats(quats, dtype=np.float32):
    sfx, _ = _dt(dtype)
    q = _c(quats, dtype)
    out = np.empty((q.shape[0], 3, 3), dtype)
    getattr(lib(), f"orc_quats_to_rotmats_{sfx}")(C.c_int64(q.shape[0]), _p(q), _p(out))
    return out


def projection_ut_3dgs_fused(means, quats, scales, opacities, viewmats0, viewmats1, Ks, width, height,
                             eps2d=0.3, near_plane=0.01, far_plane=1e4, radius_clip=0.0, calc_compensations=False,
                             camera_model=PINHOLE, ut_params=None, rs_type=GLOBAL,
                             radial_coeffs=None, tangential_coeffs=None, thin_prism_coeffs=None, dtype=np.float32):
    """Returns radii[C,N,2] i32, means2d[C,N,2], depths[C,N], conics[C,N,3], compensations|None.
    Culled entries hold 0 (the reference leaves them uninitialised)."""
    sfx, ct = _dt(dtype)
    means, quats, scales = _c(means, dtype), _c(quats, dtype), _c(scales, dtype)
    opacities = _c(opacities, dtype)
    v0, v1, Ks = _c(viewmats0, dtype), _c(viewmats1, dtype), _c(Ks, dtype)
    rad, tan, thin = _c(radial_coeffs, dtype), _c(tangential_coeffs, dtype), _c(thin_prism_coeffs, dtype)
    Cn, N = Ks.shape[0], means.shape[0]
    radii = np.zeros((Cn, N, 2), np.int32)
    means2d = np.zeros((Cn, N, 2), dtype)
    depths = np.zeros((Cn, N), dtype)
    conics = np.zeros((Cn, N, 3), dtype)
    comp = np.zeros((Cn, N), dtype) if calc_compensations else None
    ut = _ut(ut_params, dtype)
    getattr(lib(), f"orc_projection_ut_{sfx}")(
        C.c_uint32(Cn), C.c_uint32(N), _p(means), _p(quats), _p(scales), _p(opacities), _p(v0), _p(v1), _p(Ks),
        C.c_uint32(width), C.c_uint32(height), ct(eps2d), ct(near_plane), ct(far_plane), ct(radius_clip),
        C.c_int(camera_model), _p(ut), C.c_int(rs_type),
        _p(rad), C.c_int(0 if rad is None else rad.shape[-1]), _p(tan), _p(thin), C.c_int(0 if thin is None else thin.shape[-1]),
        _p(radii), _p(means2d), _p(depths), _p(conics), _p(comp))
    return radii, means2d, depths, conics, comp


def spherical_harmonics_fwd(degree, dirs, coeffs, masks=None, dtype=np.float32):
    sfx, _ = _dt(dtype)
    dirs, coeffs = _c(dirs, dtype), _c(coeffs, dtype)
    m = None if masks is None else np.ascontiguousarray(masks, np.uint8)
    N, K = dirs.size // 3, coeffs.shape[-2]
    colors = np.zeros(dirs.shape, dtype)
    getattr(lib(), f"orc_sh_fwd_{sfx}")(C.c_int64(N), C.c_int(K), C.c_int(degree), _p(dirs), _p(coeffs), _p(m), _p(colors))
    return colors


def spherical_harmonics_bwd(degree, dirs, coeffs, masks, v_colors, compute_v_dirs=True, dtype=np.float32):
    sfx, _ = _dt(dtype)
    dirs, coeffs, v_colors = _c(dirs, dtype), _c(coeffs, dtype), _c(v_colors, dtype)
    m = None if masks is None else np.ascontiguousarray(masks, np.uint8)
    N, K = dirs.size // 3, coeffs.shape[-2]
    v_coeffs = np.zeros(coeffs.shape, dtype)
    v_dirs = np.zeros(dirs.shape, dtype) if compute_v_dirs else None
    getattr(lib(), f"orc_sh_bwd_{sfx}")(C.c_int64(N), C.c_int(K), C.c_int(degree), _p(dirs), _p(coeffs), _p(m),
                                        _p(v_colors), _p(v_coeffs), _p(v_dirs))
    return v_coeffs, v_dirs


def intersect_tile(means2d, radii, depths, C_, tile_size, tile_width, tile_height, sort=True):
    """means2d [C,N,2] f32, radii [C,N,2] i32, depths [C,N] f32 ->
    tiles_per_gauss [C,N] i32, isect_ids [I] i64, flatten_ids [I] i32."""
    means2d, depths = _c(means2d, np.float32), _c(depths, np.float32)
    radii = _c(radii, np.int32)
    N = means2d.shape[1]
    tpg = np.zeros((C_, N), np.int32)
    n = lib().orc_intersect_tile_count(C.c_uint32(C_), C.c_uint32(N), _p(means2d), _p(radii),
                                       C.c_uint32(tile_size), C.c_uint32(tile_width), C.c_uint32(tile_height), _p(tpg))
    ids = np.zeros(n, np.int64)
    flat = np.zeros(n, np.int32)
    if n:
        lib().orc_intersect_tile_emit(C.c_uint32(C_), C.c_uint32(N), _p(means2d), _p(radii), _p(depths),
                                      C.c_uint32(tile_size), C.c_uint32(tile_width), C.c_uint32(tile_height),
                                      C.c_int(int(sort)), C.c_int64(n), _p(ids), _p(flat))
    return tpg, ids, flat


def intersect_offset(isect_ids, C_, tile_width, tile_height):
    ids = _c(isect_ids, np.int64)
    out = np.zeros((C_, tile_height, tile_width), np.int32)
    lib().orc_intersect_offset(C.c_int64(ids.shape[0]), _p(ids), C.c_uint32(C_), C.c_uint32(tile_width), C.c_uint32(tile_height), _p(out))
    return out


def _raster_common(means, quats, scales, colors, opacities, backgrounds, masks, viewmats0, viewmats1, Ks,
                   radial, tangential, thin, tile_offsets, flatten_ids, dtype):
    means, quats, scales = _c(means, dtype), _c(quats, dtype), _c(scales, dtype)
    colors, opacities = _c(colors, dtype), _c(opacities, dtype)
    bg = None if backgrounds is None or np.size(backgrounds) == 0 else _c(backgrounds, dtype)
    m = None if masks is None else np.ascontiguousarray(masks, np.uint8)
    v0, v1, Ks = _c(viewmats0, dtype), _c(viewmats1, dtype), _c(Ks, dtype)
    rad, tan, th = _c(radial, dtype), _c(tangential, dtype), _c(thin, dtype)
    offs, flat = _c(tile_offsets, np.int32), _c(flatten_ids, np.int32)
    return means, quats, scales, colors, opacities, bg, m, v0, v1, Ks, rad, tan, th, offs, flat


def rasterize_fwd(means, quats, scales, colors, opacities, backgrounds, masks, width, height, tile_size,
                  viewmats0, viewmats1, Ks, camera_model, rs_type, radial_coeffs, tangential_coeffs, thin_prism_coeffs,
                  tile_offsets, flatten_ids, dtype=np.float32):
    sfx, _ = _dt(dtype)
    (means, quats, scales, colors, opacities, bg, m, v0, v1, Ks, rad, tan, th, offs, flat) = _raster_common(
        means, quats, scales, colors, opacities, backgrounds, masks, viewmats0, viewmats1, Ks,
        radial_coeffs, tangential_coeffs, thin_prism_coeffs, tile_offsets, flatten_ids, dtype)
    Cn, N, CD = offs.shape[0], means.shape[0], colors.shape[-1]
    rc = np.zeros((Cn, height, width, CD), dtype)
    ra = np.zeros((Cn, height, width, 1), dtype)
    li = np.zeros((Cn, height, width), np.int32)
    getattr(lib(), f"orc_rasterize_fwd_{sfx}")(
        C.c_uint32(Cn), C.c_uint32(N), C.c_int64(flat.shape[0]), C.c_uint32(CD), _p(means), _p(quats), _p(scales),
        _p(colors), _p(opacities), _p(bg), _p(m), C.c_uint32(width), C.c_uint32(height), C.c_uint32(tile_size),
        _p(v0), _p(v1), _p(Ks), C.c_int(camera_model), C.c_int(rs_type),
        _p(rad), C.c_int(0 if rad is None else rad.shape[-1]), _p(tan), _p(th), C.c_int(0 if th is None else th.shape[-1]),
        _p(offs), _p(flat), _p(rc), _p(ra), _p(li))
    return rc, ra, li


def rasterize_bwd(means, quats, scales, colors, opacities, backgrounds, masks, width, height, tile_size,
                  viewmats0, viewmats1, Ks, camera_model, rs_type, radial_coeffs, tangential_coeffs, thin_prism_coeffs,
                  tile_offsets, flatten_ids, render_alphas, last_ids, v_render_colors, v_render_alphas, dtype=np.float32):
    sfx, _ = _dt(dtype)
    (means, quats, scales, colors, opacities, bg, m, v0, v1, Ks, rad, tan, th, offs, flat) = _raster_common(
        means, quats, scales, colors, opacities, backgrounds, masks, viewmats0, viewmats1, Ks,
        radial_coeffs, tangential_coeffs, thin_prism_coeffs, tile_offsets, flatten_ids, dtype)
    ra, li = _c(render_alphas, dtype), _c(last_ids, np.int32)
    vrc, vra = _c(v_render_colors, dtype), _c(v_render_alphas, dtype)
    Cn, N, CD = offs.shape[0], means.shape[0], colors.shape[-1]
    v_means, v_quats, v_scales = np.zeros_like(means), np.zeros_like(quats), np.zeros_like(scales)
    v_colors, v_opac = np.zeros_like(colors), np.zeros_like(opacities)
    getattr(lib(), f"orc_rasterize_bwd_{sfx}")(
        C.c_uint32(Cn), C.c_uint32(N), C.c_int64(flat.shape[0]), C.c_uint32(CD), _p(means), _p(quats), _p(scales),
        _p(colors), _p(opacities), _p(bg), _p(m), C.c_uint32(width), C.c_uint32(height), C.c_uint32(tile_size),
        _p(v0), _p(v1), _p(Ks), C.c_int(camera_model), C.c_int(rs_type),
        _p(rad), C.c_int(0 if rad is None else rad.shape[-1]), _p(tan), _p(th), C.c_int(0 if th is None else th.shape[-1]),
        _p(offs), _p(flat), _p(ra), _p(li), _p(vrc), _p(vra),
        _p(v_means), _p(v_quats), _p(v_scales), _p(v_colors), _p(v_opac))
    return v_means, v_quats, v_scales, v_colors, v_opac


def relocation(opacities, scales, ratios, binoms, n_max, dtype=np.float32):
    sfx, _ = _dt(dtype)
    o, s, b = _c(opacities, dtype), _c(scales, dtype), _c(binoms, dtype)
    r = _c(ratios, np.int32)
    no, ns = np.zeros_like(o), np.zeros_like(s)
    getattr(lib(), f"orc_relocation_{sfx}")(C.c_int64(o.shape[0]), _p(o), _p(s), _p(r), _p(b), C.c_int(n_max), _p(no), _p(ns))
    return no, ns


def add_noise(raw_opacities, raw_scales, raw_quats, noise, means, current_lr, dtype=np.float32):
    """Returns the updated means (the op itself works in place)."""
    sfx, ct = _dt(dtype)
    o, s, q, n = _c(raw_opacities, dtype), _c(raw_scales, dtype), _c(raw_quats, dtype), _c(noise, dtype)
    m = np.array(means, dtype=dtype, copy=True, order="C")
    getattr(lib(), f"orc_add_noise_{sfx}")(C.c_int64(o.shape[0]), _p(o), _p(s), _p(q), _p(n), _p(m), ct(current_lr))
    return m


def adam_step(param, exp_avg, exp_avg_sq, grad, lr, beta1, beta2, eps, bc1_rcp, bc2_sqrt_rcp, dtype=np.float32):
    """Returns (param, exp_avg, exp_avg_sq) after one step (copies)."""
    sfx, ct = _dt(dtype)
    p = np.array(param, dtype=dtype, copy=True, order="C")
    m = np.array(exp_avg, dtype=dtype, copy=True, order="C")
    v = np.array(exp_avg_sq, dtype=dtype, copy=True, order="C")
    g = _c(grad, dtype)
    getattr(lib(), f"orc_adam_step_{sfx}")(C.c_int64(p.size), _p(p), _p(m), _p(v), _p(g),
                                           ct(lr), ct(beta1), ct(beta2), ct(eps), ct(bc1_rcp), ct(bc2_sqrt_rcp))
    return p, m, v


def pixel_rays(width, height, viewmat0, viewmat1, K, camera_model=PINHOLE, rs_type=GLOBAL,
               radial=None, tangential=None, thin_prism=None, dtype=np.float32):
    sfx, _ = _dt(dtype)
    v0, v1, K = _c(viewmat0, dtype), _c(viewmat1, dtype), _c(K, dtype)
    rad, tan, th = _c(radial, dtype), _c(tangential, dtype), _c(thin_prism, dtype)
    o = np.zeros((height, width, 3), dtype)
    d = np.zeros((height, width, 3), dtype)
    valid = np.zeros((height, width), np.uint8)
    getattr(lib(), f"orc_pixel_rays_{sfx}")(
        C.c_uint32(width), C.c_uint32(height), _p(v0), _p(v1), _p(K), C.c_int(camera_model), C.c_int(rs_type),
        _p(rad), C.c_int(0 if rad is None else rad.shape[-1]), _p(tan), _p(th), C.c_int(0 if th is None else th.shape[-1]),
        _p(o), _p(d), _p(valid))
    return o, d, valid.astype(bool)


# ---- the reference's own CPU code (oracle/_ref) -------------------------------
def ref_quat_to_rotmat(quats):
    q = _c(quats, np.float32)
    out = np.empty((q.shape[0], 3, 3), np.float32)
    ref_lib().ref_quat_to_rotmat(C.c_int64(q.shape[0]), _p(q), _p(out))
    return out


def ref_spherical_harmonics(degree, dirs, coeffs):
    d, c = _c(dirs, np.float32), _c(coeffs, np.float32)
    out = np.empty((d.shape[0], 3), np.float32)
    ref_lib().ref_spherical_harmonics(C.c_int64(d.shape[0]), C.c_int(c.shape[1]), C.c_int(degree), _p(d), _p(c), _p(out))
    return out


def ref_isect_tiles(means2d, radii, depths, tile_size, tile_width, tile_height, sort=True):
    m, d, r = _c(means2d, np.float32), _c(depths, np.float32), _c(radii, np.int32)
    Cn, N = m.shape[0], m.shape[1]
    tpg = np.zeros((Cn, N), np.int32)
    args = (C.c_int64(Cn), C.c_int64(N), _p(m), _p(r), _p(d), C.c_int(tile_size), C.c_int(tile_width), C.c_int(tile_height), C.c_int(int(sort)))
    n = ref_lib().ref_isect_tiles(*args, _p(tpg), None, None, C.c_int64(-1))
    ids, flat = np.zeros(n, np.int64), np.zeros(n, np.int32)
    ref_lib().ref_isect_tiles(*args, _p(tpg), _p(ids), _p(flat), C.c_int64(n))
    return tpg, ids, flat


def ref_fully_fused_projection(means, quats, scales, viewmat, K, width, height, eps2d=0.3, near_plane=0.01, far_plane=1e10):
    m, q, s = _c(means, np.float32), _c(quats, np.float32), _c(scales, np.float32)
    v, K = _c(viewmat, np.float32), _c(K, np.float32)
    N = m.shape[0]
    radii, m2 = np.zeros((N, 2), np.int32), np.zeros((N, 2), np.float32)
    d, c = np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    ref_lib().ref_fully_fused_projection(C.c_int64(N), _p(m), _p(q), _p(s), _p(v), _p(K), C.c_int(width), C.c_int(height),
                                         C.c_float(eps2d), C.c_float(near_plane), C.c_float(far_plane), _p(radii), _p(m2), _p(d), _p(c))
    return radii, m2, d, c


def ref_cpu_stage_fwd_bwd(means, quats, scales, coeffs, degree, viewmat, K, width, height, with_isect, threads, repeats, tile_size=16):
    """the reference's tests/torch_impl.cpp covariance + EWA projection + SH (+ tile intersection), forward + autograd backward, `repeats`
    times on `threads` libtorch threads -> (seconds per repeat [repeats], n_isects)"""
    m, q, s, c = _c(means, np.float32), _c(quats, np.float32), _c(scales, np.float32), _c(coeffs, np.float32)
    v, K = _c(viewmat, np.float32), _c(K, np.float32)
    sec = np.zeros(repeats, np.float64)
    lib_ = ref_lib()
    lib_.ref_cpu_stage_fwd_bwd.restype = C.c_int64
    n = lib_.ref_cpu_stage_fwd_bwd(C.c_int64(m.shape[0]), C.c_int(c.shape[1]), C.c_int(degree), _p(m), _p(q), _p(s), _p(c), _p(v), _p(K), C.c_int(width), C.c_int(height),
                                   C.c_int(tile_size), C.c_int(1 if with_isect else 0), C.c_int(threads), C.c_int(repeats), sec.ctypes.data_as(C.c_void_p))
    return sec, int(n)


# ---- fastgs (EWA) rasterizer (oracle_fastgs.hpp) --------------------------------
def _fg_args(dt, means, scales_raw, rot_raw, opac_raw, sh0, sh_rest, w2c, cam_pos, active_sh_bases, W, H, fx, fy, cx, cy, near, far):
    cf = C.c_float if dt == np.float32 else C.c_double
    arrs = [_c(x, dt) for x in (means, scales_raw, rot_raw, np.asarray(opac_raw).reshape(-1), sh0, sh_rest, np.asarray(w2c).reshape(4, 4), np.asarray(cam_pos).reshape(3))]
    N = arrs[0].shape[0]
    total_rest = arrs[5].shape[1] if arrs[5].ndim == 3 else 0
    return arrs, N, total_rest, [C.c_int64(N)] + [_p(a) for a in arrs] + [C.c_int(active_sh_bases), C.c_int(total_rest), C.c_int(W), C.c_int(H),
                                                                           cf(fx), cf(fy), cf(cx), cf(cy), cf(near), cf(far)]


def fastgs_forward(means, scales_raw, rot_raw, opac_raw, sh0, sh_rest, w2c, cam_pos, active_sh_bases, W, H, fx, fy, cx, cy,
                   near=0.01, far=1e10, dtype=np.float32):
    """-> dict(image [3,H,W], alpha [H,W], n_contrib [H,W], mean2d, conic_opacity, color, n_touched, bounds, offsets, ids).
    The forward keeps its state inside the library for fastgs_backward (same dtype, called right after)."""
    dt = np.dtype(dtype).type
    sfx = "f32" if dt == np.float32 else "f64"
    arrs, N, total_rest, args = _fg_args(dt, means, scales_raw, rot_raw, opac_raw, sh0, sh_rest, w2c, cam_pos, active_sh_bases, W, H, fx, fy, cx, cy, near, far)
    out = dict(image=np.zeros((3, H, W), dt), alpha=np.zeros((H, W), dt), n_contrib=np.zeros((H, W), np.int32), mean2d=np.zeros((N, 2), dt),
               conic_opacity=np.zeros((N, 4), dt), color=np.zeros((N, 3), dt), n_touched=np.zeros(N, np.int32), bounds=np.zeros((N, 4), np.int32))
    f = getattr(lib(), f"orc_fastgs_forward_{sfx}")
    f.restype = C.c_int64
    n_inst = f(*args, _p(out["image"]), _p(out["alpha"]), _p(out["n_contrib"]), _p(out["mean2d"]), _p(out["conic_opacity"]), _p(out["color"]),
               _p(out["n_touched"]), _p(out["bounds"]))
    T = ((W + 15) // 16) * ((H + 15) // 16)
    out["offsets"], out["ids"] = np.zeros(T + 1, np.int32), np.zeros(max(n_inst, 1), np.int32)
    getattr(lib(), f"orc_fastgs_lists_{sfx}")(_p(out["offsets"]), _p(out["ids"]))
    out["ids"] = out["ids"][:n_inst]
    out["_keep"] = arrs
    return out


def fastgs_backward(fwd, means, scales_raw, rot_raw, opac_raw, sh0, sh_rest, w2c, cam_pos, active_sh_bases, W, H, fx, fy, cx, cy,
                    g_image, g_alpha, near=0.01, far=1e10, dtype=np.float32, densification_info=None):
    """-> (g_means, g_scales_raw, g_rot_raw, g_opac_raw, g_sh0, g_sh_rest, densification_info [2,N])."""
    dt = np.dtype(dtype).type
    sfx = "f32" if dt == np.float32 else "f64"
    arrs, N, total_rest, args = _fg_args(dt, means, scales_raw, rot_raw, opac_raw, sh0, sh_rest, w2c, cam_pos, active_sh_bases, W, H, fx, fy, cx, cy, near, far)
    gi, ga = _c(g_image, dt), _c(np.asarray(g_alpha).reshape(H, W), dt)
    img, al, nc = _c(fwd["image"], dt), _c(fwd["alpha"], dt), _c(fwd["n_contrib"], np.int32)
    g = [np.zeros((N, 3), dt), np.zeros((N, 3), dt), np.zeros((N, 4), dt), np.zeros(N, dt), np.zeros((N, 1, 3), dt), np.zeros((N, max(total_rest, 0), 3), dt)]
    dens = np.zeros((2, N), dt) if densification_info is None else _c(densification_info, dt).copy()
    getattr(lib(), f"orc_fastgs_backward_{sfx}")(*args, _p(img), _p(al), _p(nc), _p(gi), _p(ga), *[_p(x) for x in g], _p(dens))
    return (*g, dens)


# ---- the reference's own DEVICE kernels on the CPU (oracle/_ref/libref_kernels.so; ref_kernels.cpp + ref_emul/) -------------------
_REFK = None


REFK_WHOLE_LIBRARY = False   # True: the refk_* calls below go to libref_raster_full.so, whose refk_* entry points call the reference's own gsplat:: operators
#                              (gsplat/*.cpp over gsplat/*.cu with their launch functions, ref_gsplat_capi.cpp) instead of the restated launch sequences


def refk_lib():
    """gsplat/*.cu + fastgs Adam kernels of the reference, compiled in place as host code (`make -C oracle refk`); None when absent."""
    global _REFK
    if REFK_WHOLE_LIBRARY:
        return ref_raster_lib(full=True)
    if _REFK is None:
        path = os.path.join(_HERE, "_ref", "libref_kernels.so")
        if not os.path.exists(path):
            return None
        _REFK = C.CDLL(path)
    return _REFK


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _coeffs(camera_model, radial, tangential, thin_prism):
    """the kernels read fixed-size arrays per camera: pinhole 6 radial / 2 tangential / 4 thin-prism, fisheye 4 radial (Fwd.cu:100-121)"""
    def pad(a, n):
        if a is None:
            return None
        a = _f32(a)
        out = np.zeros((a.shape[0], n), np.float32)
        out[:, :min(n, a.shape[1])] = a[:, :n]
        return out
    if camera_model == FISHEYE:
        return pad(radial, 4), None, None
    return pad(radial, 6), pad(tangential, 2), pad(thin_prism, 4)


def refk_projection_ut(means, quats, scales, opacities, viewmats0, viewmats1, Ks, width, height, eps2d=0.3, near_plane=0.01, far_plane=1e4,
                       radius_clip=0.0, calc_compensations=False, camera_model=PINHOLE, ut_params=None, rs_type=GLOBAL,
                       radial_coeffs=None, tangential_coeffs=None, thin_prism_coeffs=None):
    means, quats, scales, opacities = _f32(means), _f32(quats), _f32(scales), _f32(opacities)
    v0, v1, Ks = _f32(viewmats0), _f32(viewmats1), _f32(Ks)
    rad, tan, thin = _coeffs(camera_model, radial_coeffs, tangential_coeffs, thin_prism_coeffs)
    Cn, N = Ks.shape[0], means.shape[0]
    radii = np.zeros((Cn, N, 2), np.int32)
    means2d, depths, conics = np.zeros((Cn, N, 2), np.float32), np.zeros((Cn, N), np.float32), np.zeros((Cn, N, 3), np.float32)
    comp = np.zeros((Cn, N), np.float32) if calc_compensations else None
    ut = _ut(ut_params, np.float32)
    refk_lib().refk_projection_ut(C.c_uint32(Cn), C.c_uint32(N), _p(means), _p(quats), _p(scales), _p(opacities), _p(v0), _p(v1), _p(Ks),
                                  C.c_uint32(width), C.c_uint32(height), C.c_float(eps2d), C.c_float(near_plane), C.c_float(far_plane),
                                  C.c_float(radius_clip), C.c_int(camera_model), _p(ut), C.c_int(rs_type), _p(rad), _p(tan), _p(thin),
                                  _p(radii), _p(means2d), _p(depths), _p(conics), _p(comp))
    return radii, means2d, depths, conics, comp


def _refk_raster_args(means, quats, scales, colors, opacities, backgrounds, masks, width, height, tile_size, viewmats0, viewmats1, Ks,
                      camera_model, rs_type, radial, tangential, thin, tile_offsets, flatten_ids, ut_params):
    means, quats, scales, colors, opacities = _f32(means), _f32(quats), _f32(scales), _f32(colors), _f32(opacities)
    bg = None if backgrounds is None or np.size(backgrounds) == 0 else _f32(backgrounds)
    m = None if masks is None else np.ascontiguousarray(masks, np.uint8)
    v0, v1, Ks = _f32(viewmats0), _f32(viewmats1), _f32(Ks)
    rad, tan, th_ = _coeffs(camera_model, radial, tangential, thin)
    offs, flat = _c(tile_offsets, np.int32), _c(flatten_ids, np.int32)
    Cn, N, CD = offs.shape[0], means.shape[0], colors.shape[-1]
    ut = _ut(ut_params, np.float32)
    keep = (means, quats, scales, colors, opacities, bg, m, v0, v1, Ks, rad, tan, th_, offs, flat, ut)
    args = [C.c_uint32(CD), C.c_uint32(Cn), C.c_uint32(N), C.c_uint32(flat.shape[0]), _p(means), _p(quats), _p(scales), _p(colors), _p(opacities), _p(bg),
            _p(m), C.c_uint32(width), C.c_uint32(height), C.c_uint32(tile_size), C.c_uint32(offs.shape[2]), C.c_uint32(offs.shape[1]), _p(v0), _p(v1), _p(Ks),
            C.c_int(camera_model), _p(ut), C.c_int(rs_type), _p(rad), _p(tan), _p(th_), _p(offs), _p(flat)]
    return keep, args, Cn, N, CD


def refk_rasterize_fwd(means, quats, scales, colors, opacities, backgrounds, masks, width, height, tile_size, viewmats0, viewmats1, Ks,
                       camera_model, rs_type, radial_coeffs, tangential_coeffs, thin_prism_coeffs, tile_offsets, flatten_ids, ut_params=None):
    keep, args, Cn, N, CD = _refk_raster_args(means, quats, scales, colors, opacities, backgrounds, masks, width, height, tile_size, viewmats0, viewmats1,
                                              Ks, camera_model, rs_type, radial_coeffs, tangential_coeffs, thin_prism_coeffs, tile_offsets, flatten_ids, ut_params)
    rc, ra, li = np.zeros((Cn, height, width, CD), np.float32), np.zeros((Cn, height, width, 1), np.float32), np.zeros((Cn, height, width), np.int32)
    rcode = refk_lib().refk_rasterize_fwd(*args, _p(rc), _p(ra), _p(li))
    assert rcode == 0
    return rc, ra, li


def refk_rasterize_bwd(means, quats, scales, colors, opacities, backgrounds, masks, width, height, tile_size, viewmats0, viewmats1, Ks,
                       camera_model, rs_type, radial_coeffs, tangential_coeffs, thin_prism_coeffs, tile_offsets, flatten_ids,
                       render_alphas, last_ids, v_render_colors, v_render_alphas, ut_params=None):
    keep, args, Cn, N, CD = _refk_raster_args(means, quats, scales, colors, opacities, backgrounds, masks, width, height, tile_size, viewmats0, viewmats1,
                                              Ks, camera_model, rs_type, radial_coeffs, tangential_coeffs, thin_prism_coeffs, tile_offsets, flatten_ids, ut_params)
    ra, li, vrc, vra = _f32(render_alphas), _c(last_ids, np.int32), _f32(v_render_colors), _f32(v_render_alphas)
    v_means, v_quats, v_scales = np.zeros((N, 3), np.float32), np.zeros((N, 4), np.float32), np.zeros((N, 3), np.float32)
    v_colors, v_opac = np.zeros((Cn, N, CD), np.float32), np.zeros((Cn, N), np.float32)
    rcode = refk_lib().refk_rasterize_bwd(*args, _p(ra), _p(li), _p(vrc), _p(vra), _p(v_means), _p(v_quats), _p(v_scales), _p(v_colors), _p(v_opac))
    assert rcode == 0
    return v_means, v_quats, v_scales, v_colors, v_opac


def refk_relocation(opacities, scales, ratios, binoms, n_max):
    o, s, b, r = _f32(opacities).copy(), _f32(scales).copy(), _f32(binoms).copy(), np.ascontiguousarray(ratios, np.int32).copy()
    no, ns = np.zeros_like(o), np.zeros_like(s)
    refk_lib().refk_relocation(C.c_int64(o.shape[0]), _p(o), _p(s), _p(r), _p(b), C.c_int(n_max), _p(no), _p(ns))
    return no, ns


def refk_add_noise(raw_opacities, raw_scales, raw_quats, noise, means, current_lr):
    o, s, q, n = _f32(raw_opacities).copy(), _f32(raw_scales).copy(), _f32(raw_quats).copy(), _f32(noise).copy()
    m = np.array(means, dtype=np.float32, copy=True, order="C")
    refk_lib().refk_add_noise(C.c_int64(o.shape[0]), _p(o), _p(s), _p(q), _p(n), _p(m), C.c_float(current_lr))
    return m


def refk_quats_to_rotmats(quats):
    q = _f32(quats)
    out = np.zeros((q.shape[0], 3, 3), np.float32)
    refk_lib().refk_quats_to_rotmats(C.c_int64(q.shape[0]), _p(q), _p(out))
    return out


def refk_adam_step(param, exp_avg, exp_avg_sq, grad, lr, beta1, beta2, eps, bc1_rcp, bc2_sqrt_rcp):
    p, m, v = (np.array(x, dtype=np.float32, copy=True, order="C") for x in (param, exp_avg, exp_avg_sq))
    g = _f32(grad)
    refk_lib().refk_adam_step(C.c_int64(p.size), _p(p), _p(m), _p(v), _p(g), C.c_float(lr), C.c_float(beta1), C.c_float(beta2), C.c_float(eps),
                              C.c_float(bc1_rcp), C.c_float(bc2_sqrt_rcp))
    return p, m, v


# ---- the reference's own fastgs rasterizer on the CPU (oracle/_ref/libref_fastgs.so; ref_kernels_fastgs.cpp + ref_emul/) ---------------
_REFK_FG = None


REFK_FASTGS_WHOLE = False   # True: refk_fastgs_fwd_bwd goes through the reference's own forward_wrapper / backward_wrapper (rasterization_api.cu, libref_fast_raster.so)
_REF_FAST = None


def ref_fast_raster_lib():
    """the reference's fastgs training render path end to end (`make -C oracle reffast`: fast_rasterize + autograd + rasterization_api.cu + forward.cu / backward.cu)"""
    global _REF_FAST
    if _REF_FAST is None:
        path = os.path.join(_HERE, "_ref", "libref_fast_raster.so")
        if not os.path.exists(path):
            return None
        _REF_FAST = C.CDLL(path)
    return _REF_FAST


def ref_fast_render_backward(means, sh0, shN, scaling, rotation, opacity, sh_degree, active_sh_degree, R, T, fx, fy, cx, cy, width, height, bg, v_image, v_alpha,
                             densification_info=None):
    """gs::training::fast_rasterize() of the reference for one pinhole camera + backward of sum(image * v_image) + sum(alpha * v_alpha) -> dict(image, alpha,
    g_means, g_sh0, g_shN, g_scaling, g_rotation, g_opacity [, densification_info])"""
    arrs = [_f32(x) for x in (means, sh0, shN, scaling, rotation, opacity)]
    N, K1 = arrs[0].shape[0], arrs[2].shape[1]
    R, T, bg, v_image, v_alpha = _f32(R), _f32(T), _f32(bg), _f32(v_image), _f32(v_alpha)
    dens = None if densification_info is None else _f32(densification_info).copy()
    out = dict(image=np.empty((3, height, width), np.float32), alpha=np.empty((1, height, width), np.float32), g_means=np.empty((N, 3), np.float32),
               g_sh0=np.empty((N, 1, 3), np.float32), g_shN=np.empty((N, K1, 3), np.float32), g_scaling=np.empty((N, 3), np.float32),
               g_rotation=np.empty((N, 4), np.float32), g_opacity=np.empty(N, np.float32))
    rc = ref_fast_raster_lib().reffast_render_backward(
        C.c_int64(N), C.c_int64(K1), C.c_int(sh_degree), C.c_int(active_sh_degree), *[_p(a) for a in arrs], _p(R), _p(T), C.c_float(fx), C.c_float(fy), C.c_float(cx),
        C.c_float(cy), C.c_int(width), C.c_int(height), _p(bg), _p(v_image), _p(v_alpha), None if dens is None else _p(dens), _p(out["image"]), _p(out["alpha"]),
        *[_p(out[k]) for k in ("g_means", "g_sh0", "g_shN", "g_scaling", "g_rotation", "g_opacity")])
    if rc:
        raise RuntimeError("reffast_render_backward failed")
    if dens is not None:
        out["densification_info"] = dens
    return out


def refk_fastgs_lib():
    """fastgs/rasterization of the reference (forward.cu, backward.cu and the kernels in its headers), compiled in place as host code
    (`make -C oracle refk_fastgs`); None when absent."""
    global _REFK_FG
    if _REFK_FG is None:
        path = os.path.join(_HERE, "_ref", "libref_fastgs.so")
        if not os.path.exists(path):
            return None
        _REFK_FG = C.CDLL(path)
    return _REFK_FG


def refk_fastgs_fwd_bwd(means, scales_raw, rot_raw, opac_raw, sh0, sh_rest, w2c, cam_pos, active_sh_bases, W, H, fx, fy, cx, cy, g_image=None, g_alpha=None,
                        near=0.01, far=1e10, densification_info=None):
    """forward (+ backward when g_image is given) of the reference's fastgs rasterizer -> dict(image [3,H,W], alpha [1,H,W], counts (n_visible,
    n_instances, n_buckets), and with gradients: g_means, g_scales_raw, g_rot_raw, g_opac_raw [N], g_sh0 [N,1,3], g_sh_rest, densification_info)."""
    means, scales_raw, rot_raw, opac_raw = _f32(means), _f32(scales_raw), _f32(rot_raw), _f32(np.asarray(opac_raw).reshape(-1))
    sh0, sh_rest, w2c, cam_pos = _f32(sh0), _f32(sh_rest), _f32(np.asarray(w2c).reshape(4, 4)), _f32(np.asarray(cam_pos).reshape(3))
    N, total_rest = means.shape[0], (sh_rest.shape[1] if sh_rest.ndim == 3 else 0)
    out = dict(image=np.zeros((3, H, W), np.float32), alpha=np.zeros((1, H, W), np.float32), counts=np.zeros(3, np.int32))
    g = None
    if g_image is not None:
        g_image, g_alpha = _f32(g_image), _f32(np.asarray(g_alpha).reshape(1, H, W))
        g = dict(g_means=np.zeros((N, 3), np.float32), g_scales_raw=np.zeros((N, 3), np.float32), g_rot_raw=np.zeros((N, 4), np.float32),
                 g_opac_raw=np.zeros(N, np.float32), g_sh0=np.zeros((N, 1, 3), np.float32), g_sh_rest=np.zeros((N, max(total_rest, 0), 3), np.float32))
        if densification_info is not None:
            g["densification_info"] = _f32(densification_info).copy()
    entry = ref_fast_raster_lib().reffast_wrappers if REFK_FASTGS_WHOLE else refk_fastgs_lib().refk_fastgs_fwd_bwd
    entry(
        C.c_int(N), C.c_int(active_sh_bases), C.c_int(total_rest), C.c_int(W), C.c_int(H), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy),
        C.c_float(near), C.c_float(far), _p(means), _p(scales_raw), _p(rot_raw), _p(opac_raw), _p(sh0), _p(sh_rest), _p(w2c), _p(cam_pos),
        _p(g_image) if g is not None else None, _p(g_alpha) if g is not None else None, _p(out["image"]), _p(out["alpha"]),
        *([_p(g[k]) for k in ("g_means", "g_scales_raw", "g_rot_raw", "g_opac_raw", "g_sh0", "g_sh_rest")] if g is not None else [None] * 6),
        _p(g["densification_info"]) if (g is not None and "densification_info" in g) else None, _p(out["counts"]))
    if g is not None:
        out.update(g)
    return out


# ---- the reference's own fused-SSIM and bilateral-grid kernels on the CPU (oracle/_ref/libref_loss.so; ref_kernels_loss.cpp + ref_emul/) -------
_REFK_LOSS = None


def refk_loss_lib():
    global _REFK_LOSS
    if _REFK_LOSS is None:
        path = os.path.join(_HERE, "_ref", "libref_loss.so")
        if not os.path.exists(path):
            return None
        _REFK_LOSS = C.CDLL(path)
    return _REFK_LOSS


def refk_fusedssim(C1, C2, img1, img2, train=True):
    """fusedssim of the reference (ssim.cu): img [B,CH,H,W] -> (ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12)"""
    img1, img2 = _f32(img1), _f32(img2)
    B, CH, H, W = img1.shape
    out = [np.zeros_like(img1) for _ in range(4)]
    refk_loss_lib().refk_fusedssim(C.c_int(B), C.c_int(CH), C.c_int(H), C.c_int(W), C.c_float(C1), C.c_float(C2), _p(img1), _p(img2), C.c_int(int(train)),
                                   *[_p(x) for x in out])
    return tuple(out)


def refk_fusedssim_backward(C1, C2, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12):
    img1, img2, dL_dmap = _f32(img1), _f32(img2), _f32(dL_dmap)
    B, CH, H, W = img1.shape
    out = np.zeros_like(img1)
    refk_loss_lib().refk_fusedssim_backward(C.c_int(B), C.c_int(CH), C.c_int(H), C.c_int(W), C.c_float(C1), C.c_float(C2), _p(img1), _p(img2), _p(dL_dmap),
                                            _p(_f32(dm_dmu1)), _p(_f32(dm_dsigma1_sq)), _p(_f32(dm_dsigma12)), _p(out))
    return out


def refk_bilateral_slice(grid, rgb, grad_output=None):
    """slice_forward (+ slice_backward with grad_output) of the reference: grid [12,L,H,W], rgb [h,w,3] -> out [, grad_grid, grad_rgb]"""
    grid, rgb = _f32(grid), _f32(rgb)
    _, L, H, W = grid.shape
    h, w = rgb.shape[:2]
    out = np.zeros_like(rgb)
    refk_loss_lib().refk_bilateral_slice_forward(_p(grid), _p(rgb), _p(out), C.c_int(L), C.c_int(H), C.c_int(W), C.c_int(h), C.c_int(w))
    if grad_output is None:
        return out
    gg, gr = np.zeros_like(grid), np.zeros_like(rgb)
    refk_loss_lib().refk_bilateral_slice_backward(_p(grid), _p(rgb), _p(_f32(grad_output)), _p(gg), _p(gr), C.c_int(L), C.c_int(H), C.c_int(W), C.c_int(h), C.c_int(w))
    return out, gg, gr


def refk_bilateral_tv(grids, grad_output=1.0):
    """tv_loss_forward / _backward of the reference: grids [N,12,L,H,W] -> (loss, grad_grids)"""
    grids = _f32(grids)
    N, _, L, H, W = grids.shape
    loss, gg = np.zeros(1, np.float32), np.zeros_like(grids)
    refk_loss_lib().refk_bilateral_tv_forward(_p(grids), _p(loss), C.c_int(N), C.c_int(L), C.c_int(H), C.c_int(W))
    refk_loss_lib().refk_bilateral_tv_backward(_p(grids), C.c_float(grad_output), _p(gg), C.c_int(N), C.c_int(L), C.c_int(H), C.c_int(W))
    return float(loss[0]), gg


# ---- the reference's own COLMAP reader (oracle/_ref/libref_colmap.so; ref_colmap_shim.cpp + src/loader/formats/colmap.cpp compiled in place) -------------
_REF_COLMAP = None


class _RefView(C.Structure):
    _fields_ = [("camera_id", C.c_uint32), ("colmap_model", C.c_int32), ("camera_model_type", C.c_int32), ("width", C.c_uint64), ("height", C.c_uint64),
                ("focal_x", C.c_float), ("focal_y", C.c_float), ("center_x", C.c_float), ("center_y", C.c_float), ("R", C.c_float * 9), ("T", C.c_float * 3),
                ("n_radial", C.c_int32), ("radial", C.c_float * 6), ("n_tangential", C.c_int32), ("tangential", C.c_float * 2),
                ("n_params", C.c_int32), ("params", C.c_float * 12)]


def ref_colmap_lib():
    global _REF_COLMAP
    if _REF_COLMAP is None:
        path = os.path.join(_HERE, "_ref", "libref_colmap.so")
        if not os.path.exists(path):
            return None
        lib = C.CDLL(path)
        lib.refcolmap_last_error.restype = C.c_char_p
        lib.refcolmap_image_name.restype = C.c_char_p
        lib.refcolmap_image_path.restype = C.c_char_p
        lib.refcolmap_num_views.restype = C.c_uint64
        lib.refcolmap_points.restype = C.c_int64
        _REF_COLMAP = lib
    return _REF_COLMAP


def ref_colmap_cameras(base, images_folder="images", text=False):
    """read_colmap_cameras_and_images[_text] of the reference -> (list of dicts in oracle.colmap_io.assemble's vocabulary, scene centre); raises RuntimeError
    with the reference's exception text."""
    lib = ref_colmap_lib()
    h = C.c_void_p()
    if lib.refcolmap_open(os.fsencode(base), images_folder.encode(), C.c_int(int(text)), C.byref(h)):
        raise RuntimeError(lib.refcolmap_last_error().decode())
    try:
        out, v = [], _RefView()
        for i in range(lib.refcolmap_num_views(h)):
            if lib.refcolmap_view_at(h, C.c_uint64(i), C.byref(v)):
                raise RuntimeError(lib.refcolmap_last_error().decode())
            out.append(dict(camera_id=v.camera_id, colmap_model=v.colmap_model, camera_model_type=v.camera_model_type, width=int(v.width), height=int(v.height),
                            focal_x=np.float32(v.focal_x), focal_y=np.float32(v.focal_y), center_x=np.float32(v.center_x), center_y=np.float32(v.center_y),
                            R=np.array(v.R, np.float32).reshape(3, 3), T=np.array(v.T, np.float32), radial=np.array(v.radial[:v.n_radial], np.float32),
                            tangential=np.array(v.tangential[:v.n_tangential], np.float32), params=np.array(v.params[:v.n_params], np.float32),
                            name=lib.refcolmap_image_name(h, C.c_uint64(i)).decode(), path=os.fsdecode(lib.refcolmap_image_path(h, C.c_uint64(i)))))
        c = (C.c_float * 3)()
        lib.refcolmap_scene_center(h, c)
        return out, np.array(c, np.float32)
    finally:
        lib.refcolmap_close(h)


def ref_colmap_points(base, text=False):
    lib = ref_colmap_lib()
    n = lib.refcolmap_points(os.fsencode(base), C.c_int(int(text)), None, None, C.c_int64(0))
    if n < 0:
        raise RuntimeError(lib.refcolmap_last_error().decode())
    pos, col = np.empty((n, 3), np.float32), np.empty((n, 3), np.uint8)
    lib.refcolmap_points(os.fsencode(base), C.c_int(int(text)), pos.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p), C.c_int64(n))
    return pos, col


# ---- compute_mean_neighbor_distances and write_ply_impl of the reference's splat_data.cpp (oracle/_ref/libref_splat_io.so; ref_splat_io_shim.cpp) --------
_REF_SPLAT_IO = None


def ref_splat_io_lib():
    global _REF_SPLAT_IO
    if _REF_SPLAT_IO is None:
        path = os.path.join(_HERE, "_ref", "libref_splat_io.so")
        if not os.path.exists(path):
            return None
        _REF_SPLAT_IO = C.CDLL(path)
    return _REF_SPLAT_IO


def ref_mean_neighbor_distances(points):
    points = _f32(points)
    out = np.empty(points.shape[0], np.float32)
    ref_splat_io_lib().refsplat_mean_neighbor_distances(C.c_int64(points.shape[0]), _p(points), _p(out))
    return out


def ref_write_ply(root, stem, means, sh0, shN, opacity, scaling, rotation):
    """SplatData::save_ply of the reference for the raw parameter tensors (sh0 [N,K0,3], shN [N,KN,3]) -> the bytes of <root>/<stem>.ply"""
    means, sh0, shN, opacity, scaling, rotation = [_f32(x) for x in (means, sh0, shN, opacity, scaling, rotation)]
    N = means.shape[0]
    rc = ref_splat_io_lib().refsplat_write_ply(os.fsencode(root), stem.encode(), C.c_int64(N), C.c_int64(sh0.shape[1]), C.c_int64(shN.shape[1]), _p(means), _p(sh0),
                                               _p(shN), _p(opacity), _p(scaling), _p(rotation))
    if rc:
        raise RuntimeError("refsplat_write_ply failed")
    return open(os.path.join(root, stem + ".ply"), "rb").read()


# ---- the reference's own strategy layer + FusedAdam on CPU libtorch (oracle/_ref/libref_strategy.so; ref_strategy_shim.cpp) ---------------------------------
_REF_STRATEGY = None


class RefParams(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("iterations", "sh_degree_interval", "refine_every", "start_refine", "stop_refine", "reset_every", "pause_refine_after_reset")] + \
               [(k, C.c_float) for k in ("means_lr", "shs_lr", "opacity_lr", "scaling_lr", "rotation_lr", "min_opacity", "grad_threshold", "prune_opacity", "grow_scale3d",
                                         "prune_scale3d", "opacity_reg", "scale_reg", "lambda_dssim", "init_opacity", "init_scaling")] + \
               [(k, C.c_int32) for k in ("max_cap", "revised_opacity", "sh_degree")]


def ref_strategy_lib():
    global _REF_STRATEGY
    if _REF_STRATEGY is None:
        path = os.path.join(_HERE, "_ref", "libref_strategy.so")
        if not os.path.exists(path):
            return None
        lib = C.CDLL(path)
        lib.refstrat_last_error.restype = C.c_char_p
        lib.refstrat_log_name.restype = C.c_char_p
        lib.refstrat_lr.restype = C.c_double
        for f in ("refstrat_size", "refstrat_get", "refstrat_step_count", "refstrat_log_count", "refstrat_log_numel"):
            getattr(lib, f).restype = C.c_int64
        _REF_STRATEGY = lib
    return _REF_STRATEGY


def ref_strategy_default_params():
    """gs::param::OptimizationParameters{} of the reference (include/core/parameters.hpp) -> dict"""
    p = RefParams()
    ref_strategy_lib().refstrat_default_params(C.byref(p))
    return {k: getattr(p, k) for k, _ in RefParams._fields_}


class RefStrategy:
    """gs::training::MCMC (kind 'mcmc') or DefaultStrategy ('default') of the reference, initialised from raw parameter arrays: means [N,3], sh0 [N,1,3],
    shN [N,K,3], scaling [N,3], rotation [N,4], opacity [N] (raw). State is read back as numpy arrays in the param-group order of strategy_utils.cpp:35-40."""
    NAMES = ("means", "sh0", "shN", "scaling", "rotation", "opacity")

    def __init__(self, kind, means, sh0, shN, scaling, rotation, opacity, scene_scale, sh_degree, **params):
        self.lib = ref_strategy_lib()
        p = RefParams()
        self.lib.refstrat_default_params(C.byref(p))
        for k, v in params.items():
            assert hasattr(p, k), k
            setattr(p, k, v)
        arrs = [_f32(x) for x in (means, sh0, shN, scaling, rotation, opacity)]
        N, K = arrs[0].shape[0], arrs[2].shape[1]
        self.h = C.c_void_p()
        self._ok(self.lib.refstrat_create(C.c_int({"mcmc": 0, "default": 1}[kind]), C.c_int64(N), C.c_int64(K), C.c_int(sh_degree), *[_p(a) for a in arrs],
                                          C.c_float(scene_scale), C.byref(p), C.byref(self.h)))

    def _ok(self, rc):
        if rc:
            raise RuntimeError(self.lib.refstrat_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.refstrat_destroy(self.h)
            self.h = None

    def size(self):
        return self.lib.refstrat_size(self.h)

    def get(self, which, what=0):
        """what: 0 parameter, 1 exp_avg, 2 exp_avg_sq -> flat float32 array, or None when the optimizer has no state for it yet"""
        n = self.lib.refstrat_get(self.h, C.c_int(which), C.c_int(what), None)
        if n < 0:
            return None
        out = np.empty(n, np.float32)
        self.lib.refstrat_get(self.h, C.c_int(which), C.c_int(what), _p(out))
        return out

    def state(self):
        out = {}
        for i, name in enumerate(self.NAMES):
            out[name] = self.get(i, 0)
            for what, tag in ((1, "exp_avg"), (2, "exp_avg_sq")):
                a = self.get(i, what)
                if a is not None:
                    out[f"{name}.{tag}"] = a
            out[f"{name}.step"] = np.int64(self.lib.refstrat_step_count(self.h, C.c_int(i)))
        out["lr"] = np.array([self.lib.refstrat_lr(self.h, C.c_int(g)) for g in range(6)])
        out["active_sh_degree"] = np.int64(self.lib.refstrat_active_sh_degree(self.h))
        return out

    def set_grads(self, grads):
        arrs = [_f32(g) for g in grads]
        ptrs = (C.c_void_p * 6)(*[a.ctypes.data for a in arrs])
        self._ok(self.lib.refstrat_set_grads(self.h, ptrs))

    def set_densification_info(self, info):
        info = _f32(info)
        assert info.shape == (2, self.size())
        self._ok(self.lib.refstrat_set_densification_info(self.h, _p(info)))

    def step(self, it):
        self._ok(self.lib.refstrat_step(self.h, C.c_int(it)))

    def is_refining(self, it):
        return bool(self.lib.refstrat_is_refining(self.h, C.c_int(it)))

    def post_backward(self, it, seed):
        """-> the random draws the call made, in order: [(name, array)]"""
        self._ok(self.lib.refstrat_post_backward(self.h, C.c_int(it), C.c_uint64(seed)))
        draws = []
        for i in range(self.lib.refstrat_log_count()):
            n = self.lib.refstrat_log_numel(C.c_int64(i))
            a = np.empty(n, np.int64 if self.lib.refstrat_log_is_int64(C.c_int64(i)) else np.float32)
            self.lib.refstrat_log_copy(C.c_int64(i), a.ctypes.data_as(C.c_void_p))
            draws.append((self.lib.refstrat_log_name(C.c_int64(i)).decode(), a))
        return draws

    def remove_gaussians(self, mask):
        m = np.ascontiguousarray(mask, np.uint8)
        self._ok(self.lib.refstrat_remove_gaussians(self.h, m.ctypes.data_as(C.c_void_p)))


def ref_init_model_from_pointcloud(positions, colors_u8, scene_center, sh_degree=3, init_scaling=1.0, init_opacity=0.1):
    """SplatData::init_model_from_pointcloud of the reference (splat_data.cpp:508-614, non-random branch) -> dict of the six raw parameter arrays + scene_scale"""
    positions, center = _f32(positions), _f32(scene_center)
    colors = np.ascontiguousarray(colors_u8, np.uint8)
    N, K = positions.shape[0], (sh_degree + 1) ** 2 - 1
    out = dict(means=np.empty((N, 3), np.float32), sh0=np.empty((N, 1, 3), np.float32), shN=np.empty((N, K, 3), np.float32), scaling=np.empty((N, 3), np.float32),
               rotation=np.empty((N, 4), np.float32), opacity=np.empty((N, 1), np.float32))
    scale = C.c_float()
    rc = ref_splat_io_lib().refsplat_init_model(C.c_int64(N), _p(positions), colors.ctypes.data_as(C.c_void_p), _p(center), C.c_int(sh_degree), C.c_float(init_scaling),
                                                C.c_float(init_opacity), *[_p(out[k]) for k in ("means", "sh0", "shN", "scaling", "rotation", "opacity")], C.byref(scale))
    if rc:
        raise RuntimeError("refsplat_init_model failed")
    out["scene_scale"] = np.float32(scale.value)
    return out


# ---- the reference's own training-time render path on the CPU (oracle/_ref/libref_raster.so; ref_raster_shim.cpp) ---------------------------------------------
_REF_RASTER = None


def ref_raster_lib(full=False):
    """full=False: libref_raster.so (the reference's render path over the restated launch sequences of ref_raster_shim.cpp / ref_kernels.cpp);
    full=True: libref_raster_full.so (`make -C oracle refgsplat`: the same over the reference's WHOLE gsplat library, nothing of the operator layer restated)"""
    global _REF_RASTER
    if full:
        path = os.path.join(_HERE, "_ref", "libref_raster_full.so")
        if not os.path.exists(path):
            return None
        if "full" not in _REF_RASTER_FULL:
            _REF_RASTER_FULL["full"] = C.CDLL(path)
        return _REF_RASTER_FULL["full"]
    if _REF_RASTER is None:
        path = os.path.join(_HERE, "_ref", "libref_raster.so")
        if not os.path.exists(path):
            return None
        _REF_RASTER = C.CDLL(path)
    return _REF_RASTER


_REF_RASTER_FULL = {}


def ref_links_gpu_lib():
    """oracle/_ref/libref_links_gpu.so (`make -C oracle reflink`): the reference's own L2 code (rasterizer.cpp, rasterizer_autograd.cpp, camera.cpp, bilateral_grid.cpp,
    fused_adam.cpp, compiled unmodified against its own gsplat/Ops.h + ROCm libtorch) LINKED to the product backend lichtfeld-studio_amd/liblfs_gsplat_torch.so. Same C
    entry points as libref_raster_full.so, tensors on cuda:0; the link-level drop-in proof of tests/test_gpu_reference_links.py. None if not built."""
    if "gpu" not in _REF_RASTER_FULL:
        path = os.path.join(_HERE, "_ref", "libref_links_gpu.so")
        if not os.path.exists(path):
            return None
        import torch  # noqa: F401  (libtorch / libtorch_hip must be in the process before the library's dependencies resolve)
        _REF_RASTER_FULL["gpu"] = C.CDLL(path)
    return _REF_RASTER_FULL["gpu"]


def ref_links_fused_adam_steps(params, grads, lrs, iteration0, n_steps):
    """gs::training::FusedAdam (built as strategy_utils.cpp:20-48 builds it) stepping `n_steps` times from `iteration0` over the linked backend's adam_step_wrapper:
    six float arrays each for params / grads -> (params, exp_avg, exp_avg_sq, step_counts)."""
    lib = ref_links_gpu_lib()
    p = [_f32(x).reshape(-1).copy() for x in params]
    g = [_f32(x).reshape(-1) for x in grads]
    m, v = [np.zeros_like(x) for x in p], [np.zeros_like(x) for x in p]
    sizes = (C.c_int64 * 6)(*[x.size for x in p])
    ptrs = lambda arrs: (C.c_void_p * 6)(*[a.ctypes.data for a in arrs])
    counts = (C.c_int64 * 6)()
    rc = lib.reflink_fused_adam_steps(sizes, ptrs(p), ptrs(g), ptrs(m), ptrs(v), (C.c_double * 6)(*[float(x) for x in lrs]), C.c_int(iteration0), C.c_int(n_steps), counts)
    if rc:
        raise RuntimeError("reflink_fused_adam_steps failed")
    return p, m, v, list(counts)


def ref_links_mse_train_steps(mode, means, sh0, shN, scaling, rotation, opacity, sh_degree, active_sh_degree, R, T, fx, fy, cx, cy, width, height, bg, gt_image, lrs,
                              iteration0, n_steps, timed_from=-1):
    """`n_steps` MSE training steps on the reference's own SplatData / Camera / FusedAdam objects in libref_links_gpu.so: mode 0 = the reference's sequence (rasterize()
    -> mse_loss -> backward -> FusedAdam::step -> zero_grad), mode 1 = INTEGRATION.md §1b's patch (optimizer state read with FusedAdam's real members, the step as one
    lfs::GutTrainStep::step call) -> dict(params [6], exp_avg [6], exp_avg_sq [6], losses [n_steps], n_isects)"""
    lib = ref_links_gpu_lib()
    p = [_f32(x).copy() for x in (means, sh0, shN, scaling, rotation, opacity)]
    N, K1 = p[0].shape[0], p[2].shape[1]
    m, v = [np.zeros_like(x) for x in p], [np.zeros_like(x) for x in p]
    R, T, gt = _f32(R), _f32(T), _f32(gt_image)
    bg = None if bg is None else _f32(bg)
    ptrs = lambda arrs: (C.c_void_p * 6)(*[a.ctypes.data for a in arrs])
    losses, n_isects, ms = np.zeros(n_steps, np.float32), C.c_int64(-1), C.c_double(0.0)
    rc = lib.reflink_mse_train_steps(C.c_int(mode), C.c_int64(N), C.c_int64(K1), C.c_int(sh_degree), C.c_int(active_sh_degree), ptrs(p), _p(R), _p(T), C.c_float(fx),
                                     C.c_float(fy), C.c_float(cx), C.c_float(cy), C.c_int(width), C.c_int(height), None if bg is None else _p(bg), _p(gt),
                                     (C.c_double * 6)(*[float(x) for x in lrs]), C.c_int(iteration0), C.c_int(n_steps), _p(losses), ptrs(m), ptrs(v), C.byref(n_isects),
                                     C.c_int(timed_from), C.byref(ms))
    if rc:
        raise RuntimeError("reflink_mse_train_steps failed")
    return dict(params=p, exp_avg=m, exp_avg_sq=v, losses=losses, n_isects=n_isects.value, ms_per_step=(ms.value if timed_from >= 0 else None))


def ref_render_backward(means, sh0, shN, scaling, rotation, opacity, sh_degree, active_sh_degree, R, T, fx, fy, cx, cy, width, height, bg, v_image, v_alpha=None,
                        full=False, lib=None):
    """gs::training::rasterize() of the reference for one pinhole camera + backward of sum(image * v_image) [+ sum(alpha * v_alpha)] through its autograd
    Functions: raw parameters as SplatData holds them (sh0 [N,1,3], shN [N,K,3], opacity [N]) -> dict(image [3,H,W], alpha [1,H,W], radii [N], viewmat [4,4],
    K [3,3], g_means, g_sh0, g_shN, g_scaling, g_rotation, g_opacity)"""
    arrs = [_f32(x) for x in (means, sh0, shN, scaling, rotation, opacity)]
    N, K1 = arrs[0].shape[0], arrs[2].shape[1]
    R, T, v_image = _f32(R), _f32(T), _f32(v_image)
    bg = None if bg is None else _f32(bg)
    v_alpha = None if v_alpha is None else _f32(v_alpha)
    out = dict(image=np.empty((3, height, width), np.float32), alpha=np.empty((1, height, width), np.float32), radii=np.empty(N, np.int32),
               g_means=np.empty((N, 3), np.float32), g_sh0=np.empty((N, 1, 3), np.float32), g_shN=np.empty((N, K1, 3), np.float32), g_scaling=np.empty((N, 3), np.float32),
               g_rotation=np.empty((N, 4), np.float32), g_opacity=np.empty(N, np.float32), viewmat=np.empty((4, 4), np.float32), K=np.empty((3, 3), np.float32))
    rc = (lib or ref_raster_lib(full)).refraster_render_backward(
        C.c_int64(N), C.c_int64(K1), C.c_int(sh_degree), C.c_int(active_sh_degree), *[_p(a) for a in arrs], _p(R), _p(T), C.c_float(fx), C.c_float(fy), C.c_float(cx),
        C.c_float(cy), C.c_int(width), C.c_int(height), C.c_int(width), C.c_int(height), None if bg is None else _p(bg), _p(v_image),
        None if v_alpha is None else _p(v_alpha), _p(out["image"]), _p(out["alpha"]), out["radii"].ctypes.data_as(C.c_void_p),
        *[_p(out[k]) for k in ("g_means", "g_sh0", "g_shN", "g_scaling", "g_rotation", "g_opacity", "viewmat", "K")])
    if rc:
        raise RuntimeError("refraster_render_backward failed")
    return out


# ---- the reference's loss-side host code on CPU libtorch (oracle/_ref/libref_loss_host.so; ref_loss_host_shim.cpp) --------------------------------------------
_REF_LOSS_HOST = None


def ref_loss_host_lib():
    global _REF_LOSS_HOST
    if _REF_LOSS_HOST is None:
        path = os.path.join(_HERE, "_ref", "libref_loss_host.so")
        if not os.path.exists(path):
            return None
        _REF_LOSS_HOST = C.CDLL(path)
    return _REF_LOSS_HOST


def ref_fused_ssim(img1, img2, padding="valid"):
    """fused_ssim(img1, img2, padding, train=true) of the reference's fused_ssim.cuh on [3,H,W] images -> (mean SSIM, d/d img1)"""
    a, b = _f32(img1), _f32(img2)
    v, g = C.c_float(), np.empty_like(a)
    assert ref_loss_host_lib().reflh_fused_ssim(C.c_int(a.shape[1]), C.c_int(a.shape[2]), _p(a), _p(b), C.c_int(int(padding == "valid")), C.byref(v), _p(g)) == 0
    return np.float32(v.value), g


def ref_photometric_loss(rendered, gt, lambda_dssim=0.2):
    """Trainer::compute_photometric_loss (trainer.cpp:103-130) on [3,H,W] images -> (loss, d loss / d rendered)"""
    a, b = _f32(rendered), _f32(gt)
    v, g = C.c_float(), np.empty_like(a)
    assert ref_loss_host_lib().reflh_photometric(C.c_int(a.shape[1]), C.c_int(a.shape[2]), _p(a), _p(b), C.c_float(lambda_dssim), C.byref(v), _p(g)) == 0
    return np.float32(v.value), g


def ref_bilateral_grid(num_images, gW, gH, gL, image_idx, delta, rgb, v_out, tv_weight):
    """BilateralGrid(num_images, gW, gH, gL) of the reference with grids = identity + delta: apply(rgb [3,h,w], image_idx) and tv_loss(), and the gradients of
    sum(out * v_out) + tv_weight * tv -> dict(identity, out, tv, g_grids, g_rgb)"""
    delta, rgb, v_out = _f32(delta), _f32(rgb), _f32(v_out)
    assert delta.shape == (num_images, 12, gL, gH, gW)
    out = dict(identity=np.empty_like(delta), out=np.empty_like(rgb), g_grids=np.empty_like(delta), g_rgb=np.empty_like(rgb))
    tv = C.c_float()
    assert ref_loss_host_lib().reflh_bilateral(C.c_int(num_images), C.c_int(gW), C.c_int(gH), C.c_int(gL), C.c_int(image_idx), _p(delta), C.c_int(rgb.shape[1]),
                                               C.c_int(rgb.shape[2]), _p(rgb), _p(v_out), C.c_float(tv_weight), _p(out["identity"]), _p(out["out"]), C.byref(tv),
                                               _p(out["g_grids"]), _p(out["g_rgb"])) == 0
    out["tv"] = np.float32(tv.value)
    return out


def ref_warmup_schedule(lr0, gamma, warmup_steps, warmup_start_factor, n):
    """WarmupExponentialLR of the reference (scheduler.cpp:27-63) over one parameter group: the learning rate after each of n steps (float64)"""
    lrs = np.empty(n, np.float64)
    assert ref_loss_host_lib().reflh_warmup_schedule(C.c_double(lr0), C.c_double(gamma), C.c_int(warmup_steps), C.c_double(warmup_start_factor), C.c_int(n),
                                                     lrs.ctypes.data_as(C.c_void_p)) == 0
    return lrs


# ---- SH and tile-intersection kernels of the reference on the CPU (libref_kernels.so: SphericalHarmonicsCUDA.cu, IntersectTile.cu under ref_emul/) ---------
def refk_sh_fwd(degree, dirs, coeffs, masks=None):
    dirs, coeffs = _f32(dirs), _f32(coeffs)
    N, K = dirs.shape[0], coeffs.shape[1]
    m = None if masks is None else np.ascontiguousarray(masks, np.uint8)
    colors = np.zeros((N, 3), np.float32)             # at::empty in the reference: masked-out rows are not written
    refk_lib().refk_sh_fwd(C.c_uint32(N), C.c_uint32(K), C.c_uint32(degree), _p(dirs), _p(coeffs), None if m is None else m.ctypes.data_as(C.c_void_p), _p(colors))
    return colors


def refk_sh_bwd(degree, dirs, coeffs, masks, v_colors, compute_v_dirs=True):
    dirs, coeffs, v_colors = _f32(dirs), _f32(coeffs), _f32(v_colors)
    N, K = dirs.shape[0], coeffs.shape[1]
    m = None if masks is None else np.ascontiguousarray(masks, np.uint8)
    v_coeffs, v_dirs = np.zeros_like(coeffs), np.zeros_like(dirs)
    refk_lib().refk_sh_bwd(C.c_uint32(N), C.c_uint32(K), C.c_uint32(degree), _p(dirs), _p(coeffs), None if m is None else m.ctypes.data_as(C.c_void_p), _p(v_colors),
                           _p(v_coeffs), _p(v_dirs) if compute_v_dirs else None)
    return v_coeffs, (v_dirs if compute_v_dirs else None)


def refk_intersect_tile(means2d, radii, depths, tile_size, tile_width, tile_height, sort=True):
    """gsplat::intersect_tile (non-packed) over the reference's kernels -> (tiles_per_gauss int32 [C,N], isect_ids int64, flatten_ids int32)"""
    means2d, depths = _f32(means2d), _f32(depths)
    radii = np.ascontiguousarray(radii, np.int32)
    Cn, N = depths.shape
    lib = refk_lib()
    lib.refk_intersect_tile.restype = C.c_int64
    tpg = np.zeros((Cn, N), np.int32)
    args = (C.c_uint32(Cn), C.c_uint32(N), _p(means2d), radii.ctypes.data_as(C.c_void_p), _p(depths), C.c_uint32(tile_size), C.c_uint32(tile_width),
            C.c_uint32(tile_height), C.c_int(int(sort)), tpg.ctypes.data_as(C.c_void_p))
    n = lib.refk_intersect_tile(*args, None, None, C.c_int64(0))
    ids, flat = np.empty(n, np.int64), np.empty(n, np.int32)
    if n:
        lib.refk_intersect_tile(*args, ids.ctypes.data_as(C.c_void_p), flat.ctypes.data_as(C.c_void_p), C.c_int64(n))
    return tpg, ids, flat


def refk_intersect_offset(isect_ids, Cn, tile_width, tile_height):
    ids = np.ascontiguousarray(isect_ids, np.int64)
    off = np.empty((Cn, tile_height, tile_width), np.int32)
    refk_lib().refk_intersect_offset(C.c_int64(len(ids)), ids.ctypes.data_as(C.c_void_p), C.c_uint32(Cn), C.c_uint32(tile_width), C.c_uint32(tile_height),
                                     off.ctypes.data_as(C.c_void_p))
    return off


# ---- the reference's own PLY reader on CPU libtorch (oracle/_ref/libref_ply.so; ref_ply_shim.cpp) --------------------------------------------------------------
_REF_PLY = None


def ref_ply_lib():
    global _REF_PLY
    if _REF_PLY is None:
        path = os.path.join(_HERE, "_ref", "libref_ply.so")
        if not os.path.exists(path):
            return None
        lib = C.CDLL(path)
        lib.refply_last_error.restype = C.c_char_p
        lib.refply_load.restype = C.c_int64
        lib.refply_get.restype = C.c_int64
        lib.refply_scene_scale.restype = C.c_float
        _REF_PLY = lib
    return _REF_PLY


def ref_load_ply(path):
    """gs::loader::load_ply of the reference -> dict(means, sh0, shN, scaling, rotation, opacity (raw tensors as SplatData holds them), sh_degree, scene_scale);
    raises RuntimeError with the reference's message"""
    lib = ref_ply_lib()
    if lib.refply_load(os.fsencode(path)) < 0:
        raise RuntimeError(lib.refply_last_error().decode())
    out = {}
    for i, name in enumerate(("means", "sh0", "shN", "scaling", "rotation", "opacity")):
        shape = (C.c_int64 * 3)()
        n = lib.refply_get(C.c_int(i), shape, None)
        a = np.empty(n, np.float32)
        lib.refply_get(C.c_int(i), shape, _p(a))
        dims = [int(s) for s in shape]
        while len(dims) > 1 and dims[-1] == 0 and int(np.prod(dims[:-1])) == n:
            dims.pop()
        out[name] = a.reshape(dims) if int(np.prod(dims)) == n else a
    out["sh_degree"], out["scene_scale"] = lib.refply_sh_degree(), lib.refply_scene_scale()
    return out


def ref_train_loss_backward(means, sh0, shN, scaling, rotation, opacity, sh_degree, active_sh_degree, R, T, fx, fy, cx, cy, width, height, bg, gt_image, lambda_dssim,
                            scale_reg, opacity_reg, bilateral=None, lib=None):
    """The loss of one training step as Trainer::train_step composes it - rasterize() [-> BilateralGrid::apply] -> L1 + D-SSIM (fused_ssim "valid") -> + scale /
    opacity regularisers [+ tv_weight * tv_loss()] - and its gradients w.r.t. the six raw tensors [and the grids], over the reference's whole gsplat library, its own
    ssim.cu and bilateral-grid code (libref_raster_full.so) -> dict(loss, image, g_* [, g_grids]). bilateral = dict(n_images, gW, gH, gL, image_idx, delta
    [n_images,12,gL,gH,gW] added to the identity grids, tv_weight)."""
    arrs = [_f32(x) for x in (means, sh0, shN, scaling, rotation, opacity)]
    N, K1 = arrs[0].shape[0], arrs[2].shape[1]
    R, T, gt = _f32(R), _f32(T), _f32(gt_image)
    bg = None if bg is None else _f32(bg)
    out = dict(image=np.empty((3, height, width), np.float32), g_means=np.empty((N, 3), np.float32), g_sh0=np.empty((N, 1, 3), np.float32),
               g_shN=np.empty((N, K1, 3), np.float32), g_scaling=np.empty((N, 3), np.float32), g_rotation=np.empty((N, 4), np.float32), g_opacity=np.empty(N, np.float32))
    b = bilateral or dict(n_images=0, gW=0, gH=0, gL=0, image_idx=0, delta=None, tv_weight=0.0)
    delta = None if b["delta"] is None else _f32(b["delta"])
    if delta is not None:
        out["g_grids"] = np.empty_like(delta)
    loss = C.c_float()
    rc = (lib or ref_raster_lib(full=True)).refraster_train_loss_backward(
        C.c_int64(N), C.c_int64(K1), C.c_int(sh_degree), C.c_int(active_sh_degree), *[_p(a) for a in arrs], _p(R), _p(T), C.c_float(fx), C.c_float(fy), C.c_float(cx),
        C.c_float(cy), C.c_int(width), C.c_int(height), None if bg is None else _p(bg), _p(gt), C.c_float(lambda_dssim), C.c_float(scale_reg), C.c_float(opacity_reg),
        C.byref(loss), _p(out["image"]), *[_p(out[k]) for k in ("g_means", "g_sh0", "g_shN", "g_scaling", "g_rotation", "g_opacity")],
        C.c_int(b["n_images"]), C.c_int(b["gW"]), C.c_int(b["gH"]), C.c_int(b["gL"]), C.c_int(b["image_idx"]), None if delta is None else _p(delta), C.c_float(b["tv_weight"]),
        _p(out["g_grids"]) if delta is not None else None)
    if rc:
        raise RuntimeError("refraster_train_loss_backward failed")
    out["loss"] = np.float32(loss.value)
    return out


def ref_fast_train_loss_backward(means, sh0, shN, scaling, rotation, opacity, sh_degree, active_sh_degree, R, T, fx, fy, cx, cy, width, height, bg, gt_image, lambda_dssim,
                                 densification_info=None):
    """The photometric loss of one training step on the reference's default path - fast_rasterize() -> L1 + D-SSIM (fused_ssim "valid", image not clamped) - and
    its gradients, over its own fastgs code and ssim.cu (libref_fast_raster.so) -> dict(loss, g_* [, densification_info])"""
    arrs = [_f32(x) for x in (means, sh0, shN, scaling, rotation, opacity)]
    N, K1 = arrs[0].shape[0], arrs[2].shape[1]
    R, T, gt, bg = _f32(R), _f32(T), _f32(gt_image), _f32(bg)
    dens = None if densification_info is None else _f32(densification_info).copy()
    out = dict(g_means=np.empty((N, 3), np.float32), g_sh0=np.empty((N, 1, 3), np.float32), g_shN=np.empty((N, K1, 3), np.float32), g_scaling=np.empty((N, 3), np.float32),
               g_rotation=np.empty((N, 4), np.float32), g_opacity=np.empty(N, np.float32))
    loss = C.c_float()
    rc = ref_fast_raster_lib().reffast_train_loss_backward(
        C.c_int64(N), C.c_int64(K1), C.c_int(sh_degree), C.c_int(active_sh_degree), *[_p(a) for a in arrs], _p(R), _p(T), C.c_float(fx), C.c_float(fy), C.c_float(cx),
        C.c_float(cy), C.c_int(width), C.c_int(height), _p(bg), _p(gt), C.c_float(lambda_dssim), None if dens is None else _p(dens), C.byref(loss),
        *[_p(out[k]) for k in ("g_means", "g_sh0", "g_shN", "g_scaling", "g_rotation", "g_opacity")])
    if rc:
        raise RuntimeError("reffast_train_loss_backward failed")
    out["loss"] = np.float32(loss.value)
    if dens is not None:
        out["densification_info"] = dens
    return out
