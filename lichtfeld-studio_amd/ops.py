"""Python mirror of `namespace gsplat` (/root/reference/gsplat/Ops.h:12-166) and of
`fast_gs::optimizer::adam_step_wrapper` (/root/reference/fastgs/optimizer/include/adam_api.h:11-21).

Same names, argument order, argument meaning, output shapes/dtypes and error behaviour as
the reference's libtorch functions; every call goes through the C ABI of liblfs_gsplat.so on
the current HIP stream.  Differences that are deliberate are documented per function
(SURVEY.md §7 "reference quirks").
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import time

import torch

from . import capi
from .capi import (CameraModelType, LfsError, ShutterType, UnscentedTransformParameters, cameras_struct, check,
                   load_library, ptr, require_gpu, stream, ut_struct, workspace)

Tensor = torch.Tensor


def _opt(t: Optional[Tensor]) -> Optional[Tensor]:
    """at::optional / undefined tensor / the empty {0} tensor the L2 code passes (rasterizer.cpp:300-303)."""
    if t is None or t.numel() == 0:
        return None
    return t


# -----------------------------------------------------------------------------------------
# Spherical harmonics (Ops.h:12-25, SphericalHarmonics.cpp:15-76)
# -----------------------------------------------------------------------------------------
def spherical_harmonics_fwd(degrees_to_use: int, dirs: Tensor, coeffs: Tensor, masks: Optional[Tensor]) -> Tensor:
    """dirs [...,3], coeffs [...,K,3], masks [...] bool -> colors [...,3].
    Masked-out rows are 0 (the reference leaves them uninitialised, SphericalHarmonics.cpp:30)."""
    masks = _opt(masks)
    require_gpu(dirs, coeffs, masks)
    if coeffs.shape[-1] != 3:
        raise LfsError("coeffs must have last dimension 3")
    if dirs.shape[-1] != 3:
        raise LfsError("dirs must have last dimension 3")
    colors = torch.empty_like(dirs)
    n, K = dirs.numel() // 3, coeffs.shape[-2]
    rc = load_library().lfs_spherical_harmonics_fwd(
        C.c_uint32(n), C.c_uint32(K), C.c_uint32(degrees_to_use), ptr(dirs), ptr(coeffs), ptr(masks), ptr(colors), stream())
    check(rc, "spherical_harmonics_fwd")
    return colors


def spherical_harmonics_bwd(K: int, degrees_to_use: int, dirs: Tensor, coeffs: Tensor, masks: Optional[Tensor],
                            v_colors: Tensor, compute_v_dirs: bool) -> Tuple[Tensor, Optional[Tensor]]:
    masks = _opt(masks)
    require_gpu(dirs, coeffs, v_colors, masks)
    if v_colors.shape[-1] != 3 or coeffs.shape[-1] != 3 or dirs.shape[-1] != 3:
        raise LfsError("v_colors / coeffs / dirs must have last dimension 3")
    if coeffs.shape[-2] != K:
        raise LfsError("K does not match coeffs")
    v_coeffs = torch.empty_like(coeffs)  # fully written by the kernel (reference: zeros_like + partial write)
    v_dirs = torch.empty_like(dirs) if compute_v_dirs else None
    n = dirs.numel() // 3
    rc = load_library().lfs_spherical_harmonics_bwd(
        C.c_uint32(n), C.c_uint32(K), C.c_uint32(degrees_to_use), ptr(dirs), ptr(coeffs), ptr(masks), ptr(v_colors),
        ptr(v_coeffs), ptr(v_dirs), stream())
    check(rc, "spherical_harmonics_bwd")
    return v_coeffs, v_dirs


# -----------------------------------------------------------------------------------------
# Tile intersection (Ops.h:27-43, Intersect.cpp:15-137)
# -----------------------------------------------------------------------------------------
def intersect_tile(means2d: Tensor, radii: Tensor, depths: Tensor, camera_ids: Optional[Tensor], gaussian_ids: Optional[Tensor],
                   C_: int, tile_size: int, tile_width: int, tile_height: int, sort: bool,
                   *, return_offsets: bool = False, overlap=None):
    """means2d [C,N,2], radii int32 [C,N,2], depths [C,N] ->
    (tiles_per_gauss int32 [C,N], isect_ids int64 [n_isects], flatten_ids int32 [n_isects]).
    One host sync for n_isects, at the same place as the reference (Intersect.cpp:76).
    `return_offsets=True` (extension) appends the [C,tile_h,tile_w] offsets the sorted path gets for free.
    `overlap` (extension): a callable that enqueues independent GPU work (the SH colours in the fused step); it runs after the asynchronous
    read-back of n_isects has been queued, and the host then waits for the read-back EVENT only, so the GPU executes that work instead of
    idling through the host round trip (~40 us per step at 1M Gaussians). Its return value is appended to the result."""
    st = intersect_tile_begin(means2d, radii, depths, C_, tile_size, tile_width, tile_height, sort, return_offsets=return_offsets, pinned=overlap is not None)
    extra = overlap() if overlap is not None else None
    out = intersect_tile_finish(st)
    return out + (extra,) if overlap is not None else out


def intersect_tile_begin(means2d: Tensor, radii: Tensor, depths: Tensor, C_: int, tile_size: int, tile_width: int, tile_height: int, sort: bool,
                         *, return_offsets: bool = False, pinned: bool = True, slot: int = 0) -> dict:
    """First half of intersect_tile (extension): enqueue the count + scan kernels and return the state intersect_tile_finish() needs. A multi-view
    step begins the intersections of ALL its views first (slot = view index: each gets its own workspace and its own pinned counter pair) and
    finishes them one by one - the host then waits once per step instead of once per view (fused.render_views_and_backward)."""
    require_gpu(means2d, radii, depths)
    if means2d.dim() == 2:
        raise LfsError("packed mode is not supported (the reference's trainer never uses it: rasterizer.cpp:56)")
    if radii.dtype != torch.int32:
        raise LfsError("radii must be int32")
    lib = load_library()
    N = means2d.shape[1]
    dev = means2d.device
    tiles_per_gauss = torch.empty(depths.shape, dtype=torch.int32, device=dev)
    ws_bytes = lib.lfs_intersect_tile_workspace_bytes(C.c_uint32(C_), C.c_uint32(N), C.c_uint32(tile_width), C.c_uint32(tile_height))
    tag = "isect" if slot == 0 else f"isect{slot}"
    ws = workspace(ws_bytes, dev, tag)
    offsets = torch.empty((C_, tile_height, tile_width), dtype=torch.int32, device=dev) if (return_offsets and sort) else None
    # flags = 0: the library zeroes the workspace counters itself on every call (a 4-byte-per-tile memset). The LFS_ISECT_COUNTERS_ZERO elision is for
    # callers that OWN their workspace memory exclusively (csrc/gut_step.cpp's step context); a Python-side cache of "this buffer is still clean" was
    # removed in round 3: the ops-level path must not depend on interpreter state to stay inside its buffers.
    flags = 0
    # pinned: the scan kernel writes the counts straight into pinned host memory (no copy kernel) and the host waits for an event recorded behind
    # that kernel; otherwise they land in device memory and .tolist() reads them
    host = _pinned_i64(slot) if pinned else None   # [n_isects, longest tile list, stamp of the call that wrote them]
    stamp = 0
    if pinned:
        _STAMP[0] += 1
        stamp = _STAMP[0]
    n_dev = None if pinned else torch.empty(2, dtype=torch.int64, device=dev)
    counts = host if host is not None else n_dev
    rc = lib.lfs_intersect_tile_count_ex(
        C.c_uint32(C_), C.c_uint32(N), ptr(means2d), ptr(radii), C.c_uint32(tile_size), C.c_uint32(tile_width), C.c_uint32(tile_height),
        ptr(tiles_per_gauss), C.c_void_p(counts.data_ptr()), C.c_void_p(counts.data_ptr() + 8), ptr(offsets), C.c_uint32(flags),
        C.c_void_p(host.data_ptr() + 16) if pinned else None, C.c_int64(stamp), ptr(ws), C.c_size_t(ws.numel()), stream())
    check(rc, "intersect_tile (count)")
    ev = None
    if pinned:
        ev = torch.cuda.Event()
        ev.record()
    return dict(means2d=means2d, radii=radii, depths=depths, C=C_, N=N, tile_size=tile_size, tw=tile_width, th=tile_height, sort=sort, tpg=tiles_per_gauss,
                ws=ws, offsets=offsets, return_offsets=return_offsets, host=host, n_dev=n_dev, ev=ev, stamp=stamp)


def intersect_tile_finish(st: dict):
    """Second half of intersect_tile: wait for the counts (the one host sync of the path), allocate the outputs, enqueue scatter + sort."""
    lib, dev = load_library(), st["means2d"].device
    if st["ev"] is not None:
        st["ev"].synchronize()
        host, t0 = st["host"], None
        while int(host[2]) != st["stamp"]:     # the kernel writes its stamp behind a system-scope fence after the counts: normally there already
            if t0 is None:
                t0 = time.monotonic()
            elif time.monotonic() - t0 > 5.0:
                raise LfsError("intersect_tile: the counts of this call never arrived in pinned host memory")
        n_isects, longest = int(host[0]), int(host[1])
    else:
        n_isects, longest = (int(x) for x in st["n_dev"].tolist())  # the one D2H sync of the path
    sort = st["sort"]
    isect_ids = torch.empty(n_isects, dtype=torch.int64, device=dev)
    flatten_ids = torch.empty(n_isects, dtype=torch.int32, device=dev)
    binned = torch.empty(n_isects, dtype=torch.int64, device=dev) if (sort and n_isects) else None   # two-pass scatter: the row-binned intermediate
    rc = lib.lfs_intersect_tile_emit_ex(
        C.c_uint32(st["C"]), C.c_uint32(st["N"]), ptr(st["means2d"]), ptr(st["radii"]), ptr(st["depths"]), C.c_uint32(st["tile_size"]), C.c_uint32(st["tw"]),
        C.c_uint32(st["th"]), C.c_int(int(bool(sort))), C.c_int64(n_isects), ptr(st["tpg"]), ptr(isect_ids), ptr(flatten_ids),
        None, ptr(binned), C.c_int64(longest), ptr(st["ws"]), C.c_size_t(st["ws"].numel()), stream())
    check(rc, "intersect_tile (emit)")
    return (st["tpg"], isect_ids, flatten_ids) + ((st["offsets"],) if st["return_offsets"] else ())


_PINNED = {}
_STAMP = [0]


def _pinned_i64(slot: int = 0) -> Tensor:
    t = _PINNED.get(("i64", slot))
    if t is None:
        t = torch.zeros(3, dtype=torch.int64).pin_memory()
        _PINNED[("i64", slot)] = t
    return t


def intersect_offset(isect_ids: Tensor, C_: int, tile_width: int, tile_height: int) -> Tensor:
    require_gpu(isect_ids)
    offsets = torch.empty((C_, tile_height, tile_width), dtype=torch.int32, device=isect_ids.device)
    rc = load_library().lfs_intersect_offset(C.c_int64(isect_ids.shape[0]), ptr(isect_ids), C.c_uint32(C_), C.c_uint32(tile_width),
                                             C.c_uint32(tile_height), ptr(offsets), stream())
    check(rc, "intersect_offset")
    return offsets


# -----------------------------------------------------------------------------------------
# MCMC helpers (Ops.h:45-65)
# -----------------------------------------------------------------------------------------
def quats_to_rotmats(quats: Tensor) -> Tensor:
    require_gpu(quats)
    N = quats.shape[0]
    rotmats = torch.empty((N, 3, 3), dtype=quats.dtype, device=quats.device)
    check(load_library().lfs_quats_to_rotmats(C.c_uint32(N), ptr(quats), ptr(rotmats), stream()), "quats_to_rotmats")
    return rotmats


def relocation(opacities: Tensor, scales: Tensor, ratios: Tensor, binoms: Tensor, n_max: int) -> Tuple[Tensor, Tensor]:
    require_gpu(opacities, scales, ratios, binoms)
    if ratios.dtype != torch.int32:
        raise LfsError("ratios must be int32")
    new_opacities, new_scales = torch.empty_like(opacities), torch.empty_like(scales)
    check(load_library().lfs_relocation(C.c_uint32(opacities.shape[0]), ptr(opacities), ptr(scales), ptr(ratios), ptr(binoms),
                                        C.c_int32(n_max), ptr(new_opacities), ptr(new_scales), stream()), "relocation")
    return new_opacities, new_scales


def add_noise(raw_opacities: Tensor, raw_scales: Tensor, raw_quats: Tensor, noise: Tensor, means: Tensor, current_lr: float) -> None:
    require_gpu(raw_opacities, raw_scales, raw_quats, noise, means)
    check(load_library().lfs_add_noise(C.c_uint32(raw_opacities.shape[0]), ptr(raw_opacities), ptr(raw_scales), ptr(raw_quats),
                                       ptr(noise), ptr(means), C.c_float(current_lr), stream()), "add_noise")


# -----------------------------------------------------------------------------------------
# 3DGUT projection (Ops.h:66-90, Projection.cpp:22-110)
# -----------------------------------------------------------------------------------------
def projection_ut_3dgs_fused(means: Tensor, quats: Tensor, scales: Tensor, opacities: Optional[Tensor],
                             viewmats0: Tensor, viewmats1: Optional[Tensor], Ks: Tensor,
                             image_width: int, image_height: int, eps2d: float, near_plane: float, far_plane: float,
                             radius_clip: float, calc_compensations: bool, camera_model: CameraModelType,
                             ut_params: Optional[UnscentedTransformParameters] = None,
                             rs_type: ShutterType = ShutterType.GLOBAL,
                             radial_coeffs: Optional[Tensor] = None, tangential_coeffs: Optional[Tensor] = None,
                             thin_prism_coeffs: Optional[Tensor] = None):
    """-> (radii int32 [C,N,2], means2d [C,N,2], depths [C,N], conics [C,N,3], compensations [C,N] | None).
    Outputs of culled Gaussians are 0 (uninitialised in the reference)."""
    opacities, viewmats1 = _opt(opacities), _opt(viewmats1)
    radial_coeffs, tangential_coeffs, thin_prism_coeffs = _opt(radial_coeffs), _opt(tangential_coeffs), _opt(thin_prism_coeffs)
    require_gpu(means, quats, scales, opacities, viewmats0, viewmats1, Ks, radial_coeffs, tangential_coeffs, thin_prism_coeffs)
    N, Cn = means.shape[0], Ks.shape[0]
    dev = means.device
    radii = torch.empty((Cn, N, 2), dtype=torch.int32, device=dev)
    means2d = torch.empty((Cn, N, 2), dtype=means.dtype, device=dev)
    depths = torch.empty((Cn, N), dtype=means.dtype, device=dev)
    conics = torch.empty((Cn, N, 3), dtype=means.dtype, device=dev)
    compensations = torch.zeros((Cn, N), dtype=means.dtype, device=dev) if calc_compensations else None
    cams = cameras_struct(viewmats0, viewmats1, Ks, image_width, image_height, camera_model, rs_type,
                          radial_coeffs, tangential_coeffs, thin_prism_coeffs)
    ut = ut_struct(ut_params)
    rc = load_library().lfs_projection_ut_3dgs_fused(
        C.c_uint32(N), ptr(means), ptr(quats), ptr(scales), ptr(opacities), C.byref(cams),
        C.c_float(eps2d), C.c_float(near_plane), C.c_float(far_plane), C.c_float(radius_clip), C.byref(ut),
        ptr(radii), ptr(means2d), ptr(depths), ptr(conics), ptr(compensations), stream())
    check(rc, "projection_ut_3dgs_fused")
    return radii, means2d, depths, conics, compensations


# -----------------------------------------------------------------------------------------
# World-space rasterization (Ops.h:92-166, Rasterization.cpp:20-261)
# -----------------------------------------------------------------------------------------
def _raster_ws(C_: int, N: int, channels: int, image_width: int, image_height: int, tile_size: int, n_isects: int, dev) -> Tensor:
    nbytes = load_library().lfs_rasterize_workspace_bytes(C.c_uint32(C_), C.c_uint32(N), C.c_uint32(channels), C.c_uint32(image_width),
                                                          C.c_uint32(image_height), C.c_uint32(tile_size), C.c_int64(n_isects))
    if nbytes == 0:
        raise LfsError(f"rasterize: unsupported tile_size {tile_size}")
    return workspace(nbytes, dev, "raster")


def rasterize_to_pixels_from_world_3dgs_fwd(
        means: Tensor, quats: Tensor, scales: Tensor, colors: Tensor, opacities: Tensor,
        backgrounds: Optional[Tensor], masks: Optional[Tensor],
        image_width: int, image_height: int, tile_size: int,
        viewmats0: Tensor, viewmats1: Optional[Tensor], Ks: Tensor, camera_model: CameraModelType,
        ut_params: Optional[UnscentedTransformParameters], rs_type: ShutterType,
        radial_coeffs: Optional[Tensor], tangential_coeffs: Optional[Tensor], thin_prism_coeffs: Optional[Tensor],
        tile_offsets: Tensor, flatten_ids: Tensor, own_workspace: bool = False):
    """-> (renders [C,H,W,channels], alphas [C,H,W,1], last_ids int32 [C,H,W]). channels in 1..4
    (the reference asserts 3, Rasterization.cpp:65, although its L2 builds 1- and 4-channel inputs).
    own_workspace=True (extension): the call gets a fresh workspace tensor, returned as a 4th value, that the
    matching backward can reuse (prepared_workspace=) instead of rebuilding records and per-cell lists."""
    backgrounds, masks, viewmats1 = _opt(backgrounds), _opt(masks), _opt(viewmats1)
    radial_coeffs, tangential_coeffs, thin_prism_coeffs = _opt(radial_coeffs), _opt(tangential_coeffs), _opt(thin_prism_coeffs)
    require_gpu(means, quats, scales, colors, opacities, backgrounds, masks, viewmats0, viewmats1, Ks,
                radial_coeffs, tangential_coeffs, thin_prism_coeffs, tile_offsets, flatten_ids)
    if opacities.dim() == 1:
        raise LfsError("packed mode is not supported")
    Cn, N, channels = tile_offsets.shape[0], means.shape[0], colors.shape[-1]
    dev = means.device
    renders = torch.empty((Cn, image_height, image_width, channels), dtype=means.dtype, device=dev)
    alphas = torch.empty((Cn, image_height, image_width, 1), dtype=means.dtype, device=dev)
    last_ids = torch.empty((Cn, image_height, image_width), dtype=torch.int32, device=dev)
    cams = cameras_struct(viewmats0, viewmats1, Ks, image_width, image_height, camera_model, rs_type,
                          radial_coeffs, tangential_coeffs, thin_prism_coeffs)
    ut = ut_struct(ut_params)
    if own_workspace:
        nbytes = load_library().lfs_rasterize_workspace_bytes(C.c_uint32(Cn), C.c_uint32(N), C.c_uint32(channels), C.c_uint32(image_width),
                                                              C.c_uint32(image_height), C.c_uint32(tile_size), C.c_int64(flatten_ids.shape[0]))
        if nbytes == 0:
            raise LfsError(f"rasterize: unsupported tile_size {tile_size}")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    else:
        ws = _raster_ws(Cn, N, channels, image_width, image_height, tile_size, flatten_ids.shape[0], dev)
    rc = load_library().lfs_rasterize_to_pixels_from_world_3dgs_fwd(
        C.c_uint32(N), C.c_uint32(channels), ptr(means), ptr(quats), ptr(scales), ptr(colors), ptr(opacities),
        ptr(backgrounds), ptr(masks), C.byref(cams), C.c_uint32(tile_size), C.byref(ut),
        ptr(tile_offsets), ptr(flatten_ids), C.c_int64(flatten_ids.shape[0]),
        ptr(renders), ptr(alphas), ptr(last_ids), ptr(ws), C.c_size_t(ws.numel()), stream())
    if rc == -2:
        raise LfsError(f"Unsupported number of channels / tile size / camera model: channels={channels}, tile_size={tile_size}")
    check(rc, "rasterize_to_pixels_from_world_3dgs_fwd")
    if own_workspace:
        return renders, alphas, last_ids, ws
    return renders, alphas, last_ids


def _out_like(out: Optional[Tensor], like: Tensor) -> Tensor:
    if out is None:
        return torch.empty_like(like)
    if tuple(out.shape) != tuple(like.shape) or out.dtype != like.dtype or not out.is_contiguous() or out.device != like.device:
        raise LfsError("output buffer: shape / dtype / contiguity mismatch")
    return out


def rasterize_to_pixels_from_world_3dgs_bwd(
        means: Tensor, quats: Tensor, scales: Tensor, colors: Tensor, opacities: Tensor,
        backgrounds: Optional[Tensor], masks: Optional[Tensor],
        image_width: int, image_height: int, tile_size: int,
        viewmats0: Tensor, viewmats1: Optional[Tensor], Ks: Tensor, camera_model: CameraModelType,
        ut_params: Optional[UnscentedTransformParameters], rs_type: ShutterType,
        radial_coeffs: Optional[Tensor], tangential_coeffs: Optional[Tensor], thin_prism_coeffs: Optional[Tensor],
        tile_offsets: Tensor, flatten_ids: Tensor, render_alphas: Tensor, last_ids: Tensor,
        v_render_colors: Tensor, v_render_alphas: Tensor, prepared_workspace: Optional[Tensor] = None,
        v_colors_out: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """-> (v_means [N,3], v_quats [N,4], v_scales [N,3], v_colors [C,N,ch], v_opacities [C,N]). v_colors_out (extension): a contiguous tensor of
    v_colors' shape to write into (a slice of a caller's multi-view buffer) instead of a fresh one."""
    backgrounds, masks, viewmats1 = _opt(backgrounds), _opt(masks), _opt(viewmats1)
    radial_coeffs, tangential_coeffs, thin_prism_coeffs = _opt(radial_coeffs), _opt(tangential_coeffs), _opt(thin_prism_coeffs)
    require_gpu(means, quats, scales, colors, opacities, backgrounds, masks, viewmats0, viewmats1, Ks,
                radial_coeffs, tangential_coeffs, thin_prism_coeffs, tile_offsets, flatten_ids,
                render_alphas, last_ids, v_render_colors, v_render_alphas)
    Cn, N, channels = tile_offsets.shape[0], means.shape[0], colors.shape[-1]
    dev = means.device
    v_means, v_quats, v_scales = torch.empty_like(means), torch.empty_like(quats), torch.empty_like(scales)
    v_colors, v_opacities = _out_like(v_colors_out, colors), torch.empty_like(opacities)
    cams = cameras_struct(viewmats0, viewmats1, Ks, image_width, image_height, camera_model, rs_type,
                          radial_coeffs, tangential_coeffs, thin_prism_coeffs)
    ut = ut_struct(ut_params)
    lib = load_library()
    if prepared_workspace is not None:
        ws, entry = prepared_workspace, lib.lfs_rasterize_to_pixels_from_world_3dgs_bwd_prepared
    else:
        ws, entry = _raster_ws(Cn, N, channels, image_width, image_height, tile_size, flatten_ids.shape[0], dev), lib.lfs_rasterize_to_pixels_from_world_3dgs_bwd
    rc = entry(
        C.c_uint32(N), C.c_uint32(channels), ptr(means), ptr(quats), ptr(scales), ptr(colors), ptr(opacities),
        ptr(backgrounds), ptr(masks), C.byref(cams), C.c_uint32(tile_size), C.byref(ut),
        ptr(tile_offsets), ptr(flatten_ids), C.c_int64(flatten_ids.shape[0]),
        ptr(render_alphas), ptr(last_ids), ptr(v_render_colors), ptr(v_render_alphas),
        ptr(v_means), ptr(v_quats), ptr(v_scales), ptr(v_colors), ptr(v_opacities),
        ptr(ws), C.c_size_t(ws.numel()), stream())
    check(rc, "rasterize_to_pixels_from_world_3dgs_bwd")
    return v_means, v_quats, v_scales, v_colors, v_opacities


def rasterize_bwd_prepared_mse(means: Tensor, quats: Tensor, scales: Tensor, colors: Tensor, opacities: Tensor, backgrounds: Optional[Tensor],
                               image_width: int, image_height: int, tile_size: int, viewmats0: Tensor, Ks: Tensor, camera_model: CameraModelType,
                               rs_type: ShutterType, tile_offsets: Tensor, flatten_ids: Tensor, render_colors: Tensor, render_alphas: Tensor, last_ids: Tensor,
                               target_chw: Tensor, weight: float, loss_acc: Tensor, prepared_workspace: Tensor, v_colors_out: Optional[Tensor] = None):
    """Extension: the prepared backward with the clamped MSE loss folded in (lfs_..._bwd_prepared_mse): loss_acc += weight * mse(clamp(render), target),
    dL/d(render) stays in registers. -> (v_means, v_quats, v_scales, v_colors [1,N,3], v_opacities [1,N])."""
    backgrounds = _opt(backgrounds)
    target_chw = target_chw.contiguous()
    require_gpu(means, quats, scales, colors, opacities, backgrounds, viewmats0, Ks, tile_offsets, flatten_ids, render_colors, render_alphas, last_ids, target_chw, loss_acc)
    N = means.shape[0]
    assert colors.shape[-1] == 3 and tile_offsets.shape[0] == 1 and tuple(target_chw.shape) == (3, image_height, image_width)
    v_means, v_quats, v_scales = torch.empty_like(means), torch.empty_like(quats), torch.empty_like(scales)
    v_colors, v_opacities = _out_like(v_colors_out, colors), torch.empty_like(opacities)
    cams = cameras_struct(viewmats0, None, Ks, image_width, image_height, camera_model, rs_type, None, None, None)
    rc = load_library().lfs_rasterize_to_pixels_from_world_3dgs_bwd_prepared_mse(
        C.c_uint32(N), ptr(means), ptr(quats), ptr(scales), ptr(colors), ptr(opacities), ptr(backgrounds), C.byref(cams), C.c_uint32(tile_size),
        ptr(tile_offsets), ptr(flatten_ids), C.c_int64(flatten_ids.shape[0]), ptr(render_colors), ptr(render_alphas), ptr(last_ids), ptr(target_chw),
        C.c_float(weight), ptr(loss_acc), ptr(v_means), ptr(v_quats), ptr(v_scales), ptr(v_colors), ptr(v_opacities),
        ptr(prepared_workspace), C.c_size_t(prepared_workspace.numel()), stream())
    check(rc, "rasterize_bwd_prepared_mse")
    return v_means, v_quats, v_scales, v_colors, v_opacities


# -----------------------------------------------------------------------------------------
# fast_gs::optimizer (adam_api.h:11-21, adam.h:9-20)
# -----------------------------------------------------------------------------------------
def adam_step_wrapper(param: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, param_grad: Tensor, lr: float, beta1: float,
                      beta2: float, eps: float, bias_correction1_rcp: float, bias_correction2_sqrt_rcp: float) -> None:
    """In place on param / exp_avg / exp_avg_sq. Launches on the current stream (the reference uses
    the legacy default stream, adam.cu:22)."""
    require_gpu(param, exp_avg, exp_avg_sq, param_grad)
    check(load_library().lfs_adam_step(ptr(param), ptr(exp_avg), ptr(exp_avg_sq), ptr(param_grad), C.c_int64(param.numel()),
                                       C.c_float(lr), C.c_float(beta1), C.c_float(beta2), C.c_float(eps),
                                       C.c_float(bias_correction1_rcp), C.c_float(bias_correction2_sqrt_rcp), stream()), "adam_step")


def adam_step_multi(entries) -> None:
    """entries: iterable of (param, exp_avg, exp_avg_sq, grad, lr, beta1, beta2, eps, bc1_rcp, bc2_sqrt_rcp);
    one launch for up to 8 tensors, same arithmetic as adam_step_wrapper per tensor."""
    entries = list(entries)
    if not entries:
        return
    if len(entries) > capi.ADAM_MAX_TENSORS:
        raise LfsError("too many tensors for one adam_step_multi launch")
    arr = (capi.AdamTensor * len(entries))()
    for i, (p, m, v, g, lr, b1, b2, eps, bc1, bc2) in enumerate(entries):
        require_gpu(p, m, v, g)
        arr[i] = capi.AdamTensor(p.data_ptr(), m.data_ptr(), v.data_ptr(), g.data_ptr(), p.numel(), lr, b1, b2, eps, bc1, bc2)
    check(load_library().lfs_adam_step_multi(arr, C.c_int32(len(entries)), stream()), "adam_step_multi")
