"""Fused L2 path for the --gut training step (extension; same arithmetic as rasterizer.py + torch autograd).

rasterizer.py mirrors the reference's L2 one libtorch op at a time (activations, `cat(sh0, shN)`, `dirs`, masks,
`clamp_min`, permute / clamp of the image, the loss and the autograd graph of all of it): ~100 small launches
and ~0.8 ms of HBM passes per step at 1M Gaussians. Here the same step is a fixed sequence of C-ABI calls
with NO autograd graph: forward and backward are issued explicitly, the element-wise glue lives in the fused
kernels of csrc/l2_fused.hip and csrc/sh.hip (`lfs_sh_model_*`), the rasterizer backward reuses the
forward's workspace, and the raw-parameter gradients are written straight into the caller's buffers (for
data parallel runs: views of the flat all-reduce bucket, so there is no gather copy).

Restricted to what the trainer uses: one camera per call, RGB render mode, global shutter, MSE loss.
tests/test_gpu_fused.py checks every stage and the end-to-end gradients against the autograd path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional

import torch

from . import ops
from .capi import (CameraModelType, ShutterType, UnscentedTransformParameters, cameras_struct, check, load_library, ptr,
                   require_gpu, stream, ut_struct)
from .rasterizer import Camera, SplatModel


def activations_fwd(raw_quats, raw_scales, raw_opacities):
    require_gpu(raw_quats, raw_scales, raw_opacities)
    N = raw_quats.shape[0]
    quats, scales, opac = torch.empty_like(raw_quats), torch.empty_like(raw_scales), torch.empty_like(raw_opacities)
    check(load_library().lfs_activations_fwd(C.c_uint32(N), ptr(raw_quats), ptr(raw_scales), ptr(raw_opacities),
                                             ptr(quats), ptr(scales), ptr(opac), stream()), "activations_fwd")
    return quats, scales, opac


def activations_bwd(raw_quats, scales, opacities, v_quats, v_scales, v_opacities, g_raw_quats, g_raw_scales, g_raw_opacities, accumulate: bool,
                    scale_reg: float = 0.0, opacity_reg: float = 0.0):
    require_gpu(raw_quats, scales, opacities, v_quats, v_scales, v_opacities, g_raw_quats, g_raw_scales, g_raw_opacities)
    check(load_library().lfs_activations_bwd(C.c_uint32(raw_quats.shape[0]), ptr(raw_quats), ptr(scales), ptr(opacities),
                                             ptr(v_quats), ptr(v_scales), ptr(v_opacities), C.c_float(scale_reg), C.c_float(opacity_reg), C.c_int(int(accumulate)),
                                             ptr(g_raw_quats), ptr(g_raw_scales), ptr(g_raw_opacities), stream()), "activations_bwd")


def sh_model_fwd(sh_degree: int, means, viewmat, sh0, shN, radii):
    require_gpu(means, viewmat, sh0, shN, radii)
    N, K = means.shape[0], 1 + shN.shape[1]
    colors = torch.empty((N, 3), dtype=means.dtype, device=means.device)
    check(load_library().lfs_sh_model_fwd(C.c_uint32(N), C.c_uint32(K), C.c_uint32(sh_degree), ptr(means), ptr(viewmat), ptr(sh0), ptr(shN),
                                          ptr(radii), ptr(colors), stream()), "sh_model_fwd")
    return colors


def sh_model_bwd(sh_degree: int, means, viewmat, sh0, shN, radii, colors, v_colors, v_sh0, v_shN, v_means, accumulate: bool):
    require_gpu(means, viewmat, sh0, shN, radii, colors, v_colors, v_sh0, v_shN, v_means)
    N, K = means.shape[0], 1 + shN.shape[1]
    check(load_library().lfs_sh_model_bwd(C.c_uint32(N), C.c_uint32(K), C.c_uint32(sh_degree), ptr(means), ptr(viewmat), ptr(sh0), ptr(shN),
                                          ptr(radii), ptr(colors), ptr(v_colors), C.c_int(int(accumulate)),
                                          ptr(v_sh0), ptr(v_shN), ptr(v_means), stream()), "sh_model_bwd")


def sh_model_bwd_adam(sh_degree: int, means, viewmat, sh0, shN, radii, colors, v_colors, v_sh0, v_means, adam: dict):
    """sh_model_bwd of a single-view step with the Adam update of shN applied in the same kernel (`adam`: FusedAdam.prepare_inline)."""
    require_gpu(means, viewmat, sh0, shN, radii, colors, v_colors, v_sh0, v_means, adam["exp_avg"], adam["exp_avg_sq"])
    N, K = means.shape[0], 1 + shN.shape[1]
    check(load_library().lfs_sh_model_bwd_adam(C.c_uint32(N), C.c_uint32(K), C.c_uint32(sh_degree), ptr(means), ptr(viewmat), ptr(sh0), ptr(shN),
                                               ptr(radii), ptr(colors), ptr(v_colors), ptr(v_sh0), ptr(v_means), ptr(adam["exp_avg"]), ptr(adam["exp_avg_sq"]),
                                               C.c_float(adam["lr"]), C.c_float(adam["beta1"]), C.c_float(adam["beta2"]), C.c_float(adam["eps"]),
                                               C.c_float(adam["bc1_rcp"]), C.c_float(adam["bc2_sqrt_rcp"]), stream()), "sh_model_bwd_adam")


# ---- the front half of the fused step: activations + projection in one kernel -------------------------------------------------------------------
def activations_project(means, raw_quats, raw_scales, raw_opacities, viewmat, Kmat, W: int, H: int, ut, camera_model=CameraModelType.PINHOLE, radii_out=None):
    """activations_fwd + ops.projection_ut_3dgs_fused (trainer constants of rasterizer.cpp:176-181) in one kernel -> (quats, scales, opacities, radii, means2d, depths)"""
    require_gpu(means, raw_quats, raw_scales, raw_opacities, viewmat, Kmat)
    N, dev = means.shape[0], means.device
    quats, scales, opac = torch.empty_like(raw_quats), torch.empty_like(raw_scales), torch.empty_like(raw_opacities)
    radii = radii_out if radii_out is not None else torch.empty((1, N, 2), dtype=torch.int32, device=dev)   # (radii_out: a [1,N,2] slice of a multi-view buffer)
    assert tuple(radii.shape) == (1, N, 2) and radii.dtype == torch.int32 and radii.is_contiguous()
    means2d, depths = torch.empty((1, N, 2), dtype=means.dtype, device=dev), torch.empty((1, N), dtype=means.dtype, device=dev)
    cams = cameras_struct(viewmat, None, Kmat, W, H, camera_model, ShutterType.GLOBAL, None, None, None)
    from .capi import ut_struct
    u = ut_struct(ut)
    check(load_library().lfs_activations_project_ut(C.c_uint32(N), ptr(means), ptr(raw_quats), ptr(raw_scales), ptr(raw_opacities), C.byref(cams), C.c_float(0.3),
                                                    C.c_float(0.01), C.c_float(10000.0), C.c_float(0.0), C.byref(u), ptr(quats), ptr(scales), ptr(opac), ptr(radii),
                                                    ptr(means2d), ptr(depths), stream()), "activations_project_ut")
    return quats, scales, opac, radii, means2d, depths


FUSE_ACT_PROJ = True   # activations + projection in one kernel (bit-identical outputs; saves one launch and the re-read of the activated values)

def _adam_scalars(a: dict):
    return (C.c_float * 6)(a["lr"], a["beta1"], a["beta2"], a["eps"], a["bc1_rcp"], a["bc2_sqrt_rcp"])


def backward_adam_all(sh_degree: int, means, raw_quats, raw_scales, raw_opac, sh0, shN, quats, scales, opac, colors, radii, bg, W: int, H: int, tile: int,
                      viewmat, Kmat, offsets, flatten_ids, render, alpha, last_ids, target_chw, weight: float, loss_acc, ws, adam: dict,
                      scale_reg: float = 0.0, opacity_reg: float = 0.0) -> None:
    """The all-inline backward of a one-view step on one rank (the reference's training configuration): rasterizer backward with the MSE folded in
    -> SH backward with Adam on sh0 / shN -> finish + activation backward + Adam on means / scales / quaternions / opacities. Three launches after the
    rasterizer kernel, no gradient tensor; `adam[name]` = FusedAdam.prepare_inline(param) for the six parameters. Element for element the operations
    of the separate kernels (tests/test_gpu_fused.py)."""
    from .capi import cameras_struct
    lib = load_library()
    N, Kc = means.shape[0], 1 + shN.shape[1]
    cams = cameras_struct(viewmat, None, Kmat, W, H, CameraModelType.PINHOLE, ShutterType.GLOBAL, None, None, None)
    check(lib.lfs_rasterize_to_pixels_from_world_3dgs_bwd_prepared_mse_acc(
        C.c_uint32(N), ptr(means), ptr(quats), ptr(scales), ptr(colors), ptr(opac), ptr(bg), C.byref(cams), C.c_uint32(tile), ptr(offsets), ptr(flatten_ids),
        C.c_int64(flatten_ids.shape[0]), ptr(render), ptr(alpha), ptr(last_ids), ptr(target_chw), C.c_float(weight), ptr(ws), C.c_size_t(ws.numel()), stream()),
        "rasterize_bwd_prepared_mse_acc")
    acc_rows = C.c_void_p(ws.data_ptr() + lib.lfs_rasterize_workspace_acc_offset(C.c_uint32(1), C.c_uint32(N)))
    a0, aN = adam["sh0"], adam["shN"]
    v_dirs = torch.empty_like(means)
    check(lib.lfs_sh_model_bwd_adam_all(C.c_uint32(N), C.c_uint32(Kc), C.c_uint32(sh_degree), ptr(means), ptr(viewmat), ptr(sh0), ptr(shN), ptr(radii), ptr(colors),
                                        acc_rows, ptr(v_dirs), ptr(a0["exp_avg"]), ptr(a0["exp_avg_sq"]), _adam_scalars(a0), ptr(aN["exp_avg"]), ptr(aN["exp_avg_sq"]),
                                        _adam_scalars(aN), stream()), "sh_model_bwd_adam_all")
    order = ["means", "raw_scales", "raw_quats", "raw_opacities"]
    m = (C.c_void_p * 4)(*[adam[k]["exp_avg"].data_ptr() for k in order])
    v = (C.c_void_p * 4)(*[adam[k]["exp_avg_sq"].data_ptr() for k in order])
    sc = (C.c_float * 24)(*[x for k in order for x in (adam[k]["lr"], adam[k]["beta1"], adam[k]["beta2"], adam[k]["eps"], adam[k]["bc1_rcp"], adam[k]["bc2_sqrt_rcp"])])
    check(lib.lfs_gut_finish_adam(C.c_uint32(N), ptr(means), ptr(raw_scales), ptr(raw_quats), ptr(raw_opac), ptr(quats), ptr(scales), ptr(opac), ptr(v_dirs), m, v, sc,
                                  C.c_float(scale_reg), C.c_float(opacity_reg), ptr(loss_acc), ptr(ws), C.c_size_t(ws.numel()), stream()), "gut_finish_adam")


def backward_grads(means, raw_quats, quats, scales, opac, colors, bg, W: int, H: int, tile: int, viewmat, Kmat, offsets, flatten_ids, render, alpha, last_ids,
                   ws, g_means, g_scales, g_quats, g_opac, accumulate: bool, scale_reg: float, opacity_reg: float, *, target_chw=None, weight: float = 0.0,
                   loss_acc=None, v_render=None, v_colors_out=None):
    """Rasterizer backward -> gradient tensors of the raw parameters in two launches (the backward kernel, then lfs_gut_finish_grads = raster_finish +
    activations_bwd + the copy of dL/dmeans in one pass). With target_chw the clamped MSE is folded into the backward (loss_acc += it), otherwise
    v_render [1,H,W,3] is the caller's dL/d(render). g_means receives the rasterizer's part of dL/dmeans (the SH backward adds dL/d(dirs) onto it);
    returns dL/dcolour [1,N,3]."""
    from .capi import cameras_struct
    lib = load_library()
    N = means.shape[0]
    cams = cameras_struct(viewmat, None, Kmat, W, H, CameraModelType.PINHOLE, ShutterType.GLOBAL, None, None, None)
    if target_chw is not None:
        check(lib.lfs_rasterize_to_pixels_from_world_3dgs_bwd_prepared_mse_acc(
            C.c_uint32(N), ptr(means), ptr(quats), ptr(scales), ptr(colors), ptr(opac), ptr(bg), C.byref(cams), C.c_uint32(tile), ptr(offsets), ptr(flatten_ids),
            C.c_int64(flatten_ids.shape[0]), ptr(render), ptr(alpha), ptr(last_ids), ptr(target_chw), C.c_float(weight), ptr(ws), C.c_size_t(ws.numel()), stream()),
            "rasterize_bwd_prepared_mse_acc")
    else:
        v_render = v_render.contiguous()
        require_gpu(v_render)
        check(lib.lfs_rasterize_to_pixels_from_world_3dgs_bwd_prepared_acc(
            C.c_uint32(N), ptr(means), ptr(quats), ptr(scales), ptr(colors), ptr(opac), ptr(bg), C.byref(cams), C.c_uint32(tile), ptr(offsets), ptr(flatten_ids),
            C.c_int64(flatten_ids.shape[0]), ptr(alpha), ptr(last_ids), ptr(v_render), None, ptr(ws), C.c_size_t(ws.numel()), stream()),
            "rasterize_bwd_prepared_acc")
    v_colors = v_colors_out if v_colors_out is not None else torch.empty((1, N, 3), dtype=means.dtype, device=means.device)
    assert tuple(v_colors.shape) == (1, N, 3) and v_colors.is_contiguous()
    require_gpu(g_means, g_scales, g_quats, g_opac, raw_quats)
    check(lib.lfs_gut_finish_grads(C.c_uint32(N), ptr(means), ptr(raw_quats), ptr(quats), ptr(scales), ptr(opac), C.c_float(scale_reg), C.c_float(opacity_reg),
                                   C.c_int(int(accumulate)), ptr(g_means), ptr(g_scales), ptr(g_quats), ptr(g_opac), ptr(v_colors),
                                   ptr(loss_acc) if target_chw is not None else None, ptr(ws), C.c_size_t(ws.numel()), stream()), "gut_finish_grads")
    return v_colors


def sh_model_fwd_views(sh_degree: int, means, viewmats, sh0, shN, radii_views):
    """Owner side of dist.ShExchange: means / sh0 / shN = the owner's n rows; viewmats [V,4,4]; radii_views [V,S,2] (first n rows of each view
    used) -> colours [V,S,3] (rows >= n zero)."""
    require_gpu(means, viewmats, sh0, shN, radii_views)
    n, K, V, S = means.shape[0], 1 + shN.shape[1], radii_views.shape[0], radii_views.shape[1]
    colors = torch.empty((V, S, 3), dtype=means.dtype, device=means.device)   # the kernel writes every row < n of every view
    if S > n:
        colors[:, n:].zero_()
    check(load_library().lfs_sh_model_fwd_views(C.c_uint32(n), C.c_uint32(K), C.c_uint32(sh_degree), C.c_uint32(V), C.c_uint32(S), ptr(means), ptr(viewmats),
                                                ptr(sh0), ptr(shN), ptr(radii_views), ptr(colors), stream()), "sh_model_fwd_views")
    return colors


def sh_model_bwd_views(sh_degree: int, means, viewmats, sh0, shN, radii_views, colors_views, v_colors_views, v_sh0, v_shN, v_means, accumulate: bool,
                       adam: Optional[dict] = None):
    """The backward of sh_model_fwd_views summed over the views: v_sh0 / v_shN (the owner's rows) written or added to, v_means += dL/d(dirs);
    with `adam` (FusedAdam.prepare_inline; accumulate False) v_shN is not touched and shN is updated in place. radii_views = colors_views = None: the rows of
    v_colors_views [V,S,3] come PRE-MASKED (visibility and colour > 0 applied by the rank that rendered the view - the factored gradient exchange,
    dist.ColorGradExchange); v_shN = None without adam: the higher-degree gradient is not wanted (iteration <= 1000)."""
    require_gpu(means, viewmats, sh0, shN, v_colors_views, v_sh0, v_means)
    if radii_views is not None:
        require_gpu(radii_views, colors_views)
    n, K, V, S = means.shape[0], 1 + shN.shape[1], v_colors_views.shape[0], v_colors_views.shape[1]
    z = C.c_float(0.0)
    sc = [C.c_float(adam[k]) for k in ("lr", "beta1", "beta2", "eps", "bc1_rcp", "bc2_sqrt_rcp")] if adam else [z] * 6
    check(load_library().lfs_sh_model_bwd_views(C.c_uint32(n), C.c_uint32(K), C.c_uint32(sh_degree), C.c_uint32(V), C.c_uint32(S), ptr(means), ptr(viewmats),
                                                ptr(sh0), ptr(shN), ptr(radii_views), ptr(colors_views), ptr(v_colors_views), C.c_int(int(accumulate)),
                                                ptr(v_sh0), None if (adam or v_shN is None) else ptr(v_shN), ptr(v_means), ptr(adam["exp_avg"]) if adam else None,
                                                ptr(adam["exp_avg_sq"]) if adam else None, *sc, stream()), "sh_model_bwd_views")


def mse_loss_fwd_bwd(render_hwc, target_chw, weight: float, loss_acc):
    """loss_acc (1-element tensor) += weight * mse(clamp(render, 0, 1), target); returns dL/d(render) [H,W,3]."""
    target_chw = target_chw.contiguous()  # (a CHW view of an HWC render is a common caller mistake; no-op otherwise)
    require_gpu(render_hwc, target_chw, loss_acc)
    H, W = render_hwc.shape[-3], render_hwc.shape[-2]
    assert render_hwc.shape[-1] == 3 and tuple(target_chw.shape) == (3, H, W), (render_hwc.shape, target_chw.shape)
    v = torch.empty_like(render_hwc)
    check(load_library().lfs_mse_loss_fwd_bwd(C.c_uint32(H), C.c_uint32(W), ptr(render_hwc), ptr(target_chw), C.c_float(weight),
                                              ptr(v), ptr(loss_acc), stream()), "mse_loss_fwd_bwd")
    return v


FUSE_MSE_INTO_BACKWARD = True   # False: separate lfs_mse_loss_fwd_bwd launch (tests compare the two)
OVERLAP_SH_EXCHANGE = True       # SH-sharded: the radii / colour all-to-alls run next to the intersection kernels (False: blocking, A/B and debugging)
FUSE_FINISH_GRADS = True         # accumulator rows -> raw-parameter gradient tensors in one pass (False: raster_finish + activations_bwd + copy; tests compare)
BEGIN_ALL_INTERSECTIONS = True   # multi-view steps: the count kernels of all views up front, one host wait per step (False: one per view)
OVERLAP_SH_WITH_READBACK = True  # False: SH colours first, then the blocking n_isects read-back (A/B timing)


@dataclass
class FusedStepOutput:
    image_hwc: torch.Tensor      # [1,H,W,3] un-clamped render (rasterizer output)
    alpha: torch.Tensor          # [1,H,W,1]
    radii: torch.Tensor          # [1,N,2]
    n_isects: int
    v_colors: Optional[torch.Tensor] = None   # [N,3] dL/dcolour (defer_sh_backward: the caller runs the SH backward)


def render_and_backward(camera: Camera, model: SplatModel, bg_color: Optional[torch.Tensor], target_chw: torch.Tensor, weight: float,
                        grads: List[torch.Tensor], loss_acc: torch.Tensor, accumulate: bool, loss: str = "mse",
                        lambda_dssim: float = 0.2, scale_reg: float = 0.0, opacity_reg: float = 0.0, sh_exchange=None, viewmats_all=None, adam_shN: Optional[dict] = None, adam_shard: Optional[dict] = None, bilateral=None, image_idx: int = 0,
                        on_geometry_grads=None, adam_all: Optional[dict] = None, given: Optional[tuple] = None, defer_sh_backward: bool = False,
                        v_colors_out: Optional[torch.Tensor] = None) -> FusedStepOutput:
    """One view: forward, loss against `target_chw` ("mse": the rasterizer-only metric of SURVEY.md §8d; "l1_ssim": the reference's
    photometric loss, trainer.cpp:115-128), backward. `grads` = six tensors shaped like model.parameters()
    (means, sh0, shN, raw_scales, raw_quats, raw_opacities); written when accumulate is False, added to otherwise.
    Constants as in rasterizer.cpp:176-181.
    With `sh_exchange` (dist.ShExchange; multi-GPU) model.shN / grads[2] hold only this rank's rows and the SH stages run on the
    owners: `viewmats_all[j]` is the view matrix rank j renders in this call (every rank calls this the same number of times).
    With `adam_shN` (FusedAdam.prepare_inline; one view per step on one rank) grads[2] is not written: shN is updated in place;
    `adam_shard` is the same for the owner's rows under `sh_exchange` (one view per rank and step).
    `bilateral` (bilateral_grid.BilateralGrid): the rendered image goes through grid `image_idx` before the loss (trainer.cpp:662-664); the grid's
    gradient is accumulated into its .grad.
    `on_geometry_grads()` (multi-GPU, last view of the step) is called as soon as the raw scale / quaternion / opacity gradients are final - before the
    SH backward is enqueued - so their all-reduce (dist.GradBucket.all_reduce_early) overlaps with it on RCCL's stream.
    `adam_all` ({parameter name: FusedAdam.prepare_inline(...)} for all six; one view per step on one rank, MSE loss): nothing is written to `grads` -
    every parameter is updated in place by the backward kernels themselves (backward_adam_all).
    `given` = (quats, scales, opacities, radii, means2d, depths, colours) of this view, computed by the caller (render_views_and_backward: projection of
    all views first, then ONE SH launch for all of them); `defer_sh_backward`: the SH backward is the caller's too - dL/dcolour comes back in the
    output, grads[1] / grads[2] are not touched, grads[0] receives the rasterizer's part of dL/dmeans."""
    assert camera.camera_model_type == CameraModelType.PINHOLE and camera.radial_distortion is None and camera.tangential_distortion is None, \
        "the fused path covers the trainer's undistorted pinhole cameras; use rasterizer.rasterize for the rest"
    W, H = int(camera.image_width), int(camera.image_height)
    viewmat, Kmat = camera.world_view_transform.contiguous(), camera.K.contiguous()
    means, sh0, shN, raw_scales, raw_quats, raw_opac = [p.detach() for p in model.parameters()]
    g_means, g_sh0, g_shN, g_scales, g_quats, g_opac = grads
    deg = model.get_active_sh_degree()
    ut = UnscentedTransformParameters()
    tile = 16
    tw, th = (W + tile - 1) // tile, (H + tile - 1) // tile
    with torch.no_grad():
        bg = None if bg_color is None else bg_color.view(1, -1).contiguous()
        isect_state = None
        if given is not None:
            quats, scales, opac, radii, means2d, depths, given_colors = given[:7]
            isect_state = given[7] if len(given) > 7 else None      # ops.intersect_tile_begin of this view, issued by the caller
        elif FUSE_ACT_PROJ:
            quats, scales, opac, radii, means2d, depths = activations_project(means, raw_quats, raw_scales, raw_opac, viewmat, Kmat, W, H, ut)
        else:
            quats, scales, opac = activations_fwd(raw_quats, raw_scales, raw_opac)
            radii, means2d, depths, _, _ = ops.projection_ut_3dgs_fused(means, quats, scales, opac, viewmat, None, Kmat, W, H, 0.3, 0.01, 10000.0, 0.0,
                                                                        False, CameraModelType.PINHOLE, ut, ShutterType.GLOBAL, None, None, None)
        # SH-sharded: this rank's radii start their way to the SH owners now (next to the intersection count), the colours come back while the
        # intersection scatter and sort run; finish_forward() below is the stream-side wait in front of the first kernel that reads them
        radii_pending = sh_exchange.begin_radii(radii[0]) if (sh_exchange is not None and OVERLAP_SH_EXCHANGE) else None

        # SH colours do not depend on the tile lists: they are enqueued while the host waits for n_isects (ops.intersect_tile `overlap`)
        def sh_stage():
            if given is not None:
                return given_colors, None
            if sh_exchange is None:
                return sh_model_fwd(deg, means, viewmat, sh0, shN, radii), None
            return sh_exchange.forward(deg, means, sh0, shN, radii[0], viewmats_all, sh_model_fwd_views, radii_pending=radii_pending, defer=OVERLAP_SH_EXCHANGE)
        if isect_state is not None:
            (colors, sh_ctx) = sh_stage()
            _, _, flatten_ids, offsets = ops.intersect_tile_finish(isect_state)
        elif OVERLAP_SH_WITH_READBACK:
            _, _, flatten_ids, offsets, (colors, sh_ctx) = ops.intersect_tile(means2d, radii, depths, None, None, 1, tile, tw, th, True, return_offsets=True,
                                                                               overlap=sh_stage)
        else:
            colors, sh_ctx = sh_stage()
            _, _, flatten_ids, offsets = ops.intersect_tile(means2d, radii, depths, None, None, 1, tile, tw, th, True, return_offsets=True)
        if sh_exchange is not None:
            sh_exchange.finish_forward(sh_ctx)
        fwd_args = (means, quats, scales, colors.unsqueeze(0), opac.unsqueeze(0), bg, None, W, H, tile, viewmat, None, Kmat,
                    CameraModelType.PINHOLE, ut, ShutterType.GLOBAL, None, None, None, offsets, flatten_ids)
        render, alpha, last_ids, ws = ops.rasterize_to_pixels_from_world_3dgs_fwd(*fwd_args, own_workspace=True)
        fuse_mse = loss == "mse" and FUSE_MSE_INTO_BACKWARD and flatten_ids.shape[0] > 0 and bilateral is None
        if bilateral is not None:    # clamp (rasterizer.cpp:399 / bilateral_grid.cpp:115) -> slice -> loss on the un-clamped result -> slice backward
            from .losses import loss_fwd_bwd
            shown = bilateral.apply_fused(render[0], image_idx, chw=False)
            v_shown = loss_fwd_bwd(loss, shown, target_chw, weight, loss_acc, chw=False, clamp=False, lambda_dssim=lambda_dssim)
            v_render = bilateral.apply_fused_backward(render[0], image_idx, v_shown, chw=False).unsqueeze(0)
        elif fuse_mse:
            v_render = None          # derived inside the rasterizer backward from `render` and the target
        elif adam_all is not None:   # the all-inline caller does not zero loss_acc (lfs_gut_finish_adam stores); nothing was rendered: separate kernels below
            loss_acc.zero_()
            v_render = mse_loss_fwd_bwd(render, target_chw, weight, loss_acc)
        elif loss == "mse":
            v_render = mse_loss_fwd_bwd(render, target_chw, weight, loss_acc)
        elif loss == "l1_ssim":
            from .losses import photometric_loss_fwd_bwd
            v_render = photometric_loss_fwd_bwd(render, target_chw, lambda_dssim, weight, loss_acc)
        else:
            raise ValueError(f"unknown loss {loss!r}")
        adam_fallback = None
        if adam_all is not None and not fuse_mse:   # nothing was rendered (no intersections): separate kernels, then the prepared Adam steps by hand
            adam_fallback, adam_all = adam_all, None
        if adam_all is not None:
            assert sh_exchange is None and not accumulate and adam_shN is None, "the all-inline step: one view, one rank, MSE folded into the backward"
            backward_adam_all(deg, means, raw_quats, raw_scales, raw_opac, sh0, shN, quats, scales, opac, colors, radii, bg, W, H, tile, viewmat, Kmat, offsets,
                              flatten_ids, render, alpha, last_ids, target_chw, weight, loss_acc, ws, adam_all, scale_reg, opacity_reg)
            return FusedStepOutput(render, alpha, radii, int(flatten_ids.shape[0]))
        fused_finish = FUSE_FINISH_GRADS and flatten_ids.shape[0] > 0 and all(g.is_contiguous() for g in (g_means, g_scales, g_quats, g_opac))
        if fused_finish:   # backward kernel + ONE pass from the accumulator rows to the raw-parameter gradients (g_means: the rasterizer's part so far)
            v_colors = backward_grads(means, raw_quats, quats, scales, opac, colors.unsqueeze(0), bg, W, H, tile, viewmat, Kmat, offsets, flatten_ids, render, alpha,
                                      last_ids, ws, g_means, g_scales, g_quats, g_opac, accumulate, scale_reg, opacity_reg,
                                      target_chw=target_chw if fuse_mse else None, weight=weight, loss_acc=loss_acc, v_render=None if fuse_mse else v_render,
                                      v_colors_out=v_colors_out)
            sh_means = g_means          # the SH backward adds dL/d(dirs) straight onto the caller's dL/dmeans
        else:
            if fuse_mse:
                v_means, v_quats, v_scales, v_colors, v_opac = ops.rasterize_bwd_prepared_mse(
                    means, quats, scales, colors.unsqueeze(0), opac.unsqueeze(0), bg, W, H, tile, viewmat, Kmat, CameraModelType.PINHOLE, ShutterType.GLOBAL,
                    offsets, flatten_ids, render, alpha, last_ids, target_chw, weight, loss_acc, ws, v_colors_out=v_colors_out)
            else:
                v_means, v_quats, v_scales, v_colors, v_opac = ops.rasterize_to_pixels_from_world_3dgs_bwd(
                    *fwd_args, alpha, last_ids, v_render, None, prepared_workspace=ws, v_colors_out=v_colors_out)
            # the regularisers of trainer.cpp:132-158 are per step, not per view: the caller passes them with the first view only
            activations_bwd(raw_quats, scales, opac, v_quats, v_scales, v_opac.squeeze(0), g_quats, g_scales, g_opac, accumulate, scale_reg, opacity_reg)
            sh_means = v_means
        if on_geometry_grads is not None:
            on_geometry_grads()
        if defer_sh_backward:
            assert sh_exchange is None and adam_shN is None
            if not fused_finish:
                (g_means.add_ if accumulate else g_means.copy_)(v_means)
            return FusedStepOutput(render, alpha, radii, int(flatten_ids.shape[0]), v_colors.squeeze(0))
        # SH backward adds dL/d(dirs) onto the rasterizer's dL/d(means) in place, then the means gradient lands in the caller's buffer
        if adam_shN is not None:     # single view, single rank: shN's gradient is consumed by its Adam update inside the SH backward
            assert sh_exchange is None and not accumulate
            sh_model_bwd_adam(deg, means, viewmat, sh0, shN, radii, colors, v_colors.squeeze(0), g_sh0, sh_means, adam_shN)
        elif sh_exchange is None:
            sh_model_bwd(deg, means, viewmat, sh0, shN, radii, colors, v_colors.squeeze(0), g_sh0, g_shN, sh_means, accumulate)
        if not fused_finish:
            if accumulate:
                g_means.add_(v_means)
            else:
                g_means.copy_(v_means)
        if sh_exchange is not None:  # owners: SH backward of every rank's view for their rows (dL/d(dirs) straight into g_means)
            sh_exchange.backward(sh_ctx, deg, means, sh0, shN, viewmats_all, v_colors.squeeze(0), g_sh0, g_shN, g_means, accumulate, sh_model_bwd_views,
                                 adam=adam_shard)
        if adam_fallback is not None:
            # the all-inline step saw a view without a single intersection: the separate kernels above wrote the (regulariser-only) gradients, and the Adam
            # updates FusedAdam.prepare_inline() promised - step counts are already advanced, optimizer.step() will skip all six - are applied here with
            # the prepared scalars: moments decay and the momentum moves the parameters, as in the reference's FusedAdam::step on such a step
            names = ("means", "sh0", "shN", "raw_scales", "raw_quats", "raw_opacities")
            params6 = (means, sh0, shN, raw_scales, raw_quats, raw_opac)
            ops.adam_step_multi([(p_, adam_fallback[n_]["exp_avg"], adam_fallback[n_]["exp_avg_sq"], g_.contiguous(), adam_fallback[n_]["lr"], adam_fallback[n_]["beta1"],
                                  adam_fallback[n_]["beta2"], adam_fallback[n_]["eps"], adam_fallback[n_]["bc1_rcp"], adam_fallback[n_]["bc2_sqrt_rcp"])
                                 for n_, p_, g_ in zip(names, params6, grads) if p_.numel()])
    return FusedStepOutput(render, alpha, radii, int(flatten_ids.shape[0]))


def render_views_and_backward(cameras: List[Camera], model: SplatModel, bg_color: Optional[torch.Tensor], targets: List[torch.Tensor], weight: float,
                              grads: List[torch.Tensor], loss_acc: torch.Tensor, loss: str = "mse", lambda_dssim: float = 0.2, scale_reg: float = 0.0,
                              opacity_reg: float = 0.0, adam_shN: Optional[dict] = None, bilateral=None, image_idxs: Optional[List[int]] = None) -> List[FusedStepOutput]:
    """Several views of ONE step on one rank (BASELINE config 4: 8 views per GPU and step): what render_and_backward does view by view, with the
    spherical-harmonics stages batched over the views - the projections of all views first, ONE launch for the colours of all views
    (lfs_sh_model_fwd_views: a Gaussian's coefficient row is read once, not once per view), the rasterizer forward / loss / backward view by view, then
    ONE SH backward over all views (lfs_sh_model_bwd_views) that sums basis x dL/dcolour over the views in registers - the 180 B / Gaussian shN
    gradient is written once instead of read-modify-written per view, or, with `adam_shN` (FusedAdam.prepare_inline), not at all: the kernel applies
    shN's Adam update on the spot. Same arithmetic per (view, Gaussian) as the view-by-view path; the sums over the views are formed in a different
    order (tests/test_gpu_fused.py). grads are written (not added to). The regularisers count once per step, as in render_and_backward."""
    V = len(cameras)
    assert V >= 1 and len(targets) == V
    means, sh0, shN, raw_scales, raw_quats, raw_opac = [p.detach() for p in model.parameters()]
    g_means, g_sh0, g_shN = grads[0], grads[1], grads[2]
    deg = model.get_active_sh_degree()
    ut = UnscentedTransformParameters()
    with torch.no_grad():
        pre = []
        radii_all = torch.empty((V, means.shape[0], 2), dtype=torch.int32, device=means.device)   # [V,N,2]: every projection writes its slice
        for k, cam in enumerate(cameras):
            W, H = int(cam.image_width), int(cam.image_height)
            pre.append(activations_project(means, raw_quats, raw_scales, raw_opac, cam.world_view_transform.contiguous(), cam.K.contiguous(), W, H, ut,
                                           radii_out=radii_all[k:k + 1]))
        vms = torch.cat([cam.world_view_transform.reshape(1, 4, 4) for cam in cameras]).contiguous()
        # the intersection counts of all views now, each into its own workspace and pinned counter pair: the per-view host wait for n_isects
        # (the GPU idles ~30 us through each: the queue has to drain first) happens once per step
        isect = []
        if BEGIN_ALL_INTERSECTIONS:
            for k, cam in enumerate(cameras):
                W, H = int(cam.image_width), int(cam.image_height)
                isect.append(ops.intersect_tile_begin(pre[k][4], pre[k][3], pre[k][5], 1, 16, (W + 15) // 16, (H + 15) // 16, True, return_offsets=True,
                                                      pinned=True, slot=k))
        colors_all = sh_model_fwd_views(deg, means, vms, sh0, shN, radii_all)                      # [V,N,3]
        v_colors_all = torch.empty_like(colors_all)
    outs = []
    for k, cam in enumerate(cameras):
        quats, scales, opac, radii, means2d, depths = pre[k]
        out = render_and_backward(cam, model, bg_color, targets[k], weight, grads, loss_acc, accumulate=k > 0, loss=loss, lambda_dssim=lambda_dssim,
                                  scale_reg=scale_reg if k == 0 else 0.0, opacity_reg=opacity_reg if k == 0 else 0.0, bilateral=bilateral,
                                  image_idx=image_idxs[k] if image_idxs is not None else k,
                                  given=(quats, scales, opac, radii, means2d, depths, colors_all[k]) + ((isect[k],) if isect else ()), defer_sh_backward=True,
                                  v_colors_out=v_colors_all[k:k + 1])          # dL/dcolour of the view lands in its slice: no copy
        out.v_colors = None
        outs.append(out)
    with torch.no_grad():
        sh_model_bwd_views(deg, means, vms, sh0, shN, radii_all, colors_all, v_colors_all, g_sh0, g_shN, g_means, False, adam_shN)
    return outs
