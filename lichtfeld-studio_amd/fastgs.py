"""Mirror of the reference's DEFAULT training rasterizer over csrc/fastgs_{prep,blend}.hip:
`fast_gs::rasterization::forward_wrapper / backward_wrapper` (fastgs/rasterization/include/rasterization_api.h:27-75),
the autograd Function `FastGSRasterize` (src/training/rasterization/fast_rasterizer_autograd.cpp:8-186) and
`fast_rasterize` (fast_rasterizer.cpp:12-68: near 0.01, far 1e10, background blended outside the op).

The reference returns four opaque buffer tensors + five integers from the forward and hands them to the backward; here the
state is two workspace tensors (primitive / instance) and `n_instances`.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from .capi import LfsError, check, load_library, ptr, require_gpu, stream
from .rasterizer import Camera, RenderOutput, SplatModel


@dataclass
class FastGSSettings:  # rasterization_api.h:12-24
    cam_position: torch.Tensor
    active_sh_bases: int
    width: int
    height: int
    focal_x: float
    focal_y: float
    center_x: float
    center_y: float
    near_plane: float
    far_plane: float


ASYNC_READBACK = True   # False: n_instances via a stream synchronisation (A/B timing)


def _frame_args(s: FastGSSettings):
    return (C.c_uint32(s.active_sh_bases), C.c_uint32(s.width), C.c_uint32(s.height), C.c_float(s.focal_x), C.c_float(s.focal_y),
            C.c_float(s.center_x), C.c_float(s.center_y), C.c_float(s.near_plane), C.c_float(s.far_plane))


def forward_wrapper(means, scales_raw, rotations_raw, opacities_raw, sh_coefficients_0, sh_coefficients_rest, w2c, s: FastGSSettings):
    """-> (image [3,H,W], alpha [1,H,W], primitive_workspace, instance_workspace, n_instances)."""
    w2c = w2c.reshape(-1, 4, 4)[0].contiguous()
    cam_position = s.cam_position.reshape(-1)[:3].contiguous()
    opac = opacities_raw.reshape(-1)
    tensors = [means, scales_raw, rotations_raw, opac, sh_coefficients_0, sh_coefficients_rest, w2c, cam_position]
    tensors = [t.contiguous() for t in tensors]
    require_gpu(*tensors)
    means, scales_raw, rotations_raw, opac, sh0, shr, w2c, cam_position = tensors
    N = means.shape[0]
    total_rest = shr.shape[1] if shr.dim() == 3 else 0
    if s.active_sh_bases > 1 + total_rest:
        raise LfsError("active_sh_bases exceeds the stored SH coefficients")
    lib, dev = load_library(), means.device
    pws = torch.empty(lib.lfs_fastgs_primitive_workspace_bytes(C.c_uint32(N), C.c_uint32(s.width), C.c_uint32(s.height)), dtype=torch.uint8, device=dev)
    n_inst_dev = torch.zeros(1, dtype=torch.int64, device=dev)
    check(lib.lfs_fastgs_preprocess(C.c_uint32(N), ptr(means), ptr(scales_raw), ptr(rotations_raw), ptr(opac), ptr(sh0), ptr(shr), C.c_uint32(total_rest),
                                    ptr(w2c), ptr(cam_position), *_frame_args(s), ptr(n_inst_dev), ptr(pws), C.c_size_t(pws.numel()), stream()), "fastgs_preprocess")
    # the one host sync (forward.cu:114-117 reads n_visible_primitives and n_instances): waits for the read-back event the library queued before
    # its SH kernel, so the GPU keeps working through the host round trip
    if ASYNC_READBACK:
        n_host = C.c_int64(0)
        check(lib.lfs_fastgs_wait_n_instances(C.byref(n_host)), "fastgs_wait_n_instances")
        n_instances = int(n_host.value)
    else:
        n_instances = int(n_inst_dev.item())
    iws = torch.empty(max(256, lib.lfs_fastgs_instance_workspace_bytes(C.c_uint32(s.width), C.c_uint32(s.height), C.c_int64(n_instances))), dtype=torch.uint8, device=dev)
    image = torch.empty((3, s.height, s.width), dtype=means.dtype, device=dev)
    alpha = torch.empty((1, s.height, s.width), dtype=means.dtype, device=dev)
    check(lib.lfs_fastgs_render(C.c_uint32(N), C.c_uint32(s.width), C.c_uint32(s.height), C.c_int64(n_instances), ptr(pws), C.c_size_t(pws.numel()),
                                ptr(iws), C.c_size_t(iws.numel()), ptr(image), ptr(alpha), stream()), "fastgs_render")
    return image, alpha, pws, iws, n_instances


def backward_wrapper(densification_info: Optional[torch.Tensor], grad_image, grad_alpha, image, alpha, means, scales_raw, rotations_raw,
                     sh_coefficients_0, sh_coefficients_rest, primitive_workspace, instance_workspace, w2c, s: FastGSSettings, n_instances: int, out=None,
                     adam_sh_rest: Optional[dict] = None):
    """-> (grad_means, grad_scales_raw, grad_rotations_raw, grad_opacities_raw [N,1], grad_sh_coefficients_0, grad_sh_coefficients_rest);
    densification_info [2,N] (when given and non-empty) is accumulated into in place (kernels_backward.cuh:229-232).
    Extension `adam_sh_rest` (FusedAdam.prepare_inline of sh_coefficients_rest; single-view steps): the rest-coefficient gradient is not
    written, the coefficients and their moments are updated in place by the SH backward (lfs_fastgs_backward_adam)."""
    w2c = w2c.reshape(-1, 4, 4)[0].contiguous()
    cam_position = s.cam_position.reshape(-1)[:3].contiguous()
    grad_image, grad_alpha, alpha = grad_image.contiguous(), grad_alpha.contiguous(), alpha.contiguous()
    means, scales_raw, rotations_raw, shr = means.contiguous(), scales_raw.contiguous(), rotations_raw.contiguous(), sh_coefficients_rest.contiguous()
    sh0 = sh_coefficients_0.contiguous()  # (the reference's backward does not need sh0; the SH backward kernel here shares its operand list with the forward)
    require_gpu(grad_image, grad_alpha, alpha, means, scales_raw, rotations_raw, sh0, shr, w2c, cam_position)
    N = means.shape[0]
    total_rest = shr.shape[1] if shr.dim() == 3 else 0
    dens = densification_info if (densification_info is not None and densification_info.numel() > 0) else None
    if dens is not None and (tuple(dens.shape) != (2, N) or not dens.is_contiguous()):
        raise LfsError("densification_info must be a contiguous [2,N] tensor")
    if out is not None:  # extension: write into the caller's (contiguous) buffers, e.g. views of the data-parallel gradient bucket
        g_means, g_scales, g_rot, g_opac, g_sh0, g_shr = out
        require_gpu(*out)
    else:
        g_means, g_scales, g_rot = torch.empty_like(means), torch.empty_like(scales_raw), torch.empty_like(rotations_raw)
        g_opac = torch.empty((N, 1), dtype=means.dtype, device=means.device)
        g_sh0 = torch.empty((N, 1, 3), dtype=means.dtype, device=means.device)
        g_shr = torch.empty_like(shr)
    if adam_sh_rest is not None and total_rest > 0:
        a = adam_sh_rest
        if not sh_coefficients_rest.is_contiguous():
            raise LfsError("adam_sh_rest needs the parameter tensor itself (contiguous), it is updated in place")
        check(load_library().lfs_fastgs_backward_adam(
            C.c_uint32(N), ptr(means), ptr(scales_raw), ptr(rotations_raw), ptr(sh0), ptr(sh_coefficients_rest), C.c_uint32(total_rest), ptr(w2c), ptr(cam_position),
            *_frame_args(s), C.c_int64(n_instances), ptr(primitive_workspace), C.c_size_t(primitive_workspace.numel()), ptr(instance_workspace),
            C.c_size_t(instance_workspace.numel()), ptr(grad_image), ptr(grad_alpha), ptr(alpha), ptr(dens), ptr(g_means), ptr(g_scales), ptr(g_rot), ptr(g_opac),
            ptr(g_sh0), ptr(a["exp_avg"]), ptr(a["exp_avg_sq"]), C.c_float(a["lr"]), C.c_float(a["beta1"]), C.c_float(a["beta2"]), C.c_float(a["eps"]),
            C.c_float(a["bc1_rcp"]), C.c_float(a["bc2_sqrt_rcp"]), stream()), "fastgs_backward_adam")
        return g_means, g_scales, g_rot, g_opac, g_sh0, g_shr
    check(load_library().lfs_fastgs_backward(
        C.c_uint32(N), ptr(means), ptr(scales_raw), ptr(rotations_raw), ptr(sh0), ptr(shr), C.c_uint32(total_rest), ptr(w2c), ptr(cam_position), *_frame_args(s),
        C.c_int64(n_instances), ptr(primitive_workspace), C.c_size_t(primitive_workspace.numel()), ptr(instance_workspace), C.c_size_t(instance_workspace.numel()),
        ptr(grad_image), ptr(grad_alpha), ptr(alpha), ptr(dens), ptr(g_means), ptr(g_scales), ptr(g_rot), ptr(g_opac), ptr(g_sh0), ptr(g_shr), stream()), "fastgs_backward")
    return g_means, g_scales, g_rot, g_opac, g_sh0, g_shr


class FastGSRasterize(torch.autograd.Function):
    """fast_rasterizer_autograd.cpp:8-186"""

    @staticmethod
    def forward(ctx, means, scales_raw, rotations_raw, opacities_raw, sh0, sh_rest, w2c, densification_info, settings: FastGSSettings):
        image, alpha, pws, iws, n_instances = forward_wrapper(means, scales_raw, rotations_raw, opacities_raw, sh0, sh_rest, w2c, settings)
        ctx.save_for_backward(image, alpha, means, scales_raw, rotations_raw, sh0, sh_rest, w2c)
        ctx.state = (pws, iws, n_instances, settings, densification_info, opacities_raw.shape)
        return image, alpha

    @staticmethod
    def backward(ctx, grad_image, grad_alpha):
        image, alpha, means, scales_raw, rotations_raw, sh0, sh_rest, w2c = ctx.saved_tensors
        pws, iws, n_instances, settings, dens, opac_shape = ctx.state
        g = backward_wrapper(dens, grad_image, grad_alpha, image, alpha, means, scales_raw, rotations_raw, sh0, sh_rest, pws, iws, w2c, settings, n_instances)
        return g[0], g[1], g[2], g[3].reshape(opac_shape), g[4], g[5], None, None, None


def fast_rasterize(camera: Camera, model: SplatModel, bg_color: torch.Tensor, densification_info: Optional[torch.Tensor] = None) -> RenderOutput:
    """fast_rasterizer.cpp:12-68"""
    W, H = int(camera.image_width), int(camera.image_height)
    K = camera.K.reshape(-1, 3, 3)[0]
    w2c = camera.world_view_transform
    R, t = w2c.reshape(-1, 4, 4)[0][:3, :3], w2c.reshape(-1, 4, 4)[0][:3, 3]
    deg = model.get_active_sh_degree()
    settings = FastGSSettings(cam_position=(-(R.T @ t)).contiguous(), active_sh_bases=(deg + 1) ** 2, width=W, height=H,
                              focal_x=float(K[0, 0]), focal_y=float(K[1, 1]), center_x=float(K[0, 2]), center_y=float(K[1, 2]), near_plane=0.01, far_plane=1e10)
    image, alpha = FastGSRasterize.apply(model.means, model.raw_scales, model.raw_quats, model.raw_opacities, model.sh0, model.shN, w2c, densification_info, settings)
    image = image + (1.0 - alpha) * bg_color.view(3, 1, 1)
    return RenderOutput(image=image, alpha=alpha, depth=None, means2d=None, depths=None, radii=None, visibility=None, width=W, height=H, n_isects=0)


def mse_loss_chw_fwd_bwd(render_chw: torch.Tensor, target_chw: torch.Tensor, weight: float, loss_acc: torch.Tensor) -> torch.Tensor:
    """loss_acc += weight * mse(render, target) (no clamp: fast_rasterize hands the image on as is); returns dL/d(render) [3,H,W]."""
    target_chw = target_chw.contiguous()
    require_gpu(render_chw, target_chw, loss_acc)
    H, W = render_chw.shape[-2], render_chw.shape[-1]
    v = torch.empty_like(render_chw)
    check(load_library().lfs_mse_loss_chw_fwd_bwd(C.c_uint32(H), C.c_uint32(W), ptr(render_chw), ptr(target_chw), C.c_float(weight), ptr(v), ptr(loss_acc), stream()),
          "mse_loss_chw_fwd_bwd")
    return v


def render_and_backward(settings: FastGSSettings, w2c: torch.Tensor, model: SplatModel, target_chw: torch.Tensor, weight: float, grads, loss_acc: torch.Tensor,
                        densification_info: Optional[torch.Tensor] = None, loss: str = "mse", lambda_dssim: float = 0.2, bilateral=None, image_idx: int = 0,
                        adam_shN: Optional[dict] = None):
    """One training view without an autograd graph (black background; loss "mse" or the trainer's "l1_ssim", trainer.cpp:122-125):
    forward, [bilateral-grid slice, trainer.cpp:662-664,] loss, backward; the gradients of (means, sh0, shN, raw_scales, raw_quats,
    raw_opacities) are WRITTEN into `grads` (param-group order); the bilateral grid's gradient is accumulated into its .grad."""
    means, sh0, shN, raw_scales, raw_quats, raw_opac = [p.detach() for p in model.parameters()]
    g_means, g_sh0, g_shN, g_scales, g_quats, g_opac = grads
    with torch.no_grad():
        image, alpha, pws, iws, n_inst = forward_wrapper(means, raw_scales, raw_quats, raw_opac, sh0, shN, w2c, settings)
        shown = image if bilateral is None else bilateral.apply_fused(image, image_idx, chw=True)
        if loss == "l1_ssim":
            from .losses import photometric_loss_chw_fwd_bwd
            v_image = photometric_loss_chw_fwd_bwd(shown, target_chw, lambda_dssim, weight, loss_acc)
        elif loss == "mse":
            v_image = mse_loss_chw_fwd_bwd(shown, target_chw, weight, loss_acc)
        else:
            raise ValueError(f"unknown loss {loss!r}")
        if bilateral is not None:
            v_image = bilateral.apply_fused_backward(image, image_idx, v_image, chw=True)
        if not hasattr(render_and_backward, "_zero") or render_and_backward._zero.shape != alpha.shape or render_and_backward._zero.device != alpha.device:
            render_and_backward._zero = torch.zeros_like(alpha)
        backward_wrapper(densification_info, v_image, render_and_backward._zero, image, alpha, means, raw_scales, raw_quats, sh0, shN, pws, iws, w2c, settings, n_inst,
                         out=(g_means, g_scales, g_quats, g_opac.view(-1, 1), g_sh0, g_shN), adam_sh_rest=adam_shN)
    return image, alpha, n_inst
