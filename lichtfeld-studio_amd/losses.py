"""Mirror of the reference's fused SSIM API over the HIP kernels of csrc/ssim.hip:
`fusedssim` / `fusedssim_backward` (include/kernels/ssim.cuh:11-30, src/training/kernels/ssim.cu:430-510), the autograd
wrapper `fused_ssim(img1, img2, padding, train)` (include/kernels/fused_ssim.cuh:30-131: "valid" crops 5 pixels per side when
H, W > 10; returns the MEAN of the map) and the photometric loss of Trainer::compute_photometric_loss
(src/training/trainer.cpp:115-128): (1 - lambda) * L1 + lambda * (1 - SSIM).
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import torch

from .capi import LfsError, check, load_library, ptr, require_gpu, stream, workspace

K_C1 = 0.01 * 0.01
K_C2 = 0.03 * 0.03


def fusedssim(C1: float, C2: float, img1: torch.Tensor, img2: torch.Tensor, train: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12), all [B,CH,H,W]; the derivative tensors are empty when train is False."""
    img1, img2 = img1.contiguous(), img2.contiguous()
    require_gpu(img1, img2)
    if img1.dim() != 4 or img1.shape != img2.shape:
        raise LfsError("fusedssim expects two [B,CH,H,W] tensors of the same shape")
    B, CH, H, W = img1.shape
    ssim_map = torch.empty_like(img1)
    if train:
        dm1, ds1, ds12 = torch.empty_like(img1), torch.empty_like(img1), torch.empty_like(img1)
    else:
        dm1 = ds1 = ds12 = None
    check(load_library().lfs_fused_ssim_fwd(C.c_uint32(B), C.c_uint32(CH), C.c_uint32(H), C.c_uint32(W), C.c_float(C1), C.c_float(C2),
                                            ptr(img1), ptr(img2), ptr(ssim_map), ptr(dm1), ptr(ds1), ptr(ds12), stream()), "fusedssim")
    if not train:
        e = torch.empty(0, dtype=img1.dtype, device=img1.device)
        return ssim_map, e, e, e
    return ssim_map, dm1, ds1, ds12


def fusedssim_backward(C1: float, C2: float, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12) -> torch.Tensor:
    img1, img2, dL_dmap = img1.contiguous(), img2.contiguous(), dL_dmap.contiguous()
    dm_dmu1, dm_dsigma1_sq, dm_dsigma12 = dm_dmu1.contiguous(), dm_dsigma1_sq.contiguous(), dm_dsigma12.contiguous()
    require_gpu(img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12)
    B, CH, H, W = img1.shape
    out = torch.empty_like(img1)
    check(load_library().lfs_fused_ssim_bwd(C.c_uint32(B), C.c_uint32(CH), C.c_uint32(H), C.c_uint32(W), C.c_float(C1), C.c_float(C2),
                                            ptr(img1), ptr(img2), ptr(dL_dmap), ptr(dm_dmu1), ptr(dm_dsigma1_sq), ptr(dm_dsigma12), ptr(out),
                                            stream()), "fusedssim_backward")
    return out


class _FusedSSIM(torch.autograd.Function):
    """fused_ssim.cuh:30-107"""

    @staticmethod
    def forward(ctx, img1, img2, padding: str, train: bool):
        if padding not in ("same", "valid"):
            raise ValueError(f'fused_ssim: padding must be "same" or "valid" (got "{padding}")')
        img1, img2 = img1.contiguous(), img2.contiguous()
        if img1.dim() == 3:
            img1 = img1.unsqueeze(0)
        if img2.dim() == 3:
            img2 = img2.unsqueeze(0)
        if img1.dim() != 4 or img2.dim() != 4:
            raise ValueError("fused_ssim expects 4D tensors [N,C,H,W]")
        if img1.shape != img2.shape:
            raise ValueError("img1 and img2 must have the same shape")
        m, dm1, ds1, ds12 = fusedssim(K_C1, K_C2, img1, img2, train)
        h, w = m.shape[2], m.shape[3]
        if padding == "valid" and h > 10 and w > 10:
            m = m[:, :, 5:h - 5, 5:w - 5]
        ctx.save_for_backward(img1.detach(), img2, dm1, ds1, ds12)
        ctx.padding = padding
        return m

    @staticmethod
    def backward(ctx, grad_out):
        img1, img2, dm1, ds1, ds12 = ctx.saved_tensors
        dL_dmap = grad_out
        if ctx.padding == "valid":
            full = torch.zeros_like(img1)
            h, w = full.shape[2], full.shape[3]
            if h > 10 and w > 10:
                full[:, :, 5:h - 5, 5:w - 5] = dL_dmap
            dL_dmap = full
        return fusedssim_backward(K_C1, K_C2, img1, img2, dL_dmap, dm1, ds1, ds12), None, None, None


def fused_ssim(img1: torch.Tensor, img2: torch.Tensor, padding: str = "same", train: bool = True) -> torch.Tensor:
    """Mean SSIM (scalar tensor), differentiable w.r.t. img1 (fused_ssim.cuh:124-131)."""
    return _FusedSSIM.apply(img1.contiguous(), img2, padding, train).mean()


def photometric_loss(rendered: torch.Tensor, gt: torch.Tensor, lambda_dssim: float = 0.2) -> torch.Tensor:
    """Trainer::compute_photometric_loss (trainer.cpp:115-128) on [3,H,W] or [B,3,H,W] images, through autograd."""
    if rendered.dim() == 3:
        rendered = rendered.unsqueeze(0)
    if gt.dim() == 3:
        gt = gt.unsqueeze(0)
    l1 = torch.nn.functional.l1_loss(rendered, gt)
    ssim_loss = 1.0 - fused_ssim(rendered, gt, "valid", True)
    return (1.0 - lambda_dssim) * l1 + lambda_dssim * ssim_loss


def photometric_loss_fwd_bwd(render_hwc: torch.Tensor, target_chw: torch.Tensor, lambda_dssim: float, weight: float, loss_acc: torch.Tensor) -> torch.Tensor:
    """Fused extension used by fused.py: loss_acc += weight * photometric_loss(clamp(render, 0, 1), target); returns dL/d(render)
    in the rasterizer's HWC layout. Two kernel launches, no intermediate image tensors."""
    target_chw = target_chw.contiguous()
    require_gpu(render_hwc, target_chw, loss_acc)
    H, W = render_hwc.shape[-3], render_hwc.shape[-2]
    assert render_hwc.shape[-1] == 3 and tuple(target_chw.shape) == (3, H, W), (render_hwc.shape, target_chw.shape)
    lib = load_library()
    ws = workspace(lib.lfs_photometric_loss_workspace_bytes(C.c_uint32(H), C.c_uint32(W)), render_hwc.device, "photometric")
    v = torch.empty_like(render_hwc)
    check(lib.lfs_photometric_loss_fwd_bwd(C.c_uint32(H), C.c_uint32(W), ptr(render_hwc), ptr(target_chw), C.c_float(lambda_dssim), C.c_float(weight),
                                           ptr(v), ptr(loss_acc), ptr(ws), C.c_size_t(ws.numel()), stream()), "photometric_loss_fwd_bwd")
    return v


def photometric_loss_chw_fwd_bwd(render_chw: torch.Tensor, target_chw: torch.Tensor, lambda_dssim: float, weight: float, loss_acc: torch.Tensor) -> torch.Tensor:
    """The same for the fastgs rasterizer's CHW image, which reaches the loss un-clamped (fast_rasterizer.cpp:63): returns dL/d(render) [3,H,W]."""
    target_chw = target_chw.contiguous()
    require_gpu(render_chw, target_chw, loss_acc)
    H, W = render_chw.shape[-2], render_chw.shape[-1]
    assert render_chw.is_contiguous() and tuple(render_chw.shape[-3:]) == (3, H, W) and tuple(target_chw.shape) == (3, H, W), (render_chw.shape, target_chw.shape)
    lib = load_library()
    ws = workspace(lib.lfs_photometric_loss_workspace_bytes(C.c_uint32(H), C.c_uint32(W)), render_chw.device, "photometric")
    v = torch.empty_like(render_chw)
    check(lib.lfs_photometric_loss_chw_fwd_bwd(C.c_uint32(H), C.c_uint32(W), ptr(render_chw), ptr(target_chw), C.c_float(lambda_dssim), C.c_float(weight),
                                               ptr(v), ptr(loss_acc), ptr(ws), C.c_size_t(ws.numel()), stream()), "photometric_loss_chw_fwd_bwd")
    return v


def loss_fwd_bwd(kind: str, render: torch.Tensor, target_chw: torch.Tensor, weight: float, loss_acc: torch.Tensor, chw: bool, clamp: bool,
                 lambda_dssim: float = 0.2) -> torch.Tensor:
    """General fused loss: kind "mse" | "l1_ssim"; render [H,W,3] (chw False) or [3,H,W]; clamp = clamp(render, 0, 1) first (gradient masked).
    loss_acc += weight * loss; returns dL/d(render) in the render's layout."""
    target_chw = target_chw.contiguous()
    render = render.contiguous()
    require_gpu(render, target_chw, loss_acc)
    H, W = (render.shape[-2], render.shape[-1]) if chw else (render.shape[-3], render.shape[-2])
    assert tuple(target_chw.shape) == (3, H, W) and render.shape[0 if chw else -1] == 3, (render.shape, target_chw.shape)
    lib = load_library()
    v = torch.empty_like(render)
    if kind == "mse":
        check(lib.lfs_mse_loss_ex_fwd_bwd(C.c_uint32(H), C.c_uint32(W), ptr(render), C.c_uint32(int(chw)), C.c_uint32(int(clamp)), ptr(target_chw), C.c_float(weight),
                                          ptr(v), ptr(loss_acc), stream()), "mse_loss_ex_fwd_bwd")
    elif kind == "l1_ssim":
        ws = workspace(lib.lfs_photometric_loss_workspace_bytes(C.c_uint32(H), C.c_uint32(W)), render.device, "photometric")
        check(lib.lfs_photometric_loss_ex_fwd_bwd(C.c_uint32(H), C.c_uint32(W), ptr(render), C.c_uint32(int(chw)), C.c_uint32(int(clamp)), ptr(target_chw),
                                                  C.c_float(lambda_dssim), C.c_float(weight), ptr(v), ptr(loss_acc), ptr(ws), C.c_size_t(ws.numel()), stream()),
              "photometric_loss_ex_fwd_bwd")
    else:
        raise ValueError(f"unknown loss {kind!r}")
    return v
