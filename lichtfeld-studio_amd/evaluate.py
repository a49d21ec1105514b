"""Evaluation metrics of the reference's MetricsEvaluator (src/training/metrics/metrics.cpp): PSNR (:40-59, data range 1, MSE clamped at
1e-10), SSIM (:68-125: 11x11 Gaussian window, sigma 1.5, zero padding, mean over the map - the "same" mode of the fused SSIM kernel) and
the evaluation loop (:380-520): every validation view rendered with fast_rasterize (the EWA rasterizer, whatever was trained with),
clamped to [0,1], metrics averaged over the views. LPIPS needs the TorchScript VGG blob the reference ships separately
(.MISSING_LARGE_BLOBS) and is not computed.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import torch


def psnr(pred: torch.Tensor, target: torch.Tensor, data_range: float = 1.0) -> float:
    """[B,C,H,W] (or [C,H,W]) -> mean over the batch of 20 log10(data_range / sqrt(mse))."""
    if pred.shape != target.shape:
        raise ValueError("Prediction and target must have the same shape")
    if pred.dim() == 3:
        pred, target = pred[None], target[None]
    mse = (pred.contiguous() - target.contiguous()).pow(2).reshape(pred.shape[0], -1).mean(1, keepdim=True).clamp_min(1e-10)
    return float((20.0 * torch.log10(data_range / mse.sqrt())).mean())


def ssim(pred: torch.Tensor, target: torch.Tensor) -> float:
    from .losses import fused_ssim
    if pred.dim() == 3:
        pred, target = pred[None], target[None]
    return float(fused_ssim(pred.contiguous(), target.contiguous(), padding="same", train=False))


@dataclass
class EvalMetrics:
    psnr: float
    ssim: float
    num_gaussians: int
    iteration: int
    n_images: int


@torch.no_grad()
def evaluate(model, cameras: List, images: List[torch.Tensor], iteration: int = 0, background: Optional[torch.Tensor] = None,
             rasterizer: str = "fastgs") -> EvalMetrics:
    """cameras: rasterizer.Camera per validation view; images: the ground truth [3,H,W] in [0,1]. rasterizer="fastgs" is the reference's protocol
    (metrics.cpp:430 renders with fast_rasterize whatever was trained with); "gut" renders with the 3DGUT rasterizer instead - what a model trained with
    --gut was optimised for: 3DGUT has no screen-space dilation, the EWA renderer adds its 0.3-pixel low-pass to Gaussians the training shrank below a
    pixel (profiles/r04/scale_train_*: up to 13 dB between the two numbers on an MCMC model with noise injection)."""
    from .fastgs import fast_rasterize
    from .rasterizer import rasterize
    if rasterizer not in ("fastgs", "gut"):
        raise ValueError("rasterizer must be 'fastgs' or 'gut'")
    render = fast_rasterize if rasterizer == "fastgs" else rasterize
    dev = model.means.device
    bg = background if background is not None else torch.zeros(3, device=dev)
    ps, ss = [], []
    for cam, gt in zip(cameras, images):
        img = torch.clamp(render(cam, model, bg).image, 0.0, 1.0)
        ps.append(psnr(img, gt.to(dev)))
        ss.append(ssim(img, gt.to(dev)))
    n = max(len(ps), 1)
    return EvalMetrics(sum(ps) / n, sum(ss) / n, int(model.means.shape[0]), iteration, len(ps))
