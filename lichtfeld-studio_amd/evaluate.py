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
def evaluate(model, cameras: List, images: List[torch.Tensor], iteration: int = 0, background: Optional[torch.Tensor] = None) -> EvalMetrics:
    """cameras: rasterizer.Camera per validation view; images: the ground truth [3,H,W] in [0,1]."""
    from .fastgs import fast_rasterize
    dev = model.means.device
    bg = background if background is not None else torch.zeros(3, device=dev)
    ps, ss = [], []
    for cam, gt in zip(cameras, images):
        img = torch.clamp(fast_rasterize(cam, model, bg).image, 0.0, 1.0)
        ps.append(psnr(img, gt.to(dev)))
        ss.append(ssim(img, gt.to(dev)))
    n = max(len(ps), 1)
    return EvalMetrics(sum(ps) / n, sum(ss) / n, int(model.means.shape[0]), iteration, len(ps))
