"""Mirror of the reference's bilateral-grid appearance model (BASELINE config 5) over csrc/bilateral_grid.hip:
`slice_forward` / `slice_backward` / `tv_loss_forward` / `tv_loss_backward` (include/kernels/bilateral_grid.cuh:12-33), the two
autograd Functions and the `BilateralGrid` module of src/training/components/bilateral_grid.cpp:10-136 (identity-initialised
[N,12,L,H,W] grids; `apply(rgb [3,h,w] | [1,3,h,w], image_idx)` clamps to [0,1] and slices; `tv_loss()`).
Extension: `apply_fwd_bwd`-style entry points for the no-autograd trainer take a `chw` flag so neither rasterizer layout needs a permute.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import torch

from .capi import LfsError, check, load_library, ptr, require_gpu, stream


def _u(v):
    return C.c_uint32(int(v))


def _grid_dims(grid: torch.Tensor):
    if grid.dim() != 4 or grid.shape[0] != 12:
        raise LfsError("Grid must be [12, L, H, W]")
    return grid.shape[1], grid.shape[2], grid.shape[3]


def _image_dims(rgb: torch.Tensor, chw: bool):
    if rgb.dim() != 3 or rgb.shape[0 if chw else 2] != 3:
        raise LfsError("RGB must be [3, H, W]" if chw else "RGB must be [H, W, 3]")
    return (rgb.shape[1], rgb.shape[2]) if chw else (rgb.shape[0], rgb.shape[1])


def slice_forward(grid: torch.Tensor, rgb: torch.Tensor, chw: bool = False, clamp_input: bool = False) -> torch.Tensor:
    """gs::bilateral_grid::slice_forward_cuda (bilateral_grid_forward.cu:96-115): grid [12,L,H,W], rgb [h,w,3] -> [h,w,3]."""
    grid, rgb = grid.contiguous(), rgb.contiguous()
    require_gpu(grid, rgb)
    L, H, W = _grid_dims(grid)
    h, w = _image_dims(rgb, chw)
    out = torch.empty_like(rgb)
    check(load_library().lfs_bilateral_slice_fwd(_u(L), _u(H), _u(W), _u(h), _u(w), ptr(grid), ptr(rgb), _u(chw), _u(clamp_input), ptr(out), stream()),
          "bilateral_grid::slice_forward")
    return out


def slice_backward(grid: torch.Tensor, rgb: torch.Tensor, grad_output: torch.Tensor, chw: bool = False, clamp_input: bool = False,
                   grad_grid: torch.Tensor | None = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """gs::bilateral_grid::slice_backward_cuda (bilateral_grid_backward.cu:155-183) -> (grad_grid, grad_rgb).
    Passing `grad_grid` accumulates into it (the trainer's parameter gradient) instead of allocating a zeroed tensor."""
    grid, rgb, grad_output = grid.contiguous(), rgb.contiguous(), grad_output.contiguous()
    require_gpu(grid, rgb, grad_output)
    L, H, W = _grid_dims(grid)
    h, w = _image_dims(rgb, chw)
    if grad_grid is None:
        grad_grid = torch.zeros_like(grid)
    elif not grad_grid.is_contiguous() or grad_grid.shape != grid.shape:
        raise LfsError("grad_grid must be a contiguous tensor of the grid's shape")
    grad_rgb = torch.empty_like(rgb)
    check(load_library().lfs_bilateral_slice_bwd(_u(L), _u(H), _u(W), _u(h), _u(w), ptr(grid), ptr(rgb), ptr(grad_output), _u(chw), _u(clamp_input),
                                                 ptr(grad_grid), ptr(grad_rgb), stream()), "bilateral_grid::slice_backward")
    return grad_grid, grad_rgb


def tv_loss_forward(grids: torch.Tensor, weight: float = 1.0, loss_acc: torch.Tensor | None = None) -> torch.Tensor:
    """gs::bilateral_grid::tv_loss_forward_cuda (bilateral_grid_tv.cu:137-160): grids [N,12,L,H,W] -> scalar.
    With `loss_acc` (a zero-dim or [1] float tensor) adds weight * tv into it instead."""
    grids = grids.contiguous()
    require_gpu(grids)
    if grids.dim() != 5 or grids.shape[1] != 12:
        raise LfsError("Grids must be [N, 12, L, H, W]")
    N, _, L, H, W = grids.shape
    out = loss_acc if loss_acc is not None else torch.zeros((), dtype=grids.dtype, device=grids.device)
    check(load_library().lfs_bilateral_tv_loss_fwd(_u(N), _u(L), _u(H), _u(W), ptr(grids), C.c_float(weight), ptr(out), stream()), "bilateral_grid::tv_loss_forward")
    return out


def tv_loss_backward(grids: torch.Tensor, grad_output, grad_grids: torch.Tensor | None = None) -> torch.Tensor:
    """gs::bilateral_grid::tv_loss_backward_cuda (bilateral_grid_tv.cu:162-188); grad_output is read on the host, as there (:181)."""
    grids = grids.contiguous()
    require_gpu(grids)
    N, _, L, H, W = grids.shape
    g = float(grad_output.item()) if isinstance(grad_output, torch.Tensor) else float(grad_output)
    accumulate = grad_grids is not None
    if grad_grids is None:
        grad_grids = torch.empty_like(grids)
    check(load_library().lfs_bilateral_tv_loss_bwd(_u(N), _u(L), _u(H), _u(W), ptr(grids), C.c_float(g), _u(accumulate), ptr(grad_grids), stream()),
          "bilateral_grid::tv_loss_backward")
    return grad_grids


class _Slice(torch.autograd.Function):
    """BilateralGridSliceFunction (components/bilateral_grid.cpp:11-53)"""

    @staticmethod
    def forward(ctx, grid, rgb):
        ctx.save_for_backward(grid, rgb)
        return slice_forward(grid, rgb)

    @staticmethod
    def backward(ctx, grad_output):
        grid, rgb = ctx.saved_tensors
        return slice_backward(grid, rgb, grad_output)


class _TV(torch.autograd.Function):
    """BilateralGridTVLossFunction (components/bilateral_grid.cpp:56-77)"""

    @staticmethod
    def forward(ctx, grids):
        ctx.save_for_backward(grids)
        return tv_loss_forward(grids)

    @staticmethod
    def backward(ctx, grad_output):
        return tv_loss_backward(ctx.saved_tensors[0], grad_output)


class BilateralGrid:
    """gs::training::BilateralGrid (components/bilateral_grid.hpp, .cpp:80-136)."""

    def __init__(self, num_images: int, grid_W: int = 16, grid_H: int = 16, grid_L: int = 8, device="cuda:0"):
        self.num_images, self.grid_width, self.grid_height, self.grid_guidance = num_images, grid_W, grid_H, grid_L
        eye = torch.eye(4, dtype=torch.float32)[:3]                      # identity affine transform per cell (.cpp:88-92)
        grid = eye.repeat(grid_L * grid_H * grid_W, 1).reshape(1, grid_L, grid_H, grid_W, 12).permute(0, 4, 1, 2, 3)
        self.grids = grid.repeat(num_images, 1, 1, 1, 1).contiguous().to(device).requires_grad_(True)

    def parameters(self) -> torch.Tensor:
        return self.grids

    def apply(self, rgb: torch.Tensor, image_idx: int) -> torch.Tensor:
        if not 0 <= image_idx < self.num_images:
            raise LfsError(f"Invalid image index: {image_idx}")
        if rgb.dim() == 4 and rgb.shape[0] == 1:
            x = rgb[0]
        elif rgb.dim() == 3:
            x = rgb
        else:
            raise LfsError(f"RGB must be [C, H, W] or [1, C, H, W], got {tuple(rgb.shape)}")
        x = torch.clamp(x, 0, 1).permute(1, 2, 0).contiguous()
        out = _Slice.apply(self.grids[image_idx], x).permute(2, 0, 1).contiguous()
        return out[None] if rgb.dim() == 4 else out

    def tv_loss(self) -> torch.Tensor:
        return _TV.apply(self.grids)

    # ---- no-autograd path of the fused trainer: clamp + slice in the rasterizer's own layout, gradients in place ----
    def apply_fused(self, image: torch.Tensor, image_idx: int, chw: bool) -> torch.Tensor:
        return slice_forward(self.grids[image_idx].detach(), image, chw=chw, clamp_input=True)

    def apply_fused_backward(self, image: torch.Tensor, image_idx: int, grad_output: torch.Tensor, chw: bool) -> torch.Tensor:
        """Accumulates dL/dgrid into self.grids.grad[image_idx]; returns dL/dimage (through the clamp)."""
        if self.grids.grad is None:
            self.grids.grad = torch.zeros_like(self.grids)
        _, g = slice_backward(self.grids[image_idx].detach(), image, grad_output, chw=chw, clamp_input=True, grad_grid=self.grids.grad[image_idx])
        return g

    def tv_loss_fused(self, weight: float, loss_acc: torch.Tensor) -> None:
        """loss_acc += weight * tv and grids.grad += weight * dtv/dgrids."""
        if self.grids.grad is None:
            self.grids.grad = torch.zeros_like(self.grids)
        tv_loss_forward(self.grids.detach(), weight, loss_acc)
        tv_loss_backward(self.grids.detach(), weight, self.grids.grad)
