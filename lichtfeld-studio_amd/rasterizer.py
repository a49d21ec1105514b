"""Host-side mirror of the reference's L2 render glue for the --gut path:
/root/reference/src/training/rasterization/rasterizer.cpp:46-430 (`rasterize`) and
rasterizer_autograd.cpp (`SphericalHarmonicsFunction` :15-133, `fully_fused_projection_with_ut`
:135-265, `GUTRasterizationFunction` :267-398). Same call order, constants and tensor shapes;
the ops underneath are the HIP kernels of liblfs_gsplat.so (see ops.py).
"""
from __future__ import annotations

import enum
from dataclasses import dataclass
from typing import Optional

import torch

from . import ops
from .capi import CameraModelType, ShutterType, UnscentedTransformParameters


class RenderMode(enum.IntEnum):  # src/training/rasterization/rasterizer.hpp
    RGB = 0
    D = 1
    ED = 2
    RGB_D = 3
    RGB_ED = 4


@dataclass
class Camera:
    """The slice of gs::Camera the render path reads (src/core/camera.cpp:15-100)."""
    world_view_transform: torch.Tensor  # [1,4,4] world->camera
    K: torch.Tensor                     # [1,3,3]
    image_width: int
    image_height: int
    camera_model_type: CameraModelType = CameraModelType.PINHOLE
    radial_distortion: Optional[torch.Tensor] = None
    tangential_distortion: Optional[torch.Tensor] = None


class SplatModel:
    """Raw parameters + activations of gs::SplatData (src/core/splat_data.cpp:267-286):
    get_opacity = sigmoid, get_scaling = exp, get_rotation = normalize, get_shs = cat(sh0, shN)."""

    def __init__(self, means, sh0, shN, raw_scales, raw_quats, raw_opacities, sh_degree: int, active_sh_degree: Optional[int] = None):
        """sh_degree: the degree the coefficient tensors hold (max_sh_degree of SplatData); active_sh_degree: the degree evaluated now - the
        reference starts a point-cloud initialisation at 0 (splat_data.cpp:211) and raises it every sh_degree_interval iterations
        (increment_sh_degree, :387-391); loaded / synthetic models are born at the maximum (the default here)."""
        self.means, self.sh0, self.shN = means, sh0, shN
        self.raw_scales, self.raw_quats, self.raw_opacities = raw_scales, raw_quats, raw_opacities
        self.max_sh_degree = sh_degree
        self.active_sh_degree = sh_degree if active_sh_degree is None else min(active_sh_degree, sh_degree)

    def parameters(self):
        # param-group order of strategy_utils.cpp:35-40: means, sh0, shN, scaling, rotation, opacity
        return [self.means, self.sh0, self.shN, self.raw_scales, self.raw_quats, self.raw_opacities]

    def get_means(self): return self.means
    def get_opacity(self): return torch.sigmoid(self.raw_opacities)
    def get_scaling(self): return torch.exp(self.raw_scales)
    def get_rotation(self): return torch.nn.functional.normalize(self.raw_quats, dim=-1)
    def get_shs(self): return torch.cat([self.sh0, self.shN], dim=1)
    def get_active_sh_degree(self): return self.active_sh_degree


@dataclass
class RenderOutput:
    image: Optional[torch.Tensor]
    alpha: torch.Tensor
    depth: Optional[torch.Tensor]
    means2d: torch.Tensor
    depths: torch.Tensor
    radii: torch.Tensor
    visibility: torch.Tensor
    width: int
    height: int
    n_isects: int = 0


class SphericalHarmonicsFunction(torch.autograd.Function):
    """rasterizer_autograd.cpp:15-133."""

    @staticmethod
    def forward(ctx, sh_degree: int, dirs, coeffs, masks):
        dirs, coeffs = dirs.contiguous(), coeffs.contiguous()
        if masks is None:
            masks = torch.ones(dirs.shape[:-1], dtype=torch.bool, device=dirs.device)
        masks = masks.contiguous()
        K = coeffs.shape[-2]
        colors = ops.spherical_harmonics_fwd(sh_degree, dirs.reshape(-1, 3), coeffs.reshape(-1, K, 3), masks.reshape(-1))
        ctx.save_for_backward(dirs, coeffs, masks)
        ctx.sh_degree, ctx.num_bases = sh_degree, K
        return colors.reshape(dirs.shape)

    @staticmethod
    def backward(ctx, v_colors):
        dirs, coeffs, masks = ctx.saved_tensors
        K = ctx.num_bases
        compute_v_dirs = ctx.needs_input_grad[1]
        v_coeffs, v_dirs = ops.spherical_harmonics_bwd(
            K, ctx.sh_degree, dirs.reshape(-1, 3), coeffs.reshape(-1, K, 3), masks.reshape(-1),
            v_colors.contiguous().reshape(-1, 3), compute_v_dirs)
        v_dirs = v_dirs.reshape(dirs.shape) if (v_dirs is not None and ctx.needs_input_grad[1]) else None
        v_coeffs = v_coeffs.reshape(coeffs.shape) if ctx.needs_input_grad[2] else None
        return None, v_dirs, v_coeffs, None


def spherical_harmonics(sh_degree: int, dirs, coeffs, masks=None):
    # the reference broadcasts coeffs [1,N,K,3] against dirs [C,N,3] (rasterizer.cpp:259-263)
    if coeffs.dim() == dirs.dim() + 1 and coeffs.shape[0] != dirs.shape[0]:
        coeffs = coeffs.expand(dirs.shape[0], *coeffs.shape[1:])
    return SphericalHarmonicsFunction.apply(sh_degree, dirs, coeffs, masks)


def fully_fused_projection_with_ut(means3D, quats, scales, opacities, viewmat, K, radial_coeffs, tangential_coeffs,
                                   thin_prism_coeffs, width, height, eps2d, near_plane, far_plane, radius_clip,
                                   scaling_modifier, camera_model, ut_params=None):
    """rasterizer_autograd.cpp:135-265 — not differentiable (3DGUT gets its gradients from the rasterizer).
    calc_compensations is hard-wired to False at the op, as in the reference (:231)."""
    with torch.no_grad():
        return ops.projection_ut_3dgs_fused(
            means3D.contiguous(), quats.contiguous(), (scales * scaling_modifier).contiguous(), opacities.contiguous(),
            viewmat.contiguous(), None, K.contiguous(), width, height, eps2d, near_plane, far_plane, radius_clip, False,
            camera_model, ut_params or UnscentedTransformParameters(), ShutterType.GLOBAL,
            radial_coeffs, tangential_coeffs, thin_prism_coeffs)


class GUTRasterizationFunction(torch.autograd.Function):
    """rasterizer_autograd.cpp:267-398."""

    @staticmethod
    def forward(ctx, means3D, quats, scales, colors, opacities, bg_color, masks, viewmat, K, radial_coeffs, tangential_coeffs,
                thin_prism_coeffs, isect_offsets, flatten_ids, width, height, tile_size, scaling_modifier, camera_model, ut_params):
        scales = scales * scaling_modifier
        args = (means3D.contiguous(), quats.contiguous(), scales.contiguous(), colors.contiguous(), opacities.contiguous(),
                None if bg_color is None else bg_color.contiguous(), None if masks is None else masks.contiguous(),
                width, height, tile_size, viewmat.contiguous(), None, K.contiguous(), camera_model, ut_params, ShutterType.GLOBAL,
                radial_coeffs, tangential_coeffs, thin_prism_coeffs, isect_offsets.contiguous(), flatten_ids.contiguous())
        # the forward's workspace (records + per-cell lists) is kept for the backward (own tensor, not the shared scratch)
        render_colors, render_alpha, last_ids, ctx.raster_ws = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args, own_workspace=True)
        ctx.fwd_args = args
        ctx.save_for_backward(render_alpha, last_ids)
        ctx.bg_needs_grad = bg_color is not None and bg_color.requires_grad
        return render_colors, render_alpha

    @staticmethod
    def backward(ctx, v_render_colors, v_render_alpha):
        render_alpha, last_ids = ctx.saved_tensors
        v_render_colors = v_render_colors.contiguous()
        v_render_alpha = v_render_alpha.contiguous()
        v_means, v_quats, v_scales, v_colors, v_opac = ops.rasterize_to_pixels_from_world_3dgs_bwd(
            *ctx.fwd_args, render_alpha, last_ids, v_render_colors, v_render_alpha, prepared_workspace=ctx.raster_ws)
        ctx.raster_ws = None
        v_bg = None
        if ctx.bg_needs_grad:
            v_bg = (v_render_colors * (1.0 - render_alpha)).float().sum(dim=(-3, -2))
        # (like the reference, d(scales*modifier)/d(scales) is dropped: exact for modifier == 1)
        return (v_means, v_quats, v_scales, v_colors, v_opac, v_bg) + (None,) * 14


def rasterize(camera: Camera, model: SplatModel, bg_color: Optional[torch.Tensor], scaling_modifier: float = 1.0,
              packed: bool = False, antialiased: bool = False, render_mode: RenderMode = RenderMode.RGB) -> RenderOutput:
    """rasterizer.cpp:46-430 (bounding-box filtering is a viewer feature and is not mirrored)."""
    if packed:
        raise ValueError("Packed mode is not supported in this implementation")
    image_height, image_width = int(camera.image_height), int(camera.image_width)
    viewmat, K = camera.world_view_transform, camera.K
    assert viewmat.dim() == 3 and viewmat.shape[0] == 1, "viewmat must be [1,4,4]"

    means3D = model.get_means()
    opacities = model.get_opacity()
    if opacities.dim() == 2 and opacities.shape[1] == 1:
        opacities = opacities.squeeze(-1)
    scales, rotations, sh_coeffs = model.get_scaling(), model.get_rotation(), model.get_shs()
    sh_degree = model.get_active_sh_degree()
    N = means3D.shape[0]
    if sh_coeffs.shape[1] < (sh_degree + 1) ** 2:
        raise ValueError("Not enough SH coefficients")

    prepared_bg = None
    if bg_color is not None and bg_color.numel() > 0:
        prepared_bg = bg_color.view(1, -1)

    eps2d, near_plane, far_plane, radius_clip, tile_size = 0.3, 0.01, 10000.0, 0.0, 16  # rasterizer.cpp:176-181
    radial = camera.radial_distortion
    if radial is not None and radial.numel() > 0:
        if radial.shape[-1] < 4:
            radial = torch.nn.functional.pad(radial, (0, 4 - radial.shape[-1]))
        radial = radial.reshape(1, -1).contiguous()
    else:
        radial = None
    tangential = camera.tangential_distortion
    if tangential is not None and tangential.numel() > 0:
        if tangential.shape[-1] < 2:
            tangential = torch.nn.functional.pad(tangential, (0, 2 - tangential.shape[-1]))
        tangential = tangential.reshape(1, -1).contiguous()
    else:
        tangential = None

    # Step 1: projection
    ut = UnscentedTransformParameters()
    radii, means2d, depths, conics, compensations = fully_fused_projection_with_ut(
        means3D, rotations, scales, opacities, viewmat, K, radial, tangential, None, image_width, image_height,
        eps2d, near_plane, far_plane, radius_clip, scaling_modifier, camera.camera_model_type, ut)

    # Step 2: colours from SH
    campos = torch.inverse(viewmat)[:, :3, 3]           # [C,3]
    dirs = means3D.unsqueeze(0) - campos.unsqueeze(1)   # [C,N,3]
    masks = (radii > 0).all(-1)                         # [C,N]
    colors = spherical_harmonics(sh_degree, dirs, sh_coeffs.unsqueeze(0), masks)
    colors = torch.clamp_min(colors + 0.5, 0.0)

    # Step 3: render mode
    if render_mode == RenderMode.RGB:
        render_colors, final_bg = colors, prepared_bg
    elif render_mode in (RenderMode.D, RenderMode.ED):
        render_colors = depths.unsqueeze(-1)
        final_bg = torch.zeros((1, 1), device=depths.device) if prepared_bg is not None else None
    else:
        render_colors = torch.cat([colors, depths.unsqueeze(-1)], -1)
        final_bg = torch.cat([prepared_bg, torch.zeros((1, 1), device=depths.device)], -1) if prepared_bg is not None else None

    # Step 4: opacities (compensations only when antialiased; the op never computes them, as in the reference)
    final_opacities = opacities.unsqueeze(0)
    if antialiased and compensations is not None and compensations.numel() > 0:
        final_opacities = final_opacities * compensations

    # Step 5: tile intersection (the sorted path returns the offsets too; intersect_offset stays available)
    tile_width = (image_width + tile_size - 1) // tile_size
    tile_height = (image_height + tile_size - 1) // tile_size
    with torch.no_grad():
        tiles_per_gauss, isect_ids, flatten_ids, isect_offsets = ops.intersect_tile(
            means2d.contiguous(), radii, depths, None, None, 1, tile_size, tile_width, tile_height, True, return_offsets=True)

    # Step 6: rasterization
    rendered_image, rendered_alpha = GUTRasterizationFunction.apply(
        means3D, rotations, scales, render_colors, final_opacities, final_bg, None, viewmat, K, radial, tangential, None,
        isect_offsets, flatten_ids, image_width, image_height, tile_size, scaling_modifier, camera.camera_model_type, ut)

    # Step 7: post-process
    final_image = final_depth = None
    if render_mode == RenderMode.RGB:
        final_image = rendered_image
    elif render_mode == RenderMode.D:
        final_depth = rendered_image
    elif render_mode == RenderMode.ED:
        final_depth = rendered_image / rendered_alpha.clamp_min(1e-10)
    elif render_mode == RenderMode.RGB_D:
        final_image, final_depth = rendered_image[..., :-1], rendered_image[..., -1:]
    else:
        final_image = rendered_image[..., :-1]
        final_depth = rendered_image[..., -1:] / rendered_alpha.clamp_min(1e-10)

    image = torch.clamp(final_image.squeeze(0).permute(2, 0, 1), 0.0, 1.0) if final_image is not None else None
    depth = final_depth.squeeze(0).permute(2, 0, 1) if final_depth is not None else None
    rad = radii.squeeze(0).max(-1).values
    return RenderOutput(image=image, alpha=rendered_alpha.squeeze(0).permute(2, 0, 1), depth=depth, means2d=means2d,
                        depths=depths.squeeze(0), radii=rad, visibility=rad > 0, width=image_width, height=image_height,
                        n_isects=int(flatten_ids.shape[0]))
