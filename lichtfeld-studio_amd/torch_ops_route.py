"""The DROP-IN route, for measurement and tests: the op-by-op training step of the reference's L2 code
(/root/reference/src/training/rasterization/rasterizer.cpp:224-344 under rasterizer_autograd.cpp, then FusedAdam::step's six adam_step_wrapper calls,
src/training/optimizers/fused_adam.cpp:22-95) executed through the COMPILED C++ wrappers of csrc/torch_ops.cpp - the functions a reference build links when
`gsplat_backend` / `fastgs_backend` are swapped for this library - instead of the ctypes mirror in ops.py.

`install()` points rasterizer.py (the mirror of rasterizer.cpp / rasterizer_autograd.cpp) and fused_adam.py at an adapter over the pybind module
`_lfs_torch_ops`; the adapter only converts argument conventions (enums -> int, UnscentedTransformParameters -> its 5-float tensor, the separate
intersect_offset call the reference makes at rasterizer.cpp:327). What the route cannot have, by the reference API's design: the forward's workspace handed to
the backward (Ops.h has no such argument: the backward wrapper packs and culls again), the activations / SH / loss fusion, Adam inside the backward kernels.
`bench.py --path ops` reports what that costs."""
from __future__ import annotations

import torch


class TorchOpsAdapter:
    def __init__(self):
        from . import _lfs_torch_ops as m   # built by build.build_torch_ops(); ImportError if absent - no fallback
        self.m = m

    @staticmethod
    def _ut(ut):
        return None if ut is None else ut.to_tensor()

    def spherical_harmonics_fwd(self, degrees_to_use, dirs, coeffs, masks):
        return self.m.spherical_harmonics_fwd(int(degrees_to_use), dirs, coeffs, masks)

    def spherical_harmonics_bwd(self, K, degrees_to_use, dirs, coeffs, masks, v_colors, compute_v_dirs):
        v_coeffs, v_dirs = self.m.spherical_harmonics_bwd(int(K), int(degrees_to_use), dirs, coeffs, masks, v_colors, bool(compute_v_dirs))
        return v_coeffs, (v_dirs if compute_v_dirs else None)

    def projection_ut_3dgs_fused(self, means, quats, scales, opacities, viewmats0, viewmats1, Ks, image_width, image_height, eps2d, near_plane, far_plane,
                                 radius_clip, calc_compensations, camera_model, ut_params=None, rs_type=4, radial_coeffs=None, tangential_coeffs=None,
                                 thin_prism_coeffs=None):
        return self.m.projection_ut_3dgs_fused(means, quats, scales, opacities, viewmats0, viewmats1, Ks, int(image_width), int(image_height), float(eps2d),
                                               float(near_plane), float(far_plane), float(radius_clip), bool(calc_compensations), int(camera_model), self._ut(ut_params),
                                               int(rs_type), radial_coeffs, tangential_coeffs, thin_prism_coeffs)

    def intersect_tile(self, means2d, radii, depths, camera_ids, gaussian_ids, C_, tile_size, tile_width, tile_height, sort, *, return_offsets=False, overlap=None):
        tpg, isect_ids, flatten_ids = self.m.intersect_tile(means2d, radii, depths, camera_ids, gaussian_ids, int(C_), int(tile_size), int(tile_width), int(tile_height), bool(sort))
        if not return_offsets:
            return tpg, isect_ids, flatten_ids
        return tpg, isect_ids, flatten_ids, self.m.intersect_offset(isect_ids, int(C_), int(tile_width), int(tile_height))   # rasterizer.cpp:327

    def rasterize_to_pixels_from_world_3dgs_fwd(self, means, quats, scales, colors, opacities, backgrounds, masks, image_width, image_height, tile_size, viewmats0,
                                                viewmats1, Ks, camera_model, ut_params, rs_type, radial_coeffs, tangential_coeffs, thin_prism_coeffs, tile_offsets,
                                                flatten_ids, own_workspace=False):
        out = self.m.rasterize_to_pixels_from_world_3dgs_fwd(means, quats, scales, colors, opacities, backgrounds, masks, int(image_width), int(image_height),
                                                             int(tile_size), viewmats0, viewmats1, Ks, int(camera_model), self._ut(ut_params), int(rs_type),
                                                             radial_coeffs, tangential_coeffs, thin_prism_coeffs, tile_offsets, flatten_ids)
        return (*out, None) if own_workspace else out

    def rasterize_to_pixels_from_world_3dgs_bwd(self, means, quats, scales, colors, opacities, backgrounds, masks, image_width, image_height, tile_size, viewmats0,
                                                viewmats1, Ks, camera_model, ut_params, rs_type, radial_coeffs, tangential_coeffs, thin_prism_coeffs, tile_offsets,
                                                flatten_ids, render_alphas, last_ids, v_render_colors, v_render_alphas, prepared_workspace=None, v_colors_out=None):
        return self.m.rasterize_to_pixels_from_world_3dgs_bwd(means, quats, scales, colors, opacities, backgrounds, masks, int(image_width), int(image_height),
                                                              int(tile_size), viewmats0, viewmats1, Ks, int(camera_model), self._ut(ut_params), int(rs_type),
                                                              radial_coeffs, tangential_coeffs, thin_prism_coeffs, tile_offsets, flatten_ids, render_alphas,
                                                              last_ids, v_render_colors, v_render_alphas)

    def adam_step_wrapper(self, param, exp_avg, exp_avg_sq, grad, lr, beta1, beta2, eps, bc1_rcp, bc2_sqrt_rcp):
        self.m.adam_step_wrapper(param.detach() if isinstance(param, torch.Tensor) else param, exp_avg, exp_avg_sq, grad, float(lr), float(beta1), float(beta2),
                                 float(eps), float(bc1_rcp), float(bc2_sqrt_rcp))


_SAVED = {}


def install() -> TorchOpsAdapter:
    """rasterizer.py and fused_adam.py call the compiled C++ wrappers from now on (uninstall() restores the ctypes mirror)."""
    from . import fused_adam, rasterizer
    ad = TorchOpsAdapter()
    if not _SAVED:
        _SAVED["rasterizer"], _SAVED["fused_adam"] = rasterizer.ops, fused_adam.ops
    rasterizer.ops = ad
    fused_adam.ops = ad
    return ad


def uninstall() -> None:
    from . import fused_adam, rasterizer
    if _SAVED:
        rasterizer.ops, fused_adam.ops = _SAVED.pop("rasterizer"), _SAVED.pop("fused_adam")
