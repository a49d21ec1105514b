"""ctypes binding of the C ABI declared in include/lfs_gsplat.h (liblfs_gsplat.so).

Everything here works on raw device pointers (`tensor.data_ptr()`) and the current HIP
stream of torch; torch is used only as the allocator / stream owner.  Loading fails loudly
when the library has not been built (`python lichtfeld-studio_amd/build.py`): there is no
CPU path behind this module.
"""
from __future__ import annotations

import ctypes as C
import enum
import os
from dataclasses import dataclass

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class CameraModelType(enum.IntEnum):  # gsplat/Common.h:46-50
    PINHOLE = 0
    ORTHO = 1
    FISHEYE = 2


class ShutterType(enum.IntEnum):  # gsplat/Cameras.h:16-22
    ROLLING_TOP_TO_BOTTOM = 0
    ROLLING_LEFT_TO_RIGHT = 1
    ROLLING_BOTTOM_TO_TOP = 2
    ROLLING_RIGHT_TO_LEFT = 3
    GLOBAL = 4


@dataclass
class UnscentedTransformParameters:  # gsplat/Cameras.h:27-61
    alpha: float = 0.1
    beta: float = 2.0
    kappa: float = 0.0
    in_image_margin_factor: float = 0.1
    require_all_sigma_points_valid: bool = True

    def to_tensor(self) -> torch.Tensor:
        return torch.tensor([self.alpha, self.beta, self.kappa, self.in_image_margin_factor,
                             float(self.require_all_sigma_points_valid)], dtype=torch.float32)

    @staticmethod
    def from_tensor(t: torch.Tensor) -> "UnscentedTransformParameters":
        if t.dim() != 1 or t.shape[0] != 5:
            raise ValueError("UnscentedTransformParameters must be a 1D tensor of size 5")
        v = t.tolist()
        return UnscentedTransformParameters(v[0], v[1], v[2], v[3], bool(v[4]))


class _UT(C.Structure):
    _fields_ = [("alpha", C.c_float), ("beta", C.c_float), ("kappa", C.c_float),
                ("in_image_margin_factor", C.c_float), ("require_all_sigma_points_valid", C.c_int32)]


class _Cameras(C.Structure):
    _fields_ = [("C", C.c_uint32), ("image_width", C.c_uint32), ("image_height", C.c_uint32),
                ("camera_model", C.c_int32), ("rs_type", C.c_int32),
                ("viewmats0", C.c_void_p), ("viewmats1", C.c_void_p), ("Ks", C.c_void_p),
                ("radial_coeffs", C.c_void_p), ("n_radial", C.c_int32),
                ("tangential_coeffs", C.c_void_p), ("thin_prism_coeffs", C.c_void_p), ("n_thin_prism", C.c_int32)]


class AdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("grad", C.c_void_p),
                ("n_elements", C.c_int64), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float), ("bias_correction1_rcp", C.c_float), ("bias_correction2_sqrt_rcp", C.c_float)]


class ParamRows(C.Structure):  # lfs_param_rows
    _fields_ = [("param", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("width", C.c_uint32)]


ADAM_MAX_TENSORS = 8

# every symbol include/lfs_gsplat.h declares
EXPORTS = [
    "lfs_projection_ut_3dgs_fused", "lfs_spherical_harmonics_fwd", "lfs_spherical_harmonics_bwd",
    "lfs_intersect_tile_workspace_bytes", "lfs_intersect_tile_count", "lfs_intersect_tile_emit", "lfs_intersect_tile_count_ex", "lfs_intersect_tile_emit_ex", "lfs_get_debug_flags", "lfs_intersect_offset",
    "lfs_rasterize_workspace_bytes", "lfs_set_debug_flags", "lfs_rasterize_to_pixels_from_world_3dgs_fwd", "lfs_rasterize_to_pixels_from_world_3dgs_bwd", "lfs_rasterize_to_pixels_from_world_3dgs_bwd_prepared", "lfs_rasterize_to_pixels_from_world_3dgs_bwd_prepared_mse",
    "lfs_sh_model_fwd", "lfs_sh_model_bwd", "lfs_sh_model_bwd_adam", "lfs_sh_model_fwd_views", "lfs_sh_model_bwd_views", "lfs_activations_fwd", "lfs_activations_bwd", "lfs_mse_loss_fwd_bwd", "lfs_mse_loss_chw_fwd_bwd",
    "lfs_fastgs_primitive_workspace_bytes", "lfs_fastgs_instance_workspace_bytes", "lfs_fastgs_preprocess", "lfs_fastgs_wait_n_instances", "lfs_fastgs_render", "lfs_fastgs_backward", "lfs_fastgs_backward_adam",
    "lfs_fastgs_set_debug_flags", "lfs_fused_ssim_fwd", "lfs_fused_ssim_bwd", "lfs_photometric_loss_workspace_bytes", "lfs_photometric_loss_fwd_bwd", "lfs_photometric_loss_chw_fwd_bwd", "lfs_photometric_loss_ex_fwd_bwd", "lfs_mse_loss_ex_fwd_bwd",
    "lfs_bilateral_slice_fwd", "lfs_bilateral_slice_bwd", "lfs_bilateral_tv_loss_fwd", "lfs_bilateral_tv_loss_bwd", "lfs_image_u8_to_chw_f32", "lfs_mean_neighbor_distances", "lfs_mean_neighbor_distances_exact",
    "lfs_quats_to_rotmats", "lfs_relocation", "lfs_add_noise", "lfs_mcmc_relocate_workspace_bytes", "lfs_mcmc_relocate",
    "lfs_activations_project_ut", "lfs_gut_step_layout_for", "lfs_gut_step_fits", "lfs_gut_train_step", "lfs_gut_view_forward", "lfs_gut_view_backward", "lfs_gut_view_backward_sh", "lfs_gut_view_backward_finish", "lfs_gut_view_backward_rows", "lfs_gut_step_wait", "lfs_gut_step_supported", "lfs_gut_train_step_pipelined", "lfs_gut_pipeline_join", "lfs_gut_train_step_ex",
    "lfs_rasterize_workspace_acc_offset", "lfs_rasterize_to_pixels_from_world_3dgs_bwd_prepared_mse_acc", "lfs_sh_model_bwd_adam_all", "lfs_gut_finish_adam", "lfs_gut_finish_grads", "lfs_rasterize_to_pixels_from_world_3dgs_bwd_prepared_acc", "lfs_adam_step", "lfs_adam_step_multi", "lfs_version", "lfs_profile_enable", "lfs_profile_filter", "lfs_profile_collect",
]


def library_path() -> str:
    # LFS_GSPLAT_LIB: developer A/B of a differently built variant of the SAME library (tools/build_variant.sh); never a fallback
    return os.environ.get("LFS_GSPLAT_LIB") or os.path.join(_HERE, "liblfs_gsplat.so")


def load_library():
    """Load liblfs_gsplat.so (once). Raises if it was not built: no fallback exists."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build the HIP library first (python lichtfeld-studio_amd/build.py). "
                "This package has no CPU or PyTorch fallback.")
        lib = C.CDLL(path)
        for name in EXPORTS:
            if not hasattr(lib, name):
                raise RuntimeError(f"{path} does not export {name}")
        lib.lfs_intersect_tile_workspace_bytes.restype = C.c_size_t
        lib.lfs_rasterize_workspace_bytes.restype = C.c_size_t
        lib.lfs_photometric_loss_workspace_bytes.restype = C.c_size_t
        lib.lfs_fastgs_primitive_workspace_bytes.restype = C.c_size_t
        lib.lfs_fastgs_instance_workspace_bytes.restype = C.c_size_t
        lib.lfs_mcmc_relocate_workspace_bytes.restype = C.c_size_t
        lib.lfs_rasterize_workspace_acc_offset.restype = C.c_size_t
        lib.lfs_version.restype = C.c_char_p
        if os.environ.get("LFS_DEBUG_FLAGS"):   # measurement / debugging only (tools/*.sh): e.g. 64 = the reference's tile lists + pack + cull kernels inside the step
            lib.lfs_set_debug_flags(C.c_uint32(int(os.environ["LFS_DEBUG_FLAGS"], 0)))
        _LIB = lib
    return _LIB


class LfsError(RuntimeError):
    pass


_ERR = {-1: "LFS_E_INVALID (bad argument)", -2: "LFS_E_UNSUPPORTED", -3: "LFS_E_WORKSPACE (workspace too small)"}


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise LfsError(f"{what} failed: {_ERR.get(rc, f'hipError {rc}')}")


def ptr(t: torch.Tensor | None):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(*tensors: torch.Tensor | None) -> None:
    """CHECK_INPUT of gsplat/Common.h:12-17 — device tensor + contiguous."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise LfsError("tensor must be a CUDA (HIP) tensor")
        if not t.is_contiguous():
            raise LfsError("tensor must be contiguous")


def ut_struct(ut: UnscentedTransformParameters | None) -> _UT:
    ut = ut or UnscentedTransformParameters()
    return _UT(ut.alpha, ut.beta, ut.kappa, ut.in_image_margin_factor, int(ut.require_all_sigma_points_valid))


def cameras_struct(viewmats0, viewmats1, Ks, width, height, camera_model, rs_type,
                   radial_coeffs=None, tangential_coeffs=None, thin_prism_coeffs=None) -> _Cameras:
    """Keeps no reference to the tensors: callers hold them for the duration of the call."""
    cam = _Cameras()
    cam.C = int(Ks.shape[0])
    cam.image_width, cam.image_height = int(width), int(height)
    cam.camera_model, cam.rs_type = int(camera_model), int(rs_type)
    cam.viewmats0 = viewmats0.data_ptr()
    cam.viewmats1 = viewmats1.data_ptr() if viewmats1 is not None else None
    cam.Ks = Ks.data_ptr()
    cam.radial_coeffs = radial_coeffs.data_ptr() if radial_coeffs is not None else None
    cam.n_radial = int(radial_coeffs.shape[-1]) if radial_coeffs is not None else 0
    cam.tangential_coeffs = tangential_coeffs.data_ptr() if tangential_coeffs is not None else None
    cam.thin_prism_coeffs = thin_prism_coeffs.data_ptr() if thin_prism_coeffs is not None else None
    cam.n_thin_prism = int(thin_prism_coeffs.shape[-1]) if thin_prism_coeffs is not None else 0
    return cam


# ---- workspace cache: one growing byte buffer per (device, tag) ----------------
_WS: dict[tuple[int, str], torch.Tensor] = {}


def workspace(nbytes: int, device: torch.device, tag: str) -> torch.Tensor:
    key = (device.index if device.index is not None else torch.cuda.current_device(), tag)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


def profile_enable(on: bool) -> None:
    load_library().lfs_profile_enable(C.c_int(int(on)))


def profile_filter(name: str | None) -> None:
    """Time only the scopes called `name` (None = all)."""
    load_library().lfs_profile_filter(C.c_char_p(name.encode()) if name else None)


def profile_collect(max_entries: int = 64) -> dict:
    """{kernel name: (total_ms, launches)} since the last collect; waits for the recorded events."""
    names = C.create_string_buffer(64 * max_entries)
    ms = (C.c_float * max_entries)()
    counts = (C.c_int * max_entries)()
    n = load_library().lfs_profile_collect(C.c_int(max_entries), names, ms, counts)
    out = {}
    for i in range(n):
        out[names.raw[64 * i:64 * (i + 1)].split(b"\0", 1)[0].decode()] = (float(ms[i]), int(counts[i]))
    return out
