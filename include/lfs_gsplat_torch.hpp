// libtorch-facing declarations of the drop-in backend: the SAME symbols (namespaces, names,
// parameter types and order, return types) that /root/reference/gsplat/Ops.h:12-166 and
// /root/reference/fastgs/optimizer/include/{adam_api.h:11-21, adam.h:9-20} declare, implemented
// in lichtfeld-studio_amd/csrc/torch_ops.cpp on top of the C ABI (lfs_gsplat.h).
// A reference build that links this library instead of gsplat_backend / fastgs_backend keeps
// including its own Ops.h — the declarations are link-compatible; this header exists so the
// wrappers can be compiled and tested without the reference tree.
#pragma once
#include <ATen/core/Tensor.h>
#include <c10/util/Optional.h>
#include <torch/torch.h>
#include <array>
#include <tuple>

// ---- vocabulary types (gsplat/Common.h:46-50, gsplat/Cameras.h:16-61) -------------------------
namespace gsplat {
enum CameraModelType { PINHOLE = 0, ORTHO = 1, FISHEYE = 2 };
}
enum class ShutterType { ROLLING_TOP_TO_BOTTOM, ROLLING_LEFT_TO_RIGHT, ROLLING_BOTTOM_TO_TOP, ROLLING_RIGHT_TO_LEFT, GLOBAL };

struct UnscentedTransformParameters {
    float alpha = 0.1f, beta = 2.f, kappa = 0.f;
    float in_image_margin_factor = 0.1f;
    bool require_all_sigma_points_valid = true;
    torch::Tensor to_tensor() const;
    static UnscentedTransformParameters from_tensor(const torch::Tensor& t);
};

namespace gsplat {
using OptT = at::optional<at::Tensor>;

at::Tensor spherical_harmonics_fwd(const uint32_t degrees_to_use, const at::Tensor dirs, const at::Tensor coeffs, const OptT masks);
std::tuple<at::Tensor, at::Tensor> spherical_harmonics_bwd(const uint32_t K, const uint32_t degrees_to_use, const at::Tensor dirs,
                                                           const at::Tensor coeffs, const OptT masks, const at::Tensor v_colors, bool compute_v_dirs);

std::tuple<at::Tensor, at::Tensor, at::Tensor> intersect_tile(const at::Tensor means2d, const at::Tensor radii, const at::Tensor depths,
                                                              const OptT camera_ids, const OptT gaussian_ids, const uint32_t C,
                                                              const uint32_t tile_size, const uint32_t tile_width, const uint32_t tile_height,
                                                              const bool sort);
at::Tensor intersect_offset(const at::Tensor isect_ids, const uint32_t C, const uint32_t tile_width, const uint32_t tile_height);

at::Tensor quats_to_rotmats(const at::Tensor quats);
std::tuple<at::Tensor, at::Tensor> relocation(at::Tensor opacities, at::Tensor scales, at::Tensor ratios, at::Tensor binoms, const int n_max);
void add_noise(at::Tensor raw_opacities, at::Tensor raw_scales, at::Tensor raw_quats, at::Tensor noise, at::Tensor means, const float current_lr);

std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> projection_ut_3dgs_fused(
    const at::Tensor means, const at::Tensor quats, const at::Tensor scales, const OptT opacities, const at::Tensor viewmats0,
    const OptT viewmats1, const at::Tensor Ks, const uint32_t image_width, const uint32_t image_height, const float eps2d,
    const float near_plane, const float far_plane, const float radius_clip, const bool calc_compensations,
    const CameraModelType camera_model, const UnscentedTransformParameters ut_params, ShutterType rs_type,
    const OptT radial_coeffs, const OptT tangential_coeffs, const OptT thin_prism_coeffs);

std::tuple<at::Tensor, at::Tensor, at::Tensor> rasterize_to_pixels_from_world_3dgs_fwd(
    const at::Tensor means, const at::Tensor quats, const at::Tensor scales, const at::Tensor colors, const at::Tensor opacities,
    const OptT backgrounds, const OptT masks, const uint32_t image_width, const uint32_t image_height, const uint32_t tile_size,
    const at::Tensor viewmats0, const OptT viewmats1, const at::Tensor Ks, const CameraModelType camera_model,
    const UnscentedTransformParameters ut_params, ShutterType rs_type, const OptT radial_coeffs, const OptT tangential_coeffs,
    const OptT thin_prism_coeffs, const at::Tensor tile_offsets, const at::Tensor flatten_ids);

std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> rasterize_to_pixels_from_world_3dgs_bwd(
    const at::Tensor means, const at::Tensor quats, const at::Tensor scales, const at::Tensor colors, const at::Tensor opacities,
    const OptT backgrounds, const OptT masks, const uint32_t image_width, const uint32_t image_height, const uint32_t tile_size,
    const at::Tensor viewmats0, const OptT viewmats1, const at::Tensor Ks, const CameraModelType camera_model,
    const UnscentedTransformParameters ut_params, ShutterType rs_type, const OptT radial_coeffs, const OptT tangential_coeffs,
    const OptT thin_prism_coeffs, const at::Tensor tile_offsets, const at::Tensor flatten_ids, const at::Tensor render_alphas,
    const at::Tensor last_ids, const at::Tensor v_render_colors, const at::Tensor v_render_alphas);
} // namespace gsplat

namespace fast_gs::optimizer {
void adam_step(float* param, float* exp_avg, float* exp_avg_sq, const float* param_grad, const int n_elements, const float lr,
               const float beta1, const float beta2, const float eps, const float bias_correction1_rcp, const float bias_correction2_sqrt_rcp);
void adam_step_wrapper(torch::Tensor& param, torch::Tensor& exp_avg, torch::Tensor& exp_avg_sq, const torch::Tensor& param_grad,
                       const float lr, const float beta1, const float beta2, const float eps, const float bias_correction1_rcp,
                       const float bias_correction2_sqrt_rcp);
} // namespace fast_gs::optimizer

// The forward -> backward staging slot of the rasterizer wrappers (csrc/torch_ops.cpp; Ops.h has no argument through which a forward could hand its records and
// per-cell lists to its backward, so the last forward's workspace is parked, keyed on the identity of its operands). Control surface (extension, not in Ops.h):
//   torch_raster_staging_clear()     drop the parked workspace NOW - a viewer at the end of a render, a module before it is unloaded;
//   torch_keep_raster_staging(true)  park after EVERY forward (default: only when one of means / quats / scales / colors / opacities requires_grad(), i.e. a backward can follow);
//   torch_raster_staging_stats()     counters for tests and tuning.
namespace lfs {
struct RasterStagingStats { uint64_t stores, hits, misses, skipped_forwards, parked_bytes; };
void torch_raster_staging_clear();
void torch_keep_raster_staging(bool always);
RasterStagingStats torch_raster_staging_stats();
} // namespace lfs

#include "lfs_gut_train_step.hpp" // lfs::GutTrainStep, lfs::AdamGroupState: the --gut training step as ONE call (extension; does not depend on the declarations above)

// ---- fastgs/rasterization/include/rasterization_api.h:27-75, include/kernels/ssim.cuh:11-30 ----
namespace fast_gs::rasterization {
// (image, alpha, per_primitive_buffers, per_tile_buffers, per_instance_buffers, per_bucket_buffers,
//  n_visible_primitives, n_instances, n_buckets, primitive_primitive_indices_selector, instance_primitive_indices_selector).
// The state of this implementation lives in per_primitive_buffers (primitive workspace) and per_instance_buffers (instance
// workspace); the other two buffers are empty, n_visible_primitives / n_buckets / the selectors are 0.
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, int, int, int, int, int> forward_wrapper(
    const torch::Tensor& means, const torch::Tensor& scales_raw, const torch::Tensor& rotations_raw, const torch::Tensor& opacities_raw,
    const torch::Tensor& sh_coefficients_0, const torch::Tensor& sh_coefficients_rest, const torch::Tensor& w2c, const torch::Tensor& cam_position,
    const int active_sh_bases, const int width, const int height, const float focal_x, const float focal_y, const float center_x,
    const float center_y, const float near_plane, const float far_plane);
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> backward_wrapper(
    torch::Tensor& densification_info, const torch::Tensor& grad_image, const torch::Tensor& grad_alpha, const torch::Tensor& image,
    const torch::Tensor& alpha, const torch::Tensor& means, const torch::Tensor& scales_raw, const torch::Tensor& rotations_raw,
    const torch::Tensor& sh_coefficients_rest, const torch::Tensor& per_primitive_buffers, const torch::Tensor& per_tile_buffers,
    const torch::Tensor& per_instance_buffers, const torch::Tensor& per_bucket_buffers, const torch::Tensor& w2c, const torch::Tensor& cam_position,
    const int active_sh_bases, const int width, const int height, const float focal_x, const float focal_y, const float center_x,
    const float center_y, const float near_plane, const float far_plane, const int n_visible_primitives, const int n_instances,
    const int n_buckets, const int primitive_primitive_indices_selector, const int instance_primitive_indices_selector);
} // namespace fast_gs::rasterization

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> fusedssim(float C1, float C2, torch::Tensor& img1, torch::Tensor& img2, bool train);
torch::Tensor fusedssim_backward(float C1, float C2, torch::Tensor& img1, torch::Tensor& img2, torch::Tensor& dL_dmap, torch::Tensor& dm_dmu1,
                                 torch::Tensor& dm_dsigma1_sq, torch::Tensor& dm_dsigma12);

// ---- include/kernels/bilateral_grid.cuh:12-33 ----
namespace gs::bilateral_grid {
void slice_forward_cuda(const torch::Tensor& grid, const torch::Tensor& rgb, torch::Tensor& output, bool use_uniform_coords = true);
std::tuple<torch::Tensor, torch::Tensor> slice_backward_cuda(const torch::Tensor& grid, const torch::Tensor& rgb, const torch::Tensor& grad_output);
torch::Tensor tv_loss_forward_cuda(const torch::Tensor& grids);
torch::Tensor tv_loss_backward_cuda(const torch::Tensor& grids, const torch::Tensor& grad_output);
} // namespace gs::bilateral_grid

// ---- the loaders behind libtorch types (src/loader/formats/colmap.hpp, transforms.hpp; CameraData of include/loader/...). ----
// Thin adapters over liblfs_io.so (include/lfs_io.h): failures throw std::runtime_error with the reference's messages.
#include <filesystem>
#include <string>
#include <vector>
namespace gs::loader {
struct CameraData {   // the fields read_colmap_cameras fills (colmap.cpp:645-830)
    uint32_t _camera_ID = 0;
    int _camera_model = 0;                 // COLMAP model id
    int _camera_model_type = 0;            // gsplat::CameraModelType
    uint64_t _width = 0, _height = 0;
    float _focal_x = 0, _focal_y = 0, _center_x = 0, _center_y = 0;
    torch::Tensor _R, _T, _radial_distortion, _tangential_distortion, _params;
    std::string _image_name;
    std::filesystem::path _image_path;
};
struct PointCloud { torch::Tensor means, colors; };   // float32 [N,3], uint8 [N,3] (core/point_cloud.hpp)
std::tuple<std::vector<CameraData>, torch::Tensor> read_colmap_cameras_and_images(const std::filesystem::path& base, const std::string& images_folder = "images");
std::tuple<std::vector<CameraData>, torch::Tensor> read_colmap_cameras_and_images_text(const std::filesystem::path& base, const std::string& images_folder = "images");
std::tuple<std::vector<CameraData>, torch::Tensor> read_transforms_cameras_and_images(const std::filesystem::path& trans_path);
PointCloud read_colmap_point_cloud(const std::filesystem::path& base);
PointCloud read_colmap_point_cloud_text(const std::filesystem::path& base);
// SplatData::save_ply's file (src/core/splat_data.cpp:113-169, 402-505): tensors of any device; sh0 [N,1,3] and shN [N,K,3] are written transposed
// ([N,3,K] flattened) as there. load_ply (src/loader/formats/ply.cpp:186-640) returns the same six tensors on the CPU, with the reference's defaults for columns
// the file lacks (:531-600: sh0 zeros [N,1,3], shN zeros [N,15,3], opacity 0, log-scale -5, identity quaternion); the file written by save_ply is byte-identical
// to the reference writer's, and load_ply equals the reference's reader (tests/test_loader_reference.py).
struct SplatTensors { torch::Tensor means, sh0, shN, scaling, rotation, opacity; };
void save_ply(const std::filesystem::path& path, const torch::Tensor& means, const torch::Tensor& sh0, const torch::Tensor& shN, const torch::Tensor& scaling,
              const torch::Tensor& rotation, const torch::Tensor& opacity);
SplatTensors load_ply(const std::filesystem::path& path);
} // namespace gs::loader
