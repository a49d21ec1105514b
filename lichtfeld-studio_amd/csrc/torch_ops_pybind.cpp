// Python module `_lfs_torch_ops`: exposes the libtorch wrappers of torch_ops.cpp 1:1 so that the
// C++ drop-in layer is exercised by the test-suite (tests/test_gpu_torch_ops.py).  Enums travel as
// ints, UnscentedTransformParameters as its 5-float tensor form (Cameras.h:42-61).
#include "../../include/lfs_gsplat_torch.hpp"
#include <torch/extension.h>

namespace py = pybind11;
using gsplat::OptT;

static UnscentedTransformParameters ut_of(const c10::optional<at::Tensor>& t) {
    return (t.has_value() && t->defined()) ? UnscentedTransformParameters::from_tensor(t->cpu()) : UnscentedTransformParameters{};
}

PYBIND11_MODULE(_lfs_torch_ops, m) {
    m.def("spherical_harmonics_fwd", &gsplat::spherical_harmonics_fwd);
    m.def("spherical_harmonics_bwd", &gsplat::spherical_harmonics_bwd);
    m.def("intersect_tile", &gsplat::intersect_tile);
    m.def("intersect_offset", &gsplat::intersect_offset);
    m.def("quats_to_rotmats", &gsplat::quats_to_rotmats);
    m.def("relocation", &gsplat::relocation);
    m.def("add_noise", &gsplat::add_noise);
    m.def("projection_ut_3dgs_fused",
          [](at::Tensor means, at::Tensor quats, at::Tensor scales, OptT opacities, at::Tensor viewmats0, OptT viewmats1, at::Tensor Ks,
             uint32_t W, uint32_t H, float eps2d, float near_plane, float far_plane, float radius_clip, bool calc_compensations, int camera_model,
             OptT ut, int rs_type, OptT radial, OptT tangential, OptT thin) {
              return gsplat::projection_ut_3dgs_fused(means, quats, scales, opacities, viewmats0, viewmats1, Ks, W, H, eps2d, near_plane, far_plane,
                                                      radius_clip, calc_compensations, (gsplat::CameraModelType)camera_model, ut_of(ut),
                                                      (ShutterType)rs_type, radial, tangential, thin);
          });
    m.def("rasterize_to_pixels_from_world_3dgs_fwd",
          [](at::Tensor means, at::Tensor quats, at::Tensor scales, at::Tensor colors, at::Tensor opacities, OptT backgrounds, OptT masks,
             uint32_t W, uint32_t H, uint32_t tile_size, at::Tensor viewmats0, OptT viewmats1, at::Tensor Ks, int camera_model, OptT ut, int rs_type,
             OptT radial, OptT tangential, OptT thin, at::Tensor tile_offsets, at::Tensor flatten_ids) {
              return gsplat::rasterize_to_pixels_from_world_3dgs_fwd(means, quats, scales, colors, opacities, backgrounds, masks, W, H, tile_size,
                                                                     viewmats0, viewmats1, Ks, (gsplat::CameraModelType)camera_model, ut_of(ut),
                                                                     (ShutterType)rs_type, radial, tangential, thin, tile_offsets, flatten_ids);
          });
    m.def("rasterize_to_pixels_from_world_3dgs_bwd",
          [](at::Tensor means, at::Tensor quats, at::Tensor scales, at::Tensor colors, at::Tensor opacities, OptT backgrounds, OptT masks,
             uint32_t W, uint32_t H, uint32_t tile_size, at::Tensor viewmats0, OptT viewmats1, at::Tensor Ks, int camera_model, OptT ut, int rs_type,
             OptT radial, OptT tangential, OptT thin, at::Tensor tile_offsets, at::Tensor flatten_ids, at::Tensor render_alphas, at::Tensor last_ids,
             at::Tensor v_render_colors, at::Tensor v_render_alphas) {
              return gsplat::rasterize_to_pixels_from_world_3dgs_bwd(means, quats, scales, colors, opacities, backgrounds, masks, W, H, tile_size,
                                                                     viewmats0, viewmats1, Ks, (gsplat::CameraModelType)camera_model, ut_of(ut),
                                                                     (ShutterType)rs_type, radial, tangential, thin, tile_offsets, flatten_ids,
                                                                     render_alphas, last_ids, v_render_colors, v_render_alphas);
          });
    m.def("raster_staging_clear", &lfs::torch_raster_staging_clear);
    m.def("keep_raster_staging", &lfs::torch_keep_raster_staging);
    m.def("raster_staging_stats", []() { const auto s = lfs::torch_raster_staging_stats();
                                          py::dict d; d["stores"] = s.stores; d["hits"] = s.hits; d["misses"] = s.misses; d["skipped_forwards"] = s.skipped_forwards; d["parked_bytes"] = s.parked_bytes; return d; });
    m.def("adam_step_wrapper", [](at::Tensor param, at::Tensor exp_avg, at::Tensor exp_avg_sq, at::Tensor grad, float lr, float b1, float b2,
                                  float eps, float bc1, float bc2) { fast_gs::optimizer::adam_step_wrapper(param, exp_avg, exp_avg_sq, grad, lr, b1, b2, eps, bc1, bc2); });
    m.def("fastgs_forward_wrapper",
          [](at::Tensor means, at::Tensor scales_raw, at::Tensor rotations_raw, at::Tensor opacities_raw, at::Tensor sh0, at::Tensor sh_rest, at::Tensor w2c,
             at::Tensor cam_position, int active_sh_bases, int width, int height, float fx, float fy, float cx, float cy, float near_plane, float far_plane) {
              return fast_gs::rasterization::forward_wrapper(means, scales_raw, rotations_raw, opacities_raw, sh0, sh_rest, w2c, cam_position,
                                                             active_sh_bases, width, height, fx, fy, cx, cy, near_plane, far_plane);
          });
    m.def("fastgs_backward_wrapper",
          [](at::Tensor densification_info, at::Tensor grad_image, at::Tensor grad_alpha, at::Tensor image, at::Tensor alpha, at::Tensor means,
             at::Tensor scales_raw, at::Tensor rotations_raw, at::Tensor sh_rest, at::Tensor prim, at::Tensor tile, at::Tensor inst, at::Tensor bucket,
             at::Tensor w2c, at::Tensor cam_position, int active_sh_bases, int width, int height, float fx, float fy, float cx, float cy,
             float near_plane, float far_plane, int n_visible, int n_instances, int n_buckets, int sel0, int sel1) {
              return fast_gs::rasterization::backward_wrapper(densification_info, grad_image, grad_alpha, image, alpha, means, scales_raw, rotations_raw,
                                                              sh_rest, prim, tile, inst, bucket, w2c, cam_position, active_sh_bases, width, height, fx, fy,
                                                              cx, cy, near_plane, far_plane, n_visible, n_instances, n_buckets, sel0, sel1);
          });
    // lfs::GutTrainStep: params / exp_avg / exp_avg_sq as lists of six tensors (FusedAdam group order), scalars [6][6] = lr, beta1, beta2, eps, bc1_rcp, bc2_sqrt_rcp
    py::class_<lfs::GutTrainStep>(m, "GutTrainStep")
        .def(py::init<uint32_t, int64_t>(), py::arg("tile_size") = 16, py::arg("initial_capacity") = 0)
        .def("step", [](lfs::GutTrainStep& self, std::vector<at::Tensor> params, std::vector<at::Tensor> exp_avg, std::vector<at::Tensor> exp_avg_sq,
                        std::vector<std::vector<float>> scalars, uint32_t sh_degree, at::Tensor viewmat, at::Tensor K, uint32_t W, uint32_t H, OptT background,
                        at::Tensor target, float weight, at::Tensor loss, float scale_reg, float opacity_reg) {
            TORCH_CHECK(params.size() == 6 && exp_avg.size() == 6 && exp_avg_sq.size() == 6 && scalars.size() == 6, "six parameter groups");
            std::array<lfs::AdamGroupState, 6> ad;
            for (int k = 0; k < 6; ++k) {
                TORCH_CHECK(scalars[k].size() == 6, "six Adam scalars per group");
                ad[k] = lfs::AdamGroupState{exp_avg[k], exp_avg_sq[k], scalars[k][0], scalars[k][1], scalars[k][2], scalars[k][3], scalars[k][4], scalars[k][5]};
            }
            return self.step(params[0], params[1], params[2], params[3], params[4], params[5], ad, sh_degree, viewmat, K, W, H, background, target, weight, loss,
                             scale_reg, opacity_reg);
        })
        .def("render", &lfs::GutTrainStep::render).def("alpha", &lfs::GutTrainStep::alpha).def("radii", &lfs::GutTrainStep::radii)
        .def("retries", &lfs::GutTrainStep::retries).def("capacity", &lfs::GutTrainStep::capacity);
    m.def("fusedssim", [](float C1, float C2, at::Tensor a, at::Tensor b, bool train) { return fusedssim(C1, C2, a, b, train); });
    m.def("fusedssim_backward", [](float C1, float C2, at::Tensor a, at::Tensor b, at::Tensor g, at::Tensor d1, at::Tensor d2, at::Tensor d3) {
        return fusedssim_backward(C1, C2, a, b, g, d1, d2, d3);
    });
    m.def("bilateral_slice_forward", [](at::Tensor grid, at::Tensor rgb) { at::Tensor out = at::empty_like(rgb); gs::bilateral_grid::slice_forward_cuda(grid, rgb, out, true); return out; });
    m.def("bilateral_slice_backward", [](at::Tensor grid, at::Tensor rgb, at::Tensor go) { return gs::bilateral_grid::slice_backward_cuda(grid, rgb, go); });
    m.def("bilateral_tv_loss_forward", [](at::Tensor grids) { return gs::bilateral_grid::tv_loss_forward_cuda(grids); });
    m.def("bilateral_tv_loss_backward", [](at::Tensor grids, at::Tensor go) { return gs::bilateral_grid::tv_loss_backward_cuda(grids, go); });
    // gs::loader adapters (host only)
    auto cams_to_py = [](const std::tuple<std::vector<gs::loader::CameraData>, torch::Tensor>& r) {
        py::list cams;
        for (const auto& c : std::get<0>(r)) {
            py::dict d;
            d["camera_ID"] = c._camera_ID; d["camera_model"] = c._camera_model; d["camera_model_type"] = c._camera_model_type;
            d["width"] = c._width; d["height"] = c._height; d["focal_x"] = c._focal_x; d["focal_y"] = c._focal_y; d["center_x"] = c._center_x; d["center_y"] = c._center_y;
            d["R"] = c._R; d["T"] = c._T; d["radial_distortion"] = c._radial_distortion; d["tangential_distortion"] = c._tangential_distortion; d["params"] = c._params;
            d["image_name"] = c._image_name; d["image_path"] = c._image_path.string();
            cams.append(d);
        }
        return py::make_tuple(cams, std::get<1>(r));
    };
    m.def("read_colmap_cameras_and_images", [cams_to_py](const std::string& base, const std::string& folder) { return cams_to_py(gs::loader::read_colmap_cameras_and_images(base, folder)); });
    m.def("read_colmap_cameras_and_images_text", [cams_to_py](const std::string& base, const std::string& folder) { return cams_to_py(gs::loader::read_colmap_cameras_and_images_text(base, folder)); });
    m.def("read_transforms_cameras_and_images", [cams_to_py](const std::string& path) { return cams_to_py(gs::loader::read_transforms_cameras_and_images(path)); });
    m.def("read_colmap_point_cloud", [](const std::string& base) { auto pc = gs::loader::read_colmap_point_cloud(base); return py::make_tuple(pc.means, pc.colors); });
    m.def("read_colmap_point_cloud_text", [](const std::string& base) { auto pc = gs::loader::read_colmap_point_cloud_text(base); return py::make_tuple(pc.means, pc.colors); });
    m.def("save_ply", [](const std::string& path, at::Tensor means, at::Tensor sh0, at::Tensor shN, at::Tensor scaling, at::Tensor rotation, at::Tensor opacity) {
        gs::loader::save_ply(path, means, sh0, shN, scaling, rotation, opacity);
    });
    m.def("load_ply", [](const std::string& path) { auto s = gs::loader::load_ply(path); return py::make_tuple(s.means, s.sh0, s.shN, s.scaling, s.rotation, s.opacity); });
}
