// ORACLE/_ref - TEST INFRASTRUCTURE ONLY.
// The refk_* entry points of ref_kernels.cpp (same C signatures, so that oracle/make_golden_refk*.py runs unchanged), implemented NOT by restated launch sequences
// but by calling the reference's own operators gsplat::projection_ut_3dgs_fused, spherical_harmonics_fwd / _bwd, intersect_tile / _offset,
// rasterize_to_pixels_from_world_3dgs_fwd / _bwd, relocation, add_noise, quats_to_rotmats - gsplat/*.cpp over gsplat/*.cu, compiled in place (ref_gsplat_all.cpp;
// `make -C oracle refgsplat` -> oracle/_ref/libref_raster_full.so). tests/test_oracle_refk_golden.py regenerates the committed golden files through THESE entry
// points as well and compares bit for bit: the files hold what the reference's whole operator layer computes, and the restated launch sequences equal it.
#include "Ops.h"
#include <torch/torch.h>
#include <cstring>

#define REFK_API extern "C" __attribute__((visibility("default")))

namespace {
    at::Tensor f32(const float* p, std::vector<int64_t> shape) { return torch::from_blob(const_cast<float*>(p), shape, torch::kFloat32).clone(); }
    at::optional<at::Tensor> optf(const float* p, std::vector<int64_t> shape) { return p ? at::optional<at::Tensor>(f32(p, shape)) : at::nullopt; }
    at::Tensor i32(const int32_t* p, std::vector<int64_t> shape) { return torch::from_blob(const_cast<int32_t*>(p), shape, torch::kInt32).clone(); }
    template <class T> void put(const at::Tensor& t, T* dst) {
        if (!dst || !t.defined()) return;
        auto c = t.contiguous();
        std::memcpy(dst, c.data_ptr(), c.numel() * c.element_size());
    }
    UnscentedTransformParameters ut_of(const float* ut) {
        UnscentedTransformParameters p;
        if (ut) { p.alpha = ut[0]; p.beta = ut[1]; p.kappa = ut[2]; p.in_image_margin_factor = ut[3]; p.require_all_sigma_points_valid = ut[4] != 0.f; }
        return p;
    }
    int64_t n_radial(int camera_model) { return camera_model == (int)gsplat::CameraModelType::FISHEYE ? 4 : 6; }
} // namespace

REFK_API void refk_projection_ut(uint32_t C, uint32_t N, const float* means, const float* quats, const float* scales, const float* opacities, const float* viewmats0,
                                 const float* viewmats1, const float* Ks, uint32_t W, uint32_t H, float eps2d, float near_plane, float far_plane, float radius_clip,
                                 int camera_model, const float* ut, int rs_type, const float* radial, const float* tangential, const float* thin_prism, int32_t* radii,
                                 float* means2d, float* depths, float* conics, float* compensations) {
    auto r = gsplat::projection_ut_3dgs_fused(f32(means, {N, 3}), f32(quats, {N, 4}), f32(scales, {N, 3}), optf(opacities, {N}), f32(viewmats0, {C, 4, 4}),
                                              optf(viewmats1, {C, 4, 4}), f32(Ks, {C, 3, 3}), W, H, eps2d, near_plane, far_plane, radius_clip, compensations != nullptr,
                                              gsplat::CameraModelType(camera_model), ut_of(ut), ShutterType(rs_type), optf(radial, {C, n_radial(camera_model)}),
                                              optf(tangential, {C, 2}), optf(thin_prism, {C, 4}));
    put(std::get<0>(r), radii), put(std::get<1>(r), means2d), put(std::get<2>(r), depths), put(std::get<3>(r), conics), put(std::get<4>(r), compensations);
}

REFK_API int refk_rasterize_fwd(uint32_t cdim, uint32_t C, uint32_t N, uint32_t n_isects, const float* means, const float* quats, const float* scales,
                                const float* colors, const float* opacities, const float* backgrounds, const uint8_t* masks, uint32_t W, uint32_t H, uint32_t tile_size,
                                uint32_t tw, uint32_t th, const float* vm0, const float* vm1, const float* Ks, int camera_model, const float* ut, int rs_type,
                                const float* radial, const float* tangential, const float* thin_prism, const int32_t* offsets, const int32_t* flatten_ids,
                                float* render_colors, float* render_alphas, int32_t* last_ids) {
    try {
        at::optional<at::Tensor> m = masks ? at::optional<at::Tensor>(torch::from_blob(const_cast<uint8_t*>(masks), {C, th, tw}, torch::kUInt8).clone().to(torch::kBool))
                                           : at::nullopt;
        auto r = gsplat::rasterize_to_pixels_from_world_3dgs_fwd(f32(means, {N, 3}), f32(quats, {N, 4}), f32(scales, {N, 3}), f32(colors, {C, N, cdim}),
                                                                 f32(opacities, {C, N}), optf(backgrounds, {C, cdim}), m, W, H, tile_size, f32(vm0, {C, 4, 4}),
                                                                 optf(vm1, {C, 4, 4}), f32(Ks, {C, 3, 3}), gsplat::CameraModelType(camera_model), ut_of(ut),
                                                                 ShutterType(rs_type), optf(radial, {C, n_radial(camera_model)}), optf(tangential, {C, 2}),
                                                                 optf(thin_prism, {C, 4}), i32(offsets, {C, th, tw}), i32(flatten_ids, {n_isects}));
        put(std::get<0>(r), render_colors), put(std::get<1>(r), render_alphas), put(std::get<2>(r), last_ids);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "refk_rasterize_fwd: %s\n", e.what());
        return -1;
    }
}

REFK_API int refk_rasterize_bwd(uint32_t cdim, uint32_t C, uint32_t N, uint32_t n_isects, const float* means, const float* quats, const float* scales,
                                const float* colors, const float* opacities, const float* backgrounds, const uint8_t* masks, uint32_t W, uint32_t H, uint32_t tile_size,
                                uint32_t tw, uint32_t th, const float* vm0, const float* vm1, const float* Ks, int camera_model, const float* ut, int rs_type,
                                const float* radial, const float* tangential, const float* thin_prism, const int32_t* offsets, const int32_t* flatten_ids,
                                const float* render_alphas, const int32_t* last_ids, const float* v_render_colors, const float* v_render_alphas, float* v_means,
                                float* v_quats, float* v_scales, float* v_colors, float* v_opacities) {
    try {
        at::optional<at::Tensor> m = masks ? at::optional<at::Tensor>(torch::from_blob(const_cast<uint8_t*>(masks), {C, th, tw}, torch::kUInt8).clone().to(torch::kBool))
                                           : at::nullopt;
        auto r = gsplat::rasterize_to_pixels_from_world_3dgs_bwd(
            f32(means, {N, 3}), f32(quats, {N, 4}), f32(scales, {N, 3}), f32(colors, {C, N, cdim}), f32(opacities, {C, N}), optf(backgrounds, {C, cdim}), m, W, H, tile_size,
            f32(vm0, {C, 4, 4}), optf(vm1, {C, 4, 4}), f32(Ks, {C, 3, 3}), gsplat::CameraModelType(camera_model), ut_of(ut), ShutterType(rs_type),
            optf(radial, {C, n_radial(camera_model)}), optf(tangential, {C, 2}), optf(thin_prism, {C, 4}), i32(offsets, {C, th, tw}), i32(flatten_ids, {n_isects}),
            f32(render_alphas, {C, H, W, 1}), i32(last_ids, {C, H, W}), f32(v_render_colors, {C, H, W, cdim}), f32(v_render_alphas, {C, H, W, 1}));
        put(std::get<0>(r), v_means), put(std::get<1>(r), v_quats), put(std::get<2>(r), v_scales), put(std::get<3>(r), v_colors), put(std::get<4>(r), v_opacities);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "refk_rasterize_bwd: %s\n", e.what());
        return -1;
    }
}

REFK_API void refk_relocation(int64_t N, float* opacities, float* scales, int* ratios, float* binoms, int n_max, float* new_opacities, float* new_scales) {
    auto r = gsplat::relocation(f32(opacities, {N}), f32(scales, {N, 3}), i32(ratios, {N}), f32(binoms, {n_max, n_max}), n_max);
    put(std::get<0>(r), new_opacities), put(std::get<1>(r), new_scales);
}
REFK_API void refk_add_noise(int64_t N, float* raw_opacities, float* raw_scales, float* raw_quats, float* noise, float* means, float current_lr) {
    auto m = f32(means, {N, 3});
    gsplat::add_noise(f32(raw_opacities, {N}), f32(raw_scales, {N, 3}), f32(raw_quats, {N, 4}), f32(noise, {N, 3}), m, current_lr);
    put(m, means);
}
REFK_API void refk_quats_to_rotmats(int64_t N, const float* quats, float* rotmats) { put(gsplat::quats_to_rotmats(f32(quats, {N, 4})), rotmats); }

REFK_API void refk_sh_fwd(uint32_t N, uint32_t K, uint32_t degrees_to_use, const float* dirs, const float* coeffs, const uint8_t* masks, float* colors) {
    at::optional<at::Tensor> m = masks ? at::optional<at::Tensor>(torch::from_blob(const_cast<uint8_t*>(masks), {N}, torch::kUInt8).clone().to(torch::kBool)) : at::nullopt;
    auto c = gsplat::spherical_harmonics_fwd(degrees_to_use, f32(dirs, {N, 3}), f32(coeffs, {N, K, 3}), m);
    if (m.has_value()) c = at::where(m->unsqueeze(-1), c, at::zeros_like(c));   // (at::empty in the operator: the rows the kernel skips are undefined; 0 in the files)
    put(c, colors);
}
REFK_API void refk_sh_bwd(uint32_t N, uint32_t K, uint32_t degrees_to_use, const float* dirs, const float* coeffs, const uint8_t* masks, const float* v_colors,
                          float* v_coeffs, float* v_dirs) {
    at::optional<at::Tensor> m = masks ? at::optional<at::Tensor>(torch::from_blob(const_cast<uint8_t*>(masks), {N}, torch::kUInt8).clone().to(torch::kBool)) : at::nullopt;
    auto r = gsplat::spherical_harmonics_bwd(K, degrees_to_use, f32(dirs, {N, 3}), f32(coeffs, {N, K, 3}), m, f32(v_colors, {N, 3}), v_dirs != nullptr);
    put(std::get<0>(r), v_coeffs), put(std::get<1>(r), v_dirs);
}

REFK_API int64_t refk_intersect_tile(uint32_t C, uint32_t N, const float* means2d, const int32_t* radii, const float* depths, uint32_t tile_size, uint32_t tw,
                                     uint32_t th, int sort, int32_t* tiles_per_gauss, int64_t* isect_ids, int32_t* flatten_ids, int64_t cap) {
    auto r = gsplat::intersect_tile(f32(means2d, {C, N, 2}), i32(radii, {C, N, 2}), f32(depths, {C, N}), at::nullopt, at::nullopt, C, tile_size, tw, th, sort != 0);
    put(std::get<0>(r), tiles_per_gauss);
    const int64_t n = std::get<1>(r).numel();
    if (n && n <= cap) put(std::get<1>(r), isect_ids), put(std::get<2>(r), flatten_ids);
    return n;
}
REFK_API void refk_intersect_offset(int64_t n_isects, const int64_t* isect_ids, uint32_t C, uint32_t tw, uint32_t th, int32_t* offsets) {
    auto ids = torch::from_blob(const_cast<int64_t*>(isect_ids), {n_isects}, torch::kInt64).clone();
    put(gsplat::intersect_offset(ids, C, tw, th), offsets);
}
