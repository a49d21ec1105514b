// ORACLE/_ref — TEST INFRASTRUCTURE ONLY.
// C entry points over the reference's OWN device kernels, run on the CPU: the kernel parts of
//   /root/reference/gsplat/ProjectionUT3DGSFused.cu:1-203, RasterizeToPixelsFromWorld3DGSFwd.cu:1-279, ...Bwd.cu:1-373,
//   RelocationCUDA.cu:1-43 + 87-144, QuatToRotmatCUDA.cu:1-39 (with Cameras.cuh, Utils.cuh, Common.h, Cameras.h as they include them)
//   and fastgs/optimizer/include/adam_kernels.cuh
// are compiled IN PLACE as host C++ by `make -C oracle refk` (the recipe pipes the kernel line ranges into a scratch directory that is
// deleted after the compile: the launchers below them use the <<<...>>> syntax, which no host compiler parses; nothing of the reference is
// copied into the repository or into oracle/_ref, which only receives the .so). CUDA's execution model comes from oracle/ref_emul/
// (cuda_emul.h: threads as fibers, 32-lane warps, cooperative groups; glm/: the GLM subset the kernels use - GLM is a third-party
// dependency that is not vendored in the reference tree). oracle/make_golden_refk.py runs these entry points and writes
// tests/golden/refk_*.npz, against which BOTH the oracle restatement (oracle_ops.hpp) and the HIP kernels are checked: this is the pin of
// projection_ut_3dgs_fused, rasterize_to_pixels_from_world_3dgs_{fwd,bwd}, relocation, add_noise, quats_to_rotmats and adam_step to the
// reference itself (SURVEY.md §8c lists them as unpinned by the reference's tests). Nothing here is product code.
#include "k_projection.inc"
#include "k_raster_fwd.inc"
#include "k_raster_bwd.inc"
#include "k_relocation.inc"
#include "k_quat.inc"
#include "adam_kernels.cuh"

namespace gsplat {
alignas(64) int s[1 << 18]; // `extern __shared__ int s[]` of the rasterizer kernels: 1 MiB, one workgroup runs at a time
}

#define REFK_API extern "C" __attribute__((visibility("default")))

static UnscentedTransformParameters ut_of(const float* ut) {
    UnscentedTransformParameters p;
    if (ut) { p.alpha = ut[0]; p.beta = ut[1]; p.kappa = ut[2]; p.in_image_margin_factor = ut[3]; p.require_all_sigma_points_valid = ut[4] != 0.f; }
    return p;
}

// The kernels read fixed-size coefficient arrays when the pointer is non-null (pinhole: 6 radial, 2 tangential, 4 thin-prism; fisheye: 4
// radial - Fwd.cu:100-121): the callers pass exactly those sizes.
REFK_API void refk_projection_ut(uint32_t C, uint32_t N, const float* means, const float* quats, const float* scales, const float* opacities,
                                 const float* viewmats0, const float* viewmats1, const float* Ks, uint32_t W, uint32_t H, float eps2d,
                                 float near_plane, float far_plane, float radius_clip, int camera_model, const float* ut, int rs_type,
                                 const float* radial, const float* tangential, const float* thin_prism,
                                 int32_t* radii, float* means2d, float* depths, float* conics, float* compensations) {
    const int64_t n = int64_t(C) * N;
    if (n == 0) return;
    const dim3 threads(256), grid(unsigned((n + 255) / 256));
    cuemu::launch(grid, threads, false, [&]() {
        gsplat::projection_ut_3dgs_fused_kernel<float>(C, N, means, quats, scales, opacities, viewmats0, viewmats1, Ks, W, H, eps2d, near_plane, far_plane,
                                                      radius_clip, gsplat::CameraModelType(camera_model), ut_of(ut), ShutterType(rs_type), radial, tangential,
                                                      thin_prism, radii, means2d, depths, conics, compensations);
    });
}

template <uint32_t CDIM>
static void raster_fwd(uint32_t C, uint32_t N, uint32_t n_isects, const float* means, const float* quats, const float* scales, const float* colors,
                       const float* opacities, const float* backgrounds, const bool* masks, uint32_t W, uint32_t H, uint32_t tile_size, uint32_t tw,
                       uint32_t th, const float* vm0, const float* vm1, const float* Ks, int camera_model, const float* ut, int rs_type,
                       const float* radial, const float* tangential, const float* thin_prism, const int32_t* offsets, const int32_t* flatten_ids,
                       float* render_colors, float* render_alphas, int32_t* last_ids) {
    const dim3 threads(tile_size, tile_size, 1), grid(C, th, tw);
    cuemu::launch(grid, threads, true, [&]() {
        gsplat::rasterize_to_pixels_from_world_3dgs_fwd_kernel<CDIM, float>(
            C, N, n_isects, false, reinterpret_cast<const gsplat::vec3*>(means), reinterpret_cast<const gsplat::vec4*>(quats),
            reinterpret_cast<const gsplat::vec3*>(scales), colors, opacities, backgrounds, masks, W, H, tile_size, tw, th, vm0, vm1, Ks,
            gsplat::CameraModelType(camera_model), ut_of(ut), ShutterType(rs_type), radial, tangential, thin_prism, offsets, flatten_ids,
            render_colors, render_alphas, last_ids);
    });
}

REFK_API int refk_rasterize_fwd(uint32_t cdim, uint32_t C, uint32_t N, uint32_t n_isects, const float* means, const float* quats, const float* scales,
                                const float* colors, const float* opacities, const float* backgrounds, const uint8_t* masks, uint32_t W, uint32_t H,
                                uint32_t tile_size, uint32_t tw, uint32_t th, const float* vm0, const float* vm1, const float* Ks, int camera_model,
                                const float* ut, int rs_type, const float* radial, const float* tangential, const float* thin_prism,
                                const int32_t* offsets, const int32_t* flatten_ids, float* render_colors, float* render_alphas, int32_t* last_ids) {
    static_assert(sizeof(bool) == 1, "masks are passed as bytes");
    const bool* m = reinterpret_cast<const bool*>(masks);
#define REFK_FWD(D) case D: raster_fwd<D>(C, N, n_isects, means, quats, scales, colors, opacities, backgrounds, m, W, H, tile_size, tw, th, vm0, vm1, Ks, camera_model, ut, \
                                          rs_type, radial, tangential, thin_prism, offsets, flatten_ids, render_colors, render_alphas, last_ids); return 0;
    switch (cdim) { REFK_FWD(1) REFK_FWD(2) REFK_FWD(3) REFK_FWD(4) default: return -1; }
#undef REFK_FWD
}

template <uint32_t CDIM>
static void raster_bwd(uint32_t C, uint32_t N, uint32_t n_isects, const float* means, const float* quats, const float* scales, const float* colors,
                       const float* opacities, const float* backgrounds, const bool* masks, uint32_t W, uint32_t H, uint32_t tile_size, uint32_t tw,
                       uint32_t th, const float* vm0, const float* vm1, const float* Ks, int camera_model, const float* ut, int rs_type,
                       const float* radial, const float* tangential, const float* thin_prism, const int32_t* offsets, const int32_t* flatten_ids,
                       const float* render_alphas, const int32_t* last_ids, const float* v_render_colors, const float* v_render_alphas,
                       float* v_means, float* v_quats, float* v_scales, float* v_colors, float* v_opacities) {
    if (n_isects == 0) return; // Bwd.cu:434-437
    const dim3 threads(tile_size, tile_size, 1), grid(C, th, tw);
    cuemu::launch(grid, threads, true, [&]() {
        gsplat::rasterize_to_pixels_from_world_3dgs_bwd_kernel<CDIM, float>(
            C, N, n_isects, false, reinterpret_cast<const gsplat::vec3*>(means), reinterpret_cast<const gsplat::vec4*>(quats),
            reinterpret_cast<const gsplat::vec3*>(scales), colors, opacities, backgrounds, masks, W, H, tile_size, tw, th, vm0, vm1, Ks,
            gsplat::CameraModelType(camera_model), ut_of(ut), ShutterType(rs_type), radial, tangential, thin_prism, offsets, flatten_ids,
            render_alphas, last_ids, v_render_colors, v_render_alphas, reinterpret_cast<gsplat::vec3*>(v_means),
            reinterpret_cast<gsplat::vec4*>(v_quats), reinterpret_cast<gsplat::vec3*>(v_scales), v_colors, v_opacities);
    });
}

REFK_API int refk_rasterize_bwd(uint32_t cdim, uint32_t C, uint32_t N, uint32_t n_isects, const float* means, const float* quats, const float* scales,
                                const float* colors, const float* opacities, const float* backgrounds, const uint8_t* masks, uint32_t W, uint32_t H,
                                uint32_t tile_size, uint32_t tw, uint32_t th, const float* vm0, const float* vm1, const float* Ks, int camera_model,
                                const float* ut, int rs_type, const float* radial, const float* tangential, const float* thin_prism,
                                const int32_t* offsets, const int32_t* flatten_ids, const float* render_alphas, const int32_t* last_ids,
                                const float* v_render_colors, const float* v_render_alphas, float* v_means, float* v_quats, float* v_scales,
                                float* v_colors, float* v_opacities) {
    const bool* m = reinterpret_cast<const bool*>(masks);
#define REFK_BWD(D) case D: raster_bwd<D>(C, N, n_isects, means, quats, scales, colors, opacities, backgrounds, m, W, H, tile_size, tw, th, vm0, vm1, Ks, camera_model, ut, \
                                          rs_type, radial, tangential, thin_prism, offsets, flatten_ids, render_alphas, last_ids, v_render_colors,                        \
                                          v_render_alphas, v_means, v_quats, v_scales, v_colors, v_opacities); return 0;
    switch (cdim) { REFK_BWD(1) REFK_BWD(2) REFK_BWD(3) REFK_BWD(4) default: return -1; }
#undef REFK_BWD
}

REFK_API void refk_relocation(int64_t N, float* opacities, float* scales, int* ratios, float* binoms, int n_max, float* new_opacities, float* new_scales) {
    if (N == 0) return;
    cuemu::launch(dim3(unsigned((N + 255) / 256)), dim3(256), false,
                  [&]() { gsplat::relocation_kernel<float>(int(N), opacities, scales, ratios, binoms, n_max, new_opacities, new_scales); });
}

REFK_API void refk_add_noise(int64_t N, float* raw_opacities, float* raw_scales, float* raw_quats, float* noise, float* means, float current_lr) {
    if (N == 0) return;
    cuemu::launch(dim3(unsigned((N + 255) / 256)), dim3(256), false,
                  [&]() { gsplat::add_noise_kernel<float>(int(N), raw_opacities, raw_scales, raw_quats, noise, means, current_lr); });
}

REFK_API void refk_quats_to_rotmats(int64_t N, const float* quats, float* rotmats) {
    if (N == 0) return;
    cuemu::launch(dim3(unsigned((N + 255) / 256)), dim3(256), false, [&]() { gsplat::quat_to_rotmat_kernel<float>(uint32_t(N), quats, rotmats); });
}

REFK_API void refk_adam_step(int64_t n, float* param, float* exp_avg, float* exp_avg_sq, const float* grad, float lr, float beta1, float beta2, float eps,
                             float bias_correction1_rcp, float bias_correction2_sqrt_rcp) {
    if (n == 0) return;
    cuemu::launch(dim3(unsigned((n + 255) / 256)), dim3(256), false, [&]() {
        fast_gs::optimizer::kernels::adam::adam_step_cu(param, exp_avg, exp_avg_sq, grad, int(n), lr, beta1, beta2, eps, bias_correction1_rcp,
                                                        bias_correction2_sqrt_rcp);
    });
}
