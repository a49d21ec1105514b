// ORACLE/_ref — TEST INFRASTRUCTURE ONLY.
// C entry points over the reference's OWN device kernels, run on the CPU: the kernel parts of
//   /root/reference/gsplat/ProjectionUT3DGSFused.cu:1-203, RasterizeToPixelsFromWorld3DGSFwd.cu:1-279, ...Bwd.cu:1-373,
//   RelocationCUDA.cu:1-43 + 87-144, QuatToRotmatCUDA.cu:1-39, SphericalHarmonicsCUDA.cu:1-400 + 443-481, IntersectTile.cu:1-114 + 206-252
//   (with Cameras.cuh, Utils.cuh, Common.h, Cameras.h as they include them)
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
#include "k_sh.inc"      // SphericalHarmonicsCUDA.cu:1-400 + 443-481 (the basis evaluation, its vjp, the two kernels)
#include "k_isect.inc"   // IntersectTile.cu:1-114 + 206-252 (intersect_tile_kernel, intersect_offset_kernel)
#include "adam_kernels.cuh"
#include <cmath>
#include <cub/cub.cuh>

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


// ---- spherical harmonics (K2 / K9): launch geometry of launch_spherical_harmonics_{fwd,bwd}_kernel (SphericalHarmonicsCUDA.cu:402-441, 483-529): one thread per
// (Gaussian, channel), 256 threads; the caller provides zeroed v_coeffs / v_dirs as SphericalHarmonics.cpp:60-66 does (zeros_like) ---------------------------
REFK_API void refk_sh_fwd(uint32_t N, uint32_t K, uint32_t degrees_to_use, const float* dirs, const float* coeffs, const uint8_t* masks, float* colors) {
    if (N == 0) return;
    cuemu::launch(dim3(unsigned((3ull * N + 255) / 256)), dim3(256), false, [&]() {
        gsplat::spherical_harmonics_fwd_kernel<float>(N, K, degrees_to_use, reinterpret_cast<const gsplat::vec3*>(dirs), coeffs,
                                                      reinterpret_cast<const bool*>(masks), colors);
    });
}
REFK_API void refk_sh_bwd(uint32_t N, uint32_t K, uint32_t degrees_to_use, const float* dirs, const float* coeffs, const uint8_t* masks, const float* v_colors,
                          float* v_coeffs, float* v_dirs) {
    if (N == 0) return;
    cuemu::launch(dim3(unsigned((3ull * N + 255) / 256)), dim3(256), false, [&]() {
        gsplat::spherical_harmonics_bwd_kernel<float>(N, K, degrees_to_use, reinterpret_cast<const gsplat::vec3*>(dirs), coeffs,
                                                      reinterpret_cast<const bool*>(masks), v_colors, v_coeffs, v_dirs);
    });
}

// ---- tile intersection (K3 - K6): the host sequence of gsplat::intersect_tile (Intersect.cpp:15-122) over the two passes of intersect_tile_kernel, at::cumsum
// and radix_sort_double_buffer (IntersectTile.cu:290-342: cub::DeviceRadixSort::SortPairs over the key bits [0, 32 + tile bits + camera bits), restated in
// ref_emul/cub/cub.cuh as the stable sort CUB documents). Returns n_isects; the id arrays are written when they fit `cap` (call with cap = 0 to size).
static void isect_pass(uint32_t C, uint32_t N, const float* means2d, const int32_t* radii, const float* depths, const int64_t* cum, uint32_t tile_size, uint32_t tw,
                       uint32_t th, uint32_t tile_n_bits, int32_t* tiles_per_gauss, int64_t* isect_ids, int32_t* flatten_ids) {
    const uint64_t n = uint64_t(C) * N;
    cuemu::launch(dim3(unsigned((n + 255) / 256)), dim3(256), false, [&]() {
        gsplat::intersect_tile_kernel<float>(false, C, N, 0, nullptr, nullptr, means2d, radii, depths, cum, tile_size, tw, th, tile_n_bits, tiles_per_gauss, isect_ids,
                                             flatten_ids);
    });
}
REFK_API int64_t refk_intersect_tile(uint32_t C, uint32_t N, const float* means2d, const int32_t* radii, const float* depths, uint32_t tile_size, uint32_t tw,
                                     uint32_t th, int sort, int32_t* tiles_per_gauss, int64_t* isect_ids, int32_t* flatten_ids, int64_t cap) {
    const uint64_t n = uint64_t(C) * N;
    if (n == 0) return 0;
    const uint32_t n_tiles = tw * th;
    const uint32_t tile_n_bits = (uint32_t)floor(log2(n_tiles)) + 1, cam_n_bits = (uint32_t)floor(log2(C)) + 1;   // Intersect.cpp:45-46
    isect_pass(C, N, means2d, radii, depths, nullptr, tile_size, tw, th, tile_n_bits, tiles_per_gauss, nullptr, nullptr);
    std::vector<int64_t> cum(n);
    int64_t run = 0;
    for (uint64_t i = 0; i < n; ++i) cum[i] = (run += tiles_per_gauss[i]);                                           // at::cumsum, :75
    const int64_t n_isects = run;
    if (n_isects == 0 || n_isects > cap) return n_isects;
    std::vector<int64_t> ids(n_isects), ids2(n_isects);
    std::vector<int32_t> flat(n_isects), flat2(n_isects);
    isect_pass(C, N, means2d, radii, depths, cum.data(), tile_size, tw, th, tile_n_bits, nullptr, ids.data(), flat.data());
    int64_t* ko = ids.data();
    int32_t* vo = flat.data();
    if (sort) {
        cub::DoubleBuffer<int64_t> d_keys(ids.data(), ids2.data());
        cub::DoubleBuffer<int32_t> d_values(flat.data(), flat2.data());
        size_t bytes = 0;
        cub::DeviceRadixSort::SortPairs(nullptr, bytes, d_keys, d_values, (int)n_isects, 0, int(32 + tile_n_bits + cam_n_bits));
        char ws;
        cub::DeviceRadixSort::SortPairs(&ws, bytes, d_keys, d_values, (int)n_isects, 0, int(32 + tile_n_bits + cam_n_bits));
        ko = d_keys.Current(), vo = d_values.Current();
    }
    std::copy(ko, ko + n_isects, isect_ids);
    std::copy(vo, vo + n_isects, flatten_ids);
    return n_isects;
}
// gsplat::intersect_offset (Intersect.cpp:124-137, launch_intersect_offset_kernel IntersectTile.cu:254-288): offsets.fill_(0) when there is nothing
REFK_API void refk_intersect_offset(int64_t n_isects, const int64_t* isect_ids, uint32_t C, uint32_t tw, uint32_t th, int32_t* offsets) {
    const uint32_t n_tiles = tw * th;
    if (n_isects == 0) {
        std::fill(offsets, offsets + size_t(C) * n_tiles, 0);
        return;
    }
    const uint32_t tile_n_bits = (uint32_t)floor(log2(n_tiles)) + 1;
    cuemu::launch(dim3(unsigned((n_isects + 255) / 256)), dim3(256), false,
                  [&]() { gsplat::intersect_offset_kernel(uint32_t(n_isects), isect_ids, C, n_tiles, tile_n_bits, offsets); });
}
