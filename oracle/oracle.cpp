// ORACLE — TEST INFRASTRUCTURE ONLY. C entry points (ctypes) over oracle_ops.hpp.
// Built by oracle/Makefile into oracle/liboracle.so. Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
#include "oracle_ops.hpp"
#include "oracle_fastgs.hpp"

using namespace orc;

#define ORC_API extern "C" __attribute__((visibility("default")))

template <class T> static UTParams<T> ut_from(const T* p) {
    UTParams<T> ut;
    if (p) { ut.alpha = p[0]; ut.beta = p[1]; ut.kappa = p[2]; ut.in_image_margin_factor = p[3]; ut.require_all_sigma_points_valid = p[4] != T(0); }
    return ut;
}

#define DEFINE_FLOAT_API(SFX, T)                                                                              \
ORC_API void orc_quats_to_rotmats_##SFX(int64_t N, const T* quats, T* rotmats) { quats_to_rotmats<T>(N, quats, rotmats); } \
ORC_API void orc_projection_ut_##SFX(                                                                         \
    uint32_t C, uint32_t N, const T* means, const T* quats, const T* scales, const T* opacities,              \
    const T* viewmats0, const T* viewmats1, const T* Ks, uint32_t width, uint32_t height,                     \
    T eps2d, T near_plane, T far_plane, T radius_clip, int camera_model, const T* ut5, int shutter,           \
    const T* radial, int n_radial, const T* tangential, const T* thin_prism, int n_thin,                      \
    int32_t* radii, T* means2d, T* depths, T* conics, T* compensations) {                                     \
    ProjArgs<T> a{C, N, means, quats, scales, opacities, viewmats0, viewmats1, Ks, width, height,             \
                  eps2d, near_plane, far_plane, radius_clip, camera_model, ut_from<T>(ut5), shutter,          \
                  radial, n_radial, tangential, thin_prism, n_thin, radii, means2d, depths, conics, compensations}; \
    projection_ut_3dgs_fused<T>(a);                                                                           \
}                                                                                                             \
ORC_API void orc_sh_fwd_##SFX(int64_t N, int K, int degree, const T* dirs, const T* coeffs, const uint8_t* masks, T* colors) { \
    spherical_harmonics_fwd<T>(N, K, degree, dirs, coeffs, masks, colors);                                    \
}                                                                                                             \
ORC_API void orc_sh_bwd_##SFX(int64_t N, int K, int degree, const T* dirs, const T* coeffs, const uint8_t* masks, \
                              const T* v_colors, T* v_coeffs, T* v_dirs) {                                    \
    spherical_harmonics_bwd<T>(N, K, degree, dirs, coeffs, masks, v_colors, v_coeffs, v_dirs);                \
}                                                                                                             \
ORC_API void orc_rasterize_fwd_##SFX(                                                                         \
    uint32_t C, uint32_t N, int64_t n_isects, uint32_t cdim, const T* means, const T* quats, const T* scales, \
    const T* colors, const T* opacities, const T* backgrounds, const uint8_t* masks,                          \
    uint32_t width, uint32_t height, uint32_t tile_size, const T* viewmats0, const T* viewmats1, const T* Ks, \
    int camera_model, int shutter, const T* radial, int n_radial, const T* tangential, const T* thin_prism, int n_thin, \
    const int32_t* tile_offsets, const int32_t* flatten_ids, T* render_colors, T* render_alphas, int32_t* last_ids) { \
    RasterArgs<T> a{C, N, n_isects, cdim, means, quats, scales, colors, opacities, backgrounds, masks,        \
                    width, height, tile_size, (width + tile_size - 1) / tile_size, (height + tile_size - 1) / tile_size, \
                    viewmats0, viewmats1, Ks, camera_model, shutter, radial, n_radial, tangential, thin_prism, n_thin, \
                    tile_offsets, flatten_ids};                                                               \
    rasterize_fwd<T>(a, render_colors, render_alphas, last_ids);                                              \
}                                                                                                             \
ORC_API void orc_rasterize_bwd_##SFX(                                                                         \
    uint32_t C, uint32_t N, int64_t n_isects, uint32_t cdim, const T* means, const T* quats, const T* scales, \
    const T* colors, const T* opacities, const T* backgrounds, const uint8_t* masks,                          \
    uint32_t width, uint32_t height, uint32_t tile_size, const T* viewmats0, const T* viewmats1, const T* Ks, \
    int camera_model, int shutter, const T* radial, int n_radial, const T* tangential, const T* thin_prism, int n_thin, \
    const int32_t* tile_offsets, const int32_t* flatten_ids, const T* render_alphas, const int32_t* last_ids, \
    const T* v_render_colors, const T* v_render_alphas,                                                       \
    T* v_means, T* v_quats, T* v_scales, T* v_colors, T* v_opacities) {                                       \
    RasterArgs<T> a{C, N, n_isects, cdim, means, quats, scales, colors, opacities, backgrounds, masks,        \
                    width, height, tile_size, (width + tile_size - 1) / tile_size, (height + tile_size - 1) / tile_size, \
                    viewmats0, viewmats1, Ks, camera_model, shutter, radial, n_radial, tangential, thin_prism, n_thin, \
                    tile_offsets, flatten_ids};                                                               \
    rasterize_bwd<T>(a, render_alphas, last_ids, v_render_colors, v_render_alphas, v_means, v_quats, v_scales, v_colors, v_opacities); \
}                                                                                                             \
ORC_API void orc_relocation_##SFX(int64_t N, const T* opacities, const T* scales, const int32_t* ratios,      \
                                  const T* binoms, int n_max, T* new_opacities, T* new_scales) {              \
    relocation<T>(N, opacities, scales, ratios, binoms, n_max, new_opacities, new_scales);                    \
}                                                                                                             \
ORC_API void orc_add_noise_##SFX(int64_t N, const T* raw_opacities, const T* raw_scales, const T* raw_quats,  \
                                 const T* noise, T* means, T current_lr) {                                    \
    add_noise<T>(N, raw_opacities, raw_scales, raw_quats, noise, means, current_lr);                          \
}                                                                                                             \
ORC_API void orc_adam_step_##SFX(int64_t n, T* param, T* exp_avg, T* exp_avg_sq, const T* grad,               \
                                 T lr, T beta1, T beta2, T eps, T bc1_rcp, T bc2_sqrt_rcp) {                  \
    adam_step<T>(n, param, exp_avg, exp_avg_sq, grad, lr, beta1, beta2, eps, bc1_rcp, bc2_sqrt_rcp);          \
}                                                                                                             \
/* pixel ray of one camera: lets tests check camera models in isolation */                                    \
ORC_API void orc_pixel_rays_##SFX(uint32_t width, uint32_t height, const T* viewmat0, const T* viewmat1, const T* K, \
                                  int camera_model, int shutter, const T* radial, int n_radial, const T* tangential, \
                                  const T* thin_prism, int n_thin, T* origins, T* dirs, uint8_t* valid) {     \
    Camera<T> cam = make_camera<T>(camera_model, width, height, shutter, K, radial, n_radial, tangential, thin_prism, n_thin); \
    RSParams<T> rs(viewmat0, viewmat1);                                                                       \
    for (uint32_t i = 0; i < height; ++i) for (uint32_t j = 0; j < width; ++j) {                              \
        Ray<T> r = cam.pixel_ray({T(j) + T(0.5f), T(i) + T(0.5f)}, rs);                                       \
        int64_t p = int64_t(i) * width + j;                                                                   \
        origins[3 * p] = r.o.x; origins[3 * p + 1] = r.o.y; origins[3 * p + 2] = r.o.z;                       \
        dirs[3 * p] = r.d.x; dirs[3 * p + 1] = r.d.y; dirs[3 * p + 2] = r.d.z; valid[p] = r.valid;            \
    }                                                                                                         \
}

DEFINE_FLOAT_API(f32, float)
DEFINE_FLOAT_API(f64, double)

// ---- integer stage (float32 inputs only: bit-exact contract) ----------------
ORC_API int64_t orc_intersect_tile_count(uint32_t C, uint32_t N, const float* means2d, const int32_t* radii,
                                         uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, int32_t* tiles_per_gauss) {
    return intersect_tile_count(C, N, means2d, radii, tile_size, tile_width, tile_height, tiles_per_gauss);
}
ORC_API void orc_intersect_tile_emit(uint32_t C, uint32_t N, const float* means2d, const int32_t* radii, const float* depths,
                                     uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, int sort, int64_t n_isects,
                                     int64_t* isect_ids, int32_t* flatten_ids) {
    intersect_tile_emit(C, N, means2d, radii, depths, tile_size, tile_width, tile_height, sort != 0, n_isects, isect_ids, flatten_ids);
}
ORC_API void orc_intersect_offset(int64_t n_isects, const int64_t* isect_ids, uint32_t C, uint32_t tile_width, uint32_t tile_height, int32_t* offsets) {
    intersect_offset(n_isects, isect_ids, C, tile_width, tile_height, offsets);
}
// ---- fastgs (EWA) rasterizer: forward keeps its per-primitive state and tile lists for the matching backward call ----
template <class T> struct FgState { std::vector<fg::Prim<T>> P; fg::Lists L; };
template <class T> static FgState<T>& fg_state() { static FgState<T> s; return s; }
#define DEFINE_FASTGS_API(SFX, T)                                                                                      \
ORC_API int64_t orc_fastgs_forward_##SFX(int64_t N, const T* means, const T* scales_raw, const T* rot_raw, const T* opac_raw,    \
    const T* sh0, const T* sh_rest, const T* w2c, const T* cam_pos, int active_sh_bases, int total_rest, int W, int H,  \
    T fx, T fy, T cx, T cy, T near_, T far_, T* image, T* alpha, int32_t* n_contrib,                                   \
    T* mean2d, T* conic_opacity, T* color, int32_t* n_touched, int32_t* bounds) {                                       \
    fg::Args<T> a{N, means, scales_raw, rot_raw, opac_raw, sh0, sh_rest, w2c, cam_pos, active_sh_bases, total_rest, W, H, fx, fy, cx, cy, near_, far_}; \
    FgState<T>& st = fg_state<T>();                                                                                     \
    fg::preprocess<T>(a, st.P);                                                                                         \
    fg::build_lists<T>(a, st.P, st.L);                                                                                  \
    fg::blend<T>(a, st.P, st.L, image, alpha, n_contrib);                                                               \
    for (int64_t i = 0; i < N; ++i) {                                                                                   \
        const fg::Prim<T>& p = st.P[i];                                                                                 \
        mean2d[2 * i] = p.mx; mean2d[2 * i + 1] = p.my;                                                                 \
        conic_opacity[4 * i] = p.ca; conic_opacity[4 * i + 1] = p.cb; conic_opacity[4 * i + 2] = p.cc; conic_opacity[4 * i + 3] = p.opacity; \
        color[3 * i] = p.col[0]; color[3 * i + 1] = p.col[1]; color[3 * i + 2] = p.col[2];                              \
        n_touched[i] = int32_t(p.n_touched);                                                                            \
        bounds[4 * i] = int32_t(p.x0); bounds[4 * i + 1] = int32_t(p.x1); bounds[4 * i + 2] = int32_t(p.y0); bounds[4 * i + 3] = int32_t(p.y1); \
    }                                                                                                                   \
    return int64_t(st.L.ids.size());                                                                                    \
}                                                                                                                       \
ORC_API void orc_fastgs_lists_##SFX(int32_t* offsets, int32_t* ids) {                                                   \
    FgState<T>& st = fg_state<T>();                                                                                     \
    std::copy(st.L.offsets.begin(), st.L.offsets.end(), offsets); std::copy(st.L.ids.begin(), st.L.ids.end(), ids);     \
}                                                                                                                       \
ORC_API void orc_fastgs_backward_##SFX(int64_t N, const T* means, const T* scales_raw, const T* rot_raw, const T* opac_raw,       \
    const T* sh0, const T* sh_rest, const T* w2c, const T* cam_pos, int active_sh_bases, int total_rest, int W, int H,  \
    T fx, T fy, T cx, T cy, T near_, T far_, const T* image, const T* alpha, const int32_t* n_contrib,                  \
    const T* g_image, const T* g_alpha, T* g_means, T* g_scales_raw, T* g_rot_raw, T* g_opac_raw, T* g_sh0, T* g_sh_rest, T* densification_info) { \
    fg::Args<T> a{N, means, scales_raw, rot_raw, opac_raw, sh0, sh_rest, w2c, cam_pos, active_sh_bases, total_rest, W, H, fx, fy, cx, cy, near_, far_}; \
    FgState<T>& st = fg_state<T>();                                                                                     \
    fg::backward<T>(a, st.P, st.L, image, alpha, n_contrib, g_image, g_alpha, g_means, g_scales_raw, g_rot_raw, g_opac_raw, g_sh0, g_sh_rest, densification_info); \
}
DEFINE_FASTGS_API(f32, float)
DEFINE_FASTGS_API(f64, double)

ORC_API int orc_version() { return 2; }
