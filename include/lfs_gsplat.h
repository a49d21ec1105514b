/* lfs_gsplat.h — C ABI of the MI355X-native (gfx950) 3DGS training rasterizer.
 *
 * Drop-in boundary for the reference's native operator backend
 * (MrNeRF/LichtFeld-Studio): every entry point below replaces one function of
 * /root/reference/gsplat/Ops.h (namespace gsplat::) or of
 * /root/reference/fastgs/optimizer/include/adam.h, with the at::Tensor
 * arguments flattened to raw DEVICE pointers + sizes + a hipStream_t.
 * The libtorch wrappers that restore the exact Ops.h signatures live in
 * include/lfs_gsplat_torch.hpp + lichtfeld-studio_amd/csrc/torch_ops.cpp.
 *
 * Conventions (same as the reference):
 *   - all pointers are device pointers, contiguous, fp32 unless noted;
 *   - quaternions are (w,x,y,z) and need not be normalised;
 *   - viewmats are row-major world->camera [C,4,4]; Ks row-major [C,3,3];
 *   - scales / opacities are the ACTIVATED values (exp / sigmoid applied);
 *   - every function enqueues on `stream` and returns without synchronising;
 *   - return value: 0 on success, a hipError_t (>0) from the runtime, or a
 *     negative LFS_E_* code for argument errors. Nothing is launched on error.
 *   - "workspace" buffers are caller-owned scratch (size from the matching
 *     *_workspace_bytes function, 256-byte aligned); contents are undefined
 *     afterwards unless stated.
 */
#ifndef LFS_GSPLAT_H
#define LFS_GSPLAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LFS_API __attribute__((visibility("default")))

typedef void* lfs_stream_t; /* hipStream_t */

enum { LFS_OK = 0, LFS_E_INVALID = -1, LFS_E_UNSUPPORTED = -2, LFS_E_WORKSPACE = -3 };

/* gsplat/Common.h:46-50 */
enum { LFS_CAMERA_PINHOLE = 0, LFS_CAMERA_ORTHO = 1, LFS_CAMERA_FISHEYE = 2 };
/* gsplat/Cameras.h:16-22 */
enum {
    LFS_SHUTTER_ROLLING_TOP_TO_BOTTOM = 0,
    LFS_SHUTTER_ROLLING_LEFT_TO_RIGHT = 1,
    LFS_SHUTTER_ROLLING_BOTTOM_TO_TOP = 2,
    LFS_SHUTTER_ROLLING_RIGHT_TO_LEFT = 3,
    LFS_SHUTTER_GLOBAL = 4
};

/* gsplat/Cameras.h:27-61 UnscentedTransformParameters */
typedef struct lfs_ut_params {
    float alpha;                       /* 0.1 */
    float beta;                        /* 2   */
    float kappa;                       /* 0   */
    float in_image_margin_factor;      /* 0.1 */
    int32_t require_all_sigma_points_valid; /* 1 */
} lfs_ut_params;

/* The camera block every projection / rasterization entry point of Ops.h
 * takes (viewmats0, viewmats1, Ks, camera_model, rs_type, radial/tangential/
 * thin-prism coefficients, image size). n_radial / n_thin_prism say how many
 * coefficients per camera the tensors really hold (missing ones are 0; the
 * reference reads 6 / 4 unconditionally). */
typedef struct lfs_cameras {
    uint32_t C;
    uint32_t image_width, image_height;
    int32_t camera_model;              /* LFS_CAMERA_* */
    int32_t rs_type;                   /* LFS_SHUTTER_* */
    const float* viewmats0;            /* [C,4,4] */
    const float* viewmats1;            /* [C,4,4] or NULL */
    const float* Ks;                   /* [C,3,3] */
    const float* radial_coeffs;        /* [C,n_radial] or NULL */
    int32_t n_radial;
    const float* tangential_coeffs;    /* [C,2] or NULL */
    const float* thin_prism_coeffs;    /* [C,n_thin_prism] or NULL */
    int32_t n_thin_prism;
} lfs_cameras;

/* ---- gsplat::projection_ut_3dgs_fused  (gsplat/Ops.h:66-90, Projection.cpp:22-110,
 *      ProjectionUT3DGSFused.cu:17-203). radii int32 [C,N,2]; means2d [C,N,2];
 *      depths [C,N]; conics [C,N,3]; compensations [C,N] or NULL. Entries of culled
 *      Gaussians: radii = 0, the other outputs are written as 0 (the reference leaves
 *      them uninitialised). opacities [N] may be NULL. */
LFS_API int lfs_projection_ut_3dgs_fused(
    uint32_t N, const float* means, const float* quats, const float* scales, const float* opacities,
    const lfs_cameras* cams, float eps2d, float near_plane, float far_plane, float radius_clip,
    const lfs_ut_params* ut_params,
    int32_t* radii, float* means2d, float* depths, float* conics, float* compensations,
    lfs_stream_t stream);

/* ---- gsplat::spherical_harmonics_fwd / _bwd (gsplat/Ops.h:12-25, SphericalHarmonics.cpp:15-76,
 *      SphericalHarmonicsCUDA.cu). n = number of directions, coeffs [n,K,3], masks bool[n] or NULL.
 *      fwd: colors [n,3]; masked-out rows are written as 0.
 *      bwd: v_coeffs [n,K,3] is FULLY written (zeros where masked / unused bases: no pre-zeroing
 *      needed); v_dirs [n,3] or NULL, fully written. */
LFS_API int lfs_spherical_harmonics_fwd(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* dirs, const float* coeffs, const uint8_t* masks,
    float* colors, lfs_stream_t stream);
LFS_API int lfs_spherical_harmonics_bwd(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* dirs, const float* coeffs, const uint8_t* masks,
    const float* v_colors, float* v_coeffs, float* v_dirs, lfs_stream_t stream);

/* ---- gsplat::intersect_tile (gsplat/Ops.h:27-38, Intersect.cpp:15-122, IntersectTile.cu).
 *      Two calls with one host read of *n_isects in between (the reference syncs at the same
 *      place, Intersect.cpp:76). Non-packed layout only: means2d [C,N,2], radii int32 [C,N,2],
 *      depths [C,N].
 *   1) lfs_intersect_tile_count: tiles_per_gauss int32 [C,N]; *n_isects (device int64).
 *   2) lfs_intersect_tile_emit : isect_ids int64 [n_isects], flatten_ids int32 [n_isects];
 *      sort != 0 -> ordered exactly as a stable ascending sort of isect_ids
 *      (key = cam << (32+tile_n_bits) | tile << 32 | depth bits). tile_offsets (optional,
 *      int32 [C*tile_h*tile_w], may be NULL) receives what lfs_intersect_offset would compute,
 *      for free, when sort != 0. The SAME workspace must be passed to both calls. */
LFS_API size_t lfs_intersect_tile_workspace_bytes(uint32_t C, uint32_t N, uint32_t tile_width, uint32_t tile_height);
LFS_API int lfs_intersect_tile_count(
    uint32_t C, uint32_t N, const float* means2d, const int32_t* radii,
    uint32_t tile_size, uint32_t tile_width, uint32_t tile_height,
    int32_t* tiles_per_gauss, int64_t* n_isects, void* workspace, size_t workspace_bytes, lfs_stream_t stream);
LFS_API int lfs_intersect_tile_emit(
    uint32_t C, uint32_t N, const float* means2d, const int32_t* radii, const float* depths,
    uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, int sort, int64_t n_isects,
    const int32_t* tiles_per_gauss, int64_t* isect_ids, int32_t* flatten_ids, int32_t* tile_offsets,
    void* workspace, size_t workspace_bytes, lfs_stream_t stream);

/*      Extended forms (what the fused training step and the libtorch wrapper call; same results):
 *      _count_ex: tile_offsets (nullable, int32 [C*tile_h*tile_w]) is written by the scan kernel instead of a device copy in _emit;
 *                 flags & LFS_ISECT_COUNTERS_ZERO: the workspace is the one of the caller's previous _count_ex call with the same C, N and tile
 *                 grid (the kernels leave its counters zero) - skips the memset. n_isects and max_tile_isects (nullable: the longest tile list) may
 *                 point to pinned host memory (written by the kernel); stamp_out (nullable) then receives `stamp` LAST, behind a system-scope fence: the
 *                 host waits for the completion event and then for its own stamp before it trusts the counts.
 *      _emit_ex : scratch (nullable, int64 [n_isects]): with it the scatter runs as two binning passes with coalesced stores
 *                 (C*tile_h <= 512 and bits(C*N) + bits(tile_w) <= 32; otherwise, or without it, the one-pass scatter). max_tile_isects: the value
 *                 _count_ex reported (the per-tile sort then launches only the size classes that occur), or -1. */
#define LFS_ISECT_COUNTERS_ZERO 1u
LFS_API int lfs_intersect_tile_count_ex(
    uint32_t C, uint32_t N, const float* means2d, const int32_t* radii,
    uint32_t tile_size, uint32_t tile_width, uint32_t tile_height,
    int32_t* tiles_per_gauss, int64_t* n_isects, int64_t* max_tile_isects, int32_t* tile_offsets, uint32_t flags, int64_t* stamp_out, int64_t stamp,
    void* workspace, size_t workspace_bytes, lfs_stream_t stream);
LFS_API int lfs_intersect_tile_emit_ex(
    uint32_t C, uint32_t N, const float* means2d, const int32_t* radii, const float* depths,
    uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, int sort, int64_t n_isects,
    const int32_t* tiles_per_gauss, int64_t* isect_ids, int32_t* flatten_ids, int32_t* tile_offsets, int64_t* scratch, int64_t max_tile_isects,
    void* workspace, size_t workspace_bytes, lfs_stream_t stream);

/* ---- gsplat::intersect_offset (gsplat/Ops.h:39-43, IntersectTile.cu:206-286).
 *      offsets int32 [C,tile_h,tile_w] from SORTED isect_ids. */
LFS_API int lfs_intersect_offset(
    int64_t n_isects, const int64_t* isect_ids, uint32_t C, uint32_t tile_width, uint32_t tile_height,
    int32_t* offsets, lfs_stream_t stream);

/* ---- gsplat::rasterize_to_pixels_from_world_3dgs_fwd / _bwd (gsplat/Ops.h:92-166,
 *      Rasterization.cpp, RasterizeToPixelsFromWorld3DGS{Fwd,Bwd}.cu). channels in 1..4.
 *      colors [C,N,channels], opacities [C,N], backgrounds [C,channels] or NULL,
 *      masks bool [C,tile_h,tile_w] or NULL, tile_offsets int32 [C,tile_h,tile_w],
 *      flatten_ids int32 [n_isects] (values in [0, C*N)).
 *      fwd: render_colors [C,H,W,channels], render_alphas [C,H,W,1], last_ids int32 [C,H,W].
 *      bwd: v_means [N,3], v_quats [N,4], v_scales [N,3], v_colors [C,N,channels],
 *           v_opacities [C,N] — all FULLY written (no pre-zeroing needed). v_render_alphas may be NULL (= zeros). */
/*      workspace: camera state + one 64-B record, one 64-B gradient accumulator row and one 32-B culling
 *      record per (camera, Gaussian) + the compacted per-8x8-cell lists ((tile_size/8)^2 * n_isects * 8 B,
 *      worst case). Returns 0 for an unsupported tile_size. */
LFS_API size_t lfs_rasterize_workspace_bytes(uint32_t C, uint32_t N, uint32_t channels, uint32_t image_width,
                                             uint32_t image_height, uint32_t tile_size, int64_t n_isects);
/*      developer switch, not part of the reference API: bit 0 = build the per-cell lists without culling (fwd output must be bit-identical either way;
 *      tests/test_gpu_raster.py); bits 1 - 3: unused (the experimental kernels they selected were measured and removed);
 *      bit 4 = deterministic backward accumulation (two passes, 64-bit fixed point: run-to-run bit-identical gradients; lfs_rasterize_workspace_bytes grows while it is set);
 *      bit 5 = the one-pass intersection scatter even when a scratch array is given; bit 6 = the training step packs the rasterizer's records with the separate
 *      raster_pack pass of rounds 1 - 3 instead of inside the projection kernel (A/B, tests/test_emulated_step_pack.py); bit 7: unused (round 6: the projection kernel clearing the
 *      backward's accumulator rows instead of the memset - measured slower, removed). */
LFS_API void lfs_set_debug_flags(uint32_t flags);
LFS_API uint32_t lfs_get_debug_flags(void);
LFS_API int lfs_rasterize_to_pixels_from_world_3dgs_fwd(
    uint32_t N, uint32_t channels, const float* means, const float* quats, const float* scales,
    const float* colors, const float* opacities, const float* backgrounds, const uint8_t* masks,
    const lfs_cameras* cams, uint32_t tile_size, const lfs_ut_params* ut_params,
    const int32_t* tile_offsets, const int32_t* flatten_ids, int64_t n_isects,
    float* render_colors, float* render_alphas, int32_t* last_ids,
    void* workspace, size_t workspace_bytes, lfs_stream_t stream);
LFS_API int lfs_rasterize_to_pixels_from_world_3dgs_bwd(
    uint32_t N, uint32_t channels, const float* means, const float* quats, const float* scales,
    const float* colors, const float* opacities, const float* backgrounds, const uint8_t* masks,
    const lfs_cameras* cams, uint32_t tile_size, const lfs_ut_params* ut_params,
    const int32_t* tile_offsets, const int32_t* flatten_ids, int64_t n_isects,
    const float* render_alphas, const int32_t* last_ids,
    const float* v_render_colors, const float* v_render_alphas,
    float* v_means, float* v_quats, float* v_scales, float* v_colors, float* v_opacities,
    void* workspace, size_t workspace_bytes, lfs_stream_t stream);

/*      extension (not in Ops.h): the same backward, for a caller that still owns the workspace the matching
 *      forward call filled (same inputs, nothing else ran on that workspace in between): skips rebuilding
 *      the records and the per-cell lists. Used by the autograd Function of rasterizer.py. */
LFS_API int lfs_rasterize_to_pixels_from_world_3dgs_bwd_prepared(
    uint32_t N, uint32_t channels, const float* means, const float* quats, const float* scales,
    const float* colors, const float* opacities, const float* backgrounds, const uint8_t* masks,
    const lfs_cameras* cams, uint32_t tile_size, const lfs_ut_params* ut_params,
    const int32_t* tile_offsets, const int32_t* flatten_ids, int64_t n_isects,
    const float* render_alphas, const int32_t* last_ids,
    const float* v_render_colors, const float* v_render_alphas,
    float* v_means, float* v_quats, float* v_scales, float* v_colors, float* v_opacities,
    void* workspace, size_t workspace_bytes, lfs_stream_t stream);
/* ... with the clamped MSE of lfs_mse_loss_fwd_bwd folded into the kernel's prologue (extension: one camera, 3 channels, no masks):
 * dL/d(render) is derived from render_colors [H,W,3] and target_chw [3,H,W] in registers and never stored; *loss += weight * mse. */
LFS_API int lfs_rasterize_to_pixels_from_world_3dgs_bwd_prepared_mse(
    uint32_t N, const float* means, const float* quats, const float* scales, const float* colors, const float* opacities,
    const float* backgrounds, const lfs_cameras* cams, uint32_t tile_size, const int32_t* tile_offsets, const int32_t* flatten_ids,
    int64_t n_isects, const float* render_colors, const float* render_alphas, const int32_t* last_ids, const float* target_chw, float weight,
    float* loss, float* v_means, float* v_quats, float* v_scales, float* v_colors, float* v_opacities, void* workspace, size_t workspace_bytes,
    lfs_stream_t stream);

/* ---- Fused L2 extensions (not in Ops.h; lichtfeld-studio_amd/fused.py, trainer.py): the libtorch element-wise
 *      work rasterizer.cpp:200-263 / splat_data.cpp:267-286 / the trainer's loss wrap around the operators,
 *      as single passes. Same arithmetic as the op-by-op path (tests/test_gpu_fused.py).
 *   lfs_sh_model_fwd : colors [N,3] = clamp_min(SH(normalize(means - campos(viewmat)), cat(sh0, shN)) + 0.5, 0)
 *                      with mask = all(radii > 0); viewmat = ONE rigid row-major [4,4] on the device.
 *   lfs_sh_model_bwd : v_colors = dL/dcolors (the clamp is applied inside from `colors`); v_sh0 [N,1,3], v_shN [N,K-1,3]
 *                      written (accumulate = 0) or added to (accumulate != 0); v_means [N,3] += dL/d(dirs).
 *   lfs_activations_fwd / _bwd : quats = normalize(raw), scales = exp(raw), opacities = sigmoid(raw) and the vjp.
 *   lfs_mse_loss_fwd_bwd : *loss += weight * mean((clamp(render,0,1) - target)^2), v_render = d/d(render);
 *                      render / v_render HWC [H,W,3], target CHW [3,H,W]. */
LFS_API int lfs_sh_model_fwd(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* means, const float* viewmat, const float* sh0, const float* shN,
    const int32_t* radii, float* colors, lfs_stream_t stream);
LFS_API int lfs_sh_model_bwd(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* means, const float* viewmat, const float* sh0, const float* shN,
    const int32_t* radii, const float* colors, const float* v_colors, int accumulate,
    float* v_sh0, float* v_shN, float* v_means, lfs_stream_t stream);
/* lfs_sh_model_bwd for a step with ONE view, fused with the optimizer: the gradient of shN is not stored but consumed by the
 * Adam update (fast_gs::optimizer::adam_step arithmetic, adam_kernels.cuh:13-36) of shN / its moments in place; v_sh0 written,
 * v_means += dL/d(dirs) computed from the pre-update shN. Element-wise identical to lfs_sh_model_bwd + lfs_adam_step on shN. */
LFS_API int lfs_sh_model_bwd_adam(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* means, const float* viewmat, const float* sh0, float* shN,
    const int32_t* radii, const float* colors, const float* v_colors, float* v_sh0, float* v_means,
    float* shN_exp_avg, float* shN_exp_avg_sq, float lr, float beta1, float beta2, float eps, float bias_correction1_rcp,
    float bias_correction2_sqrt_rcp, lfs_stream_t stream);
/* Multi-view forms for SH-sharded data parallelism (the owner of n Gaussians evaluates the views of all ranks in ONE launch; coefficient
 * rows are read once for all views): viewmats [V,4,4]; radii [V,view_stride,2], colors / v_colors [V,view_stride,3] with the first n rows of
 * every view used. Backward: the coefficient gradient is summed over the views, then written / added (accumulate) to v_sh0, v_shN, or -
 * shN_exp_avg != NULL, accumulate == 0 - consumed by shN's Adam update as in lfs_sh_model_bwd_adam; v_means += sum over views of dL/d(dirs). */
LFS_API int lfs_sh_model_fwd_views(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, uint32_t n_views, uint32_t view_stride, const float* means, const float* viewmats,
    const float* sh0, const float* shN, const int32_t* radii, float* colors, lfs_stream_t stream);
LFS_API int lfs_sh_model_bwd_views(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, uint32_t n_views, uint32_t view_stride, const float* means, const float* viewmats,
    const float* sh0, float* shN, const int32_t* radii, const float* colors, const float* v_colors, int accumulate,
    float* v_sh0, float* v_shN, float* v_means, float* shN_exp_avg, float* shN_exp_avg_sq, float lr, float beta1, float beta2, float eps,
    float bias_correction1_rcp, float bias_correction2_sqrt_rcp, lfs_stream_t stream);
LFS_API int lfs_activations_fwd(uint32_t N, const float* raw_quats, const float* raw_scales, const float* raw_opacities,
                                float* quats, float* scales, float* opacities, lfs_stream_t stream);
LFS_API int lfs_activations_bwd(uint32_t N, const float* raw_quats, const float* scales, const float* opacities,
                                const float* v_quats, const float* v_scales, const float* v_opacities,
                                float scale_reg, float opacity_reg, /* + d/d(activated) of scale_reg * mean(scales) + opacity_reg * mean(opacities), trainer.cpp:132-158 */
                                int accumulate,
                                float* g_raw_quats, float* g_raw_scales, float* g_raw_opacities, lfs_stream_t stream);
LFS_API int lfs_mse_loss_fwd_bwd(uint32_t H, uint32_t W, const float* render_hwc, const float* target_chw, float weight,
                                 float* v_render_hwc, float* loss, lfs_stream_t stream);
/*      the same for a CHW render that is NOT clamped (the fastgs path, fast_rasterizer.cpp:62-66) */
LFS_API int lfs_mse_loss_chw_fwd_bwd(uint32_t H, uint32_t W, const float* render_chw, const float* target_chw, float weight,
                                     float* v_render_chw, float* loss, lfs_stream_t stream);

/* ---- "next" row 1 of SURVEY.md §8f: the reference's default training rasterizer (fastgs, EWA splatting).
 *      Replaces fast_gs::rasterization::forward_wrapper / backward_wrapper (fastgs/rasterization/include/rasterization_api.h:27-75,
 *      src/rasterization_api.cu, forward.cu, backward.cu, kernels_{forward,backward}.cuh). Inputs are the RAW parameters
 *      (log-scales, un-normalised quaternions wxyz, opacity logits, sh0 [N,1,3], sh_rest [N,total_bases_sh_rest,3]); w2c is a
 *      row-major [4,4] (rows 0-2 used) and cam_position [3], both on the device. image [3,H,W], alpha [1,H,W] (CHW, no background).
 *      The reference's four opaque buffer tensors become two caller-owned workspaces that must survive until the backward:
 *        1) lfs_fastgs_preprocess   : per-primitive stage + per-tile counts; *n_instances (device int64) = list length
 *        2) (host reads n_instances: the reference syncs at the same place, forward.cu:114-117)
 *        3) lfs_fastgs_render       : instance lists (scatter + per-tile depth sort), per-cell culling, blending
 *        4) lfs_fastgs_backward     : blending backward + per-primitive backward; gradients FULLY written; sh_coefficients_0 may be
 *                                     NULL (the reference's backward does not take it either); densification_info
 *                                     [2,N] (or NULL) is accumulated into (visibility count, screen-space gradient norm). */
LFS_API size_t lfs_fastgs_primitive_workspace_bytes(uint32_t N, uint32_t width, uint32_t height);
LFS_API size_t lfs_fastgs_instance_workspace_bytes(uint32_t width, uint32_t height, int64_t n_instances);
LFS_API int lfs_fastgs_preprocess(
    uint32_t N, const float* means, const float* scales_raw, const float* rotations_raw, const float* opacities_raw,
    const float* sh_coefficients_0, const float* sh_coefficients_rest, uint32_t total_bases_sh_rest, const float* w2c, const float* cam_position,
    uint32_t active_sh_bases, uint32_t width, uint32_t height, float fx, float fy, float cx, float cy, float near_plane, float far_plane,
    int64_t* n_instances, void* primitive_workspace, size_t primitive_workspace_bytes, lfs_stream_t stream);
/* n_instances of this process's last lfs_fastgs_preprocess call, read back asynchronously: blocks until the 8-byte copy (queued before the SH kernel)
 * has landed, not until the stream is idle. Alternative to reading the device value with a stream synchronisation. */
LFS_API int lfs_fastgs_wait_n_instances(int64_t* n_instances);
LFS_API int lfs_fastgs_render(
    uint32_t N, uint32_t width, uint32_t height, int64_t n_instances, void* primitive_workspace, size_t primitive_workspace_bytes,
    void* instance_workspace, size_t instance_workspace_bytes, float* image, float* alpha, lfs_stream_t stream);
LFS_API int lfs_fastgs_backward(
    uint32_t N, const float* means, const float* scales_raw, const float* rotations_raw, const float* sh_coefficients_0, const float* sh_coefficients_rest,
    uint32_t total_bases_sh_rest, const float* w2c, const float* cam_position, uint32_t active_sh_bases, uint32_t width, uint32_t height, float fx, float fy,
    float cx, float cy, float near_plane, float far_plane, int64_t n_instances, void* primitive_workspace, size_t primitive_workspace_bytes,
    void* instance_workspace, size_t instance_workspace_bytes, const float* grad_image, const float* grad_alpha, const float* alpha,
    float* densification_info, float* grad_means, float* grad_scales_raw, float* grad_rotations_raw, float* grad_opacities_raw,
    float* grad_sh_coefficients_0, float* grad_sh_coefficients_rest, lfs_stream_t stream);
/* lfs_fastgs_backward for a step with ONE view, fused with the optimizer (extension): sh_coefficients_rest and its Adam moments are updated in
 * place by the SH backward (fast_gs::optimizer::adam_step arithmetic); the [N,total_rest,3] gradient is never stored. */
LFS_API int lfs_fastgs_backward_adam(
    uint32_t N, const float* means, const float* scales_raw, const float* rotations_raw, const float* sh_coefficients_0, float* sh_coefficients_rest,
    uint32_t total_bases_sh_rest, const float* w2c, const float* cam_position, uint32_t active_sh_bases, uint32_t width, uint32_t height, float fx, float fy,
    float cx, float cy, float near_plane, float far_plane, int64_t n_instances, void* primitive_workspace, size_t primitive_workspace_bytes,
    void* instance_workspace, size_t instance_workspace_bytes, const float* grad_image, const float* grad_alpha, const float* alpha,
    float* densification_info, float* grad_means, float* grad_scales_raw, float* grad_rotations_raw, float* grad_opacities_raw,
    float* grad_sh_coefficients_0, float* sh_rest_exp_avg, float* sh_rest_exp_avg_sq, float lr, float beta1, float beta2, float eps,
    float bias_correction1_rcp, float bias_correction2_sqrt_rcp, lfs_stream_t stream);
LFS_API void lfs_fastgs_set_debug_flags(uint32_t flags); /* bit 0: no per-cell culling (bit-identity test) */

/* ---- "next" row 2 of SURVEY.md §8f: fused SSIM (fusedssim / fusedssim_backward, include/kernels/ssim.cuh:11-30,
 *      src/training/kernels/ssim.cu:64-510) and the trainer's photometric loss (trainer.cpp:122-125).
 *      img1 / img2 / maps [B,CH,H,W]; 11-tap Gaussian window, zero padding. dm_* may all be NULL (train == false).
 *      lfs_photometric_loss_fwd_bwd (extension): *loss += weight * ((1 - lambda) * L1 + lambda * (1 - mean SSIM over the
 *      "valid" crop)) of clamp(render, 0, 1) against target; render / v_render HWC [H,W,3], target CHW [3,H,W]. */
LFS_API int lfs_fused_ssim_fwd(uint32_t B, uint32_t CH, uint32_t H, uint32_t W, float C1, float C2, const float* img1, const float* img2,
                               float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, lfs_stream_t stream);
LFS_API int lfs_fused_ssim_bwd(uint32_t B, uint32_t CH, uint32_t H, uint32_t W, float C1, float C2, const float* img1, const float* img2,
                               const float* dL_dmap, const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                               float* dL_dimg1, lfs_stream_t stream);
LFS_API size_t lfs_photometric_loss_workspace_bytes(uint32_t H, uint32_t W);
LFS_API int lfs_photometric_loss_fwd_bwd(uint32_t H, uint32_t W, const float* render_hwc, const float* target_chw, float lambda_dssim,
                                         float weight, float* v_render_hwc, float* loss, void* workspace, size_t workspace_bytes,
                                         lfs_stream_t stream);
/* ... and for a CHW render [3,H,W] (the fastgs rasterizer's image layout), which is NOT clamped (fast_rasterizer.cpp:63 hands
 * the image to the loss as is); v_render_chw [3,H,W] */
LFS_API int lfs_photometric_loss_chw_fwd_bwd(uint32_t H, uint32_t W, const float* render_chw, const float* target_chw, float lambda_dssim,
                                             float weight, float* v_render_chw, float* loss, void* workspace, size_t workspace_bytes,
                                             lfs_stream_t stream);

/* general forms of the two losses: render / v_render in either layout (render_is_chw), clamped on the way in or not (an image that went through the
 * bilateral grid reaches the loss un-clamped, trainer.cpp:662-676) */
LFS_API int lfs_photometric_loss_ex_fwd_bwd(uint32_t H, uint32_t W, const float* render, uint32_t render_is_chw, uint32_t clamp_render, const float* target_chw,
                                            float lambda_dssim, float weight, float* v_render, float* loss, void* workspace, size_t workspace_bytes,
                                            lfs_stream_t stream);
LFS_API int lfs_mse_loss_ex_fwd_bwd(uint32_t H, uint32_t W, const float* render, uint32_t render_is_chw, uint32_t clamp_render, const float* target_chw, float weight,
                                    float* v_render, float* loss, lfs_stream_t stream);

/* ---- bilateral-grid appearance model (row 2 of §8f, BASELINE config 5): gs::bilateral_grid::slice_forward_cuda /
 *      slice_backward_cuda / tv_loss_forward_cuda / tv_loss_backward_cuda (include/kernels/bilateral_grid.cuh:12-33,
 *      src/training/kernels/bilateral_grid_{forward,backward,tv}.cu). grid [12,L,H,W]; image h x w (both >= 2), uniform
 *      x/y coordinates, guidance z = luma. Extensions: chw != 0 reads / writes the image planes as [3,h,w] instead of
 *      [h,w,3]; clamp_input != 0 folds BilateralGrid::apply's clamp(rgb, 0, 1) (components/bilateral_grid.cpp:115) into
 *      both passes. grad_grid is ACCUMULATED into (the reference zero-fills, then atomically adds: zero it for the same
 *      result); grad_rgb is fully written. TV: grids [N,12,L,H,W]; *loss += weight * tv; the backward takes dL/dloss as a
 *      host float (the reference reads it with .item(), bilateral_grid_tv.cu:181). */
LFS_API int lfs_bilateral_slice_fwd(uint32_t L, uint32_t H, uint32_t W, uint32_t h, uint32_t w, const float* grid, const float* rgb,
                                    uint32_t chw, uint32_t clamp_input, float* output, lfs_stream_t stream);
LFS_API int lfs_bilateral_slice_bwd(uint32_t L, uint32_t H, uint32_t W, uint32_t h, uint32_t w, const float* grid, const float* rgb,
                                    const float* grad_output, uint32_t chw, uint32_t clamp_input, float* grad_grid, float* grad_rgb,
                                    lfs_stream_t stream);
LFS_API int lfs_bilateral_tv_loss_fwd(uint32_t N, uint32_t L, uint32_t H, uint32_t W, const float* grids, float weight, float* loss,
                                      lfs_stream_t stream);
LFS_API int lfs_bilateral_tv_loss_bwd(uint32_t N, uint32_t L, uint32_t H, uint32_t W, const float* grids, float grad_output, uint32_t accumulate,
                                      float* grad_grids, lfs_stream_t stream);

/* ---- data side (row 4 of §8f; the host formats are in lfs_io.h):
 *      lfs_image_u8_to_chw_f32: Camera::load_and_get_image (src/core/camera.cpp:101-140) fused with load_image's downscale
 *      (src/core/image_io.cpp:33-57, OpenImageIO resample(interpolate = true)): src u8 [sh,sw,3] -> dst f32 [3,dh,dw] in [0,1];
 *      when the sizes differ the image is resampled bilinearly and re-quantised to 8 bit first, as the reference's host path does.
 *      lfs_mean_neighbor_distances: compute_mean_neighbor_distances (src/core/splat_data.cpp:64-111), points [N,3] -> out [N], device pointers. The
 *      reference's nanoflann query is (1 + 10)-approximate (SearchParameters(10) sets eps, :97), so its result depends on the kd-tree: the entry copies the
 *      points to the host, builds nanoflann's tree there (single-threaded divideTree, leaf size 10), walks it on the GPU in searchLevel's order and
 *      synchronises the stream before it returns. Bit-identical to the reference's function. LFS_E_UNSUPPORTED if the tree is deeper than 128 levels.
 *      lfs_mean_neighbor_distances_exact (extension): the same quantity from an exact all-pairs search; asynchronous. */
LFS_API int lfs_image_u8_to_chw_f32(const uint8_t* src_hwc, uint32_t src_width, uint32_t src_height, float* dst_chw, uint32_t dst_width,
                                    uint32_t dst_height, lfs_stream_t stream);
LFS_API int lfs_mean_neighbor_distances(uint32_t N, const float* points, float* out, lfs_stream_t stream);
LFS_API int lfs_mean_neighbor_distances_exact(uint32_t N, const float* points, float* out, lfs_stream_t stream);

/* ---- gsplat::quats_to_rotmats (gsplat/Ops.h:45-48, QuatToRotmatCUDA.cu:14-39): rotmats [N,3,3] row-major */
LFS_API int lfs_quats_to_rotmats(uint32_t N, const float* quats, float* rotmats, lfs_stream_t stream);

/* ---- gsplat::relocation (gsplat/Ops.h:50-57, RelocationCUDA.cu:12-43): ratios int32 [N], binoms [n_max,n_max] */
LFS_API int lfs_relocation(
    uint32_t N, const float* opacities, const float* scales, const int32_t* ratios, const float* binoms, int32_t n_max,
    float* new_opacities, float* new_scales, lfs_stream_t stream);

/* ---- gsplat::add_noise (gsplat/Ops.h:59-65, RelocationCUDA.cu:88-144): means updated in place */
LFS_API int lfs_add_noise(
    uint32_t N, const float* raw_opacities, const float* raw_scales, const float* raw_quats, const float* noise,
    float* means, float current_lr, lfs_stream_t stream);

/* Extension: the front half of the fused 3DGUT training step (one camera, global shutter, 3 channels).
 *   lfs_activations_project_ut : lfs_activations_fwd + lfs_projection_ut_3dgs_fused in one pass over the raw parameters (normalize / exp / sigmoid, then the
 *       projection on the activated values - the same operations in the same order); quats / scales / opacities receive the activated values. */
LFS_API int lfs_activations_project_ut(
    uint32_t N, const float* means, const float* raw_quats, const float* raw_scales, const float* raw_opacities, const lfs_cameras* cams,
    float eps2d, float near_plane, float far_plane, float radius_clip, const lfs_ut_params* ut_params,
    float* quats, float* scales, float* opacities, int32_t* radii, float* means2d, float* depths, lfs_stream_t stream);

/* ---- The --gut training step as ONE host call (csrc/gut_step.hip; replaces Trainer::train_step -> rasterize() -> the Ops.h sequence of
 *      src/training/trainer.cpp:579-770 / rasterization/rasterizer.cpp:200-344 for one pinhole, global-shutter camera per step, RGB, MSE, fused Adam).
 *      No host read on the critical path: the intersection lists are sized for `capacity` entries and sorted by the size classes that cover
 *      `assumed_longest` entries per tile; the true count stays on the device. When it exceeds either assumption the kernels raise a device flag,
 *      empty all lists and the Adam kernels return without updating anything: the caller learns it from the counts - written to PINNED host memory
 *      host_counts[3] = {n_isects, longest tile list, stamp} early in the step - after it has enqueued everything (lfs_gut_step_wait, lfs_gut_step_fits),
 *      enlarges the workspace and calls again with the same arguments.
 *   lfs_gut_step_args: parameters are updated in place; exp_avg / exp_avg_sq / adam[k] in FusedAdam's group order (fused_adam.cpp:22-95, strategy_utils.cpp:35-40)
 *      means, sh0, shN, raw_scales, raw_quats, raw_opacities; adam[k] = {lr, beta1, beta2, eps, 1/(1-beta1^t), 1/sqrt(1-beta2^t)} (fused_adam.cpp:78-92).
 *      viewmat [4,4], Kmat [3,3], background [3] (nullable), target_chw [3,H,W] on the device; *loss = loss_weight * mse(clamp(render,0,1), target) (stored).
 *   workspace: lfs_gut_step_layout_for(...).bytes; the layout names the byte offsets of what a caller may want to look at afterwards (the render
 *      [H,W,3], alpha [H,W], radii int32 [N,2], ...). tile_offsets has T + 1 entries (the last one is n_isects).
 *   lfs_gut_view_forward / lfs_gut_view_backward: the same step split for callers that need GRADIENT TENSORS (data-parallel ranks, several views per step,
 *      losses other than MSE): forward into the workspace, then backward into grads[6] (group order; written or, accumulate != 0, added to). */
typedef struct lfs_gut_step_args {
    uint32_t N, K, sh_degree, image_width, image_height, tile_size;
    float *means, *sh0, *shN, *raw_scales, *raw_quats, *raw_opacities;
    float* exp_avg[6]; float* exp_avg_sq[6];
    float adam[6][6];
    const float* viewmat; const float* Kmat; const float* background; const float* target_chw;
    float loss_weight, scale_reg, opacity_reg;
    float* loss;
} lfs_gut_step_args;
typedef struct lfs_gut_step_layout {
    size_t bytes, render, alpha, last_ids, radii, means2d, depths, colors, quats, scales, opacities, tile_offsets, flatten_ids, isect_ids, counts, abort_flag;
} lfs_gut_step_layout;
LFS_API int lfs_gut_step_layout_for(uint32_t N, uint32_t image_width, uint32_t image_height, uint32_t tile_size, int64_t capacity, lfs_gut_step_layout* out);
LFS_API int lfs_gut_step_fits(int64_t n_isects, int64_t longest, int64_t capacity, int64_t assumed_longest); /* 1: the attempt with these assumptions was valid */
LFS_API int lfs_gut_step_supported(uint32_t N, uint32_t image_width, uint32_t image_height, uint32_t tile_size); /* 0: this shape is outside the speculative step (more than
    512 tile rows, index-bit limit, debug bit 5): lfs_gut_* return LFS_E_UNSUPPORTED - enqueue the operators one by one instead (trainer.py does) */
LFS_API int lfs_gut_train_step(const lfs_gut_step_args* args, int64_t capacity, int64_t assumed_longest, void* workspace, size_t workspace_bytes,
                               int64_t* host_counts /* pinned [3], or NULL: counts stay in the workspace */, int64_t stamp, lfs_stream_t stream);
LFS_API int lfs_gut_view_forward(const lfs_gut_step_args* args, int64_t capacity, int64_t assumed_longest, void* workspace, size_t workspace_bytes,
                                 int64_t* host_counts, int64_t stamp, lfs_stream_t stream);
LFS_API int lfs_gut_view_backward(const lfs_gut_step_args* args, int64_t capacity, const float* v_render /* [H,W,3], used when args->target_chw == NULL */,
                                  float* const* grads /* [6] host */, int accumulate, void* workspace, size_t workspace_bytes, lfs_stream_t stream);
/*   = lfs_gut_view_backward_sh (rasterizer backward + SH backward: grads[1], grads[2] final) then lfs_gut_view_backward_finish (grads[0], grads[3..5]); a
 *   data-parallel caller starts the all-reduce of the SH gradients between the two (dist.GradBucket.all_reduce_early). With args->exp_avg[2] /
 *   exp_avg_sq[2] / adam[2] set (one view per step, accumulate == 0) lfs_gut_view_backward_sh applies shN's Adam update itself, as lfs_sh_model_bwd_adam
 *   does, and grads[2] is neither read nor written: the steps whose loss is not the folded MSE (L1 + D-SSIM, bilateral grid, MCMC) keep the 45-of-59-float
 *   tensor out of the optimizer launch that way. */
LFS_API int lfs_gut_view_backward_sh(const lfs_gut_step_args* args, int64_t capacity, const float* v_render, float* const* grads, int accumulate,
                                     void* workspace, size_t workspace_bytes, lfs_stream_t stream);
LFS_API int lfs_gut_view_backward_finish(const lfs_gut_step_args* args, int64_t capacity, float* const* grads, int accumulate,
                                         void* workspace, size_t workspace_bytes, lfs_stream_t stream);
/*   lfs_gut_view_backward_rows: the view's backward WITHOUT the SH backward - grads[0] (means, no SH direction term), grads[3..5] and dL/dcolour rows [N,3] ->
 *   v_colors_out: the factored gradient exchange of the replicated data-parallel layout (dist.ColorGradExchange) gathers the rows of all ranks and runs
 *   lfs_sh_model_bwd_views over all views on every rank (radii = colors = NULL there: rows pre-masked with colour > 0 by the rank that rendered the view). */
LFS_API int lfs_gut_view_backward_rows(const lfs_gut_step_args* args, int64_t capacity, const float* v_render, float* const* grads /* [6] host */, int accumulate,
                                       float* v_colors_out, void* workspace, size_t workspace_bytes, lfs_stream_t stream);
LFS_API int lfs_gut_step_wait(const int64_t* host_counts, int64_t stamp, double timeout_s, int64_t* n_isects, int64_t* longest);
/*   lfs_gut_train_step_ex (round 6): lfs_gut_train_step with its three per-Gaussian tail passes (SH backward + Adam on sh0 / shN, finish + Adam on the other four tensors,
 *      and the NEXT step's SH colours) as ONE launch. next_viewmat (device [4,4], nullable): the view the next step will render - its SH colours are then written to the
 *      workspace's colours for every Gaussian, from the coefficient rows as they leave their Adam update. colors_ready != 0: the caller's statement that the previous call on
 *      this workspace was this entry point with next_viewmat naming the matrix args->viewmat holds now, same N / K / sh_degree, that it fitted its buffers, and that nothing
 *      has written means / sh0 / shN since - the SH colour kernel is then not launched. Same results as lfs_gut_train_step. */
LFS_API int lfs_gut_train_step_ex(const lfs_gut_step_args* args, const float* next_viewmat, int colors_ready, int64_t capacity, int64_t assumed_longest, void* workspace,
                                  size_t workspace_bytes, int64_t* host_counts, int64_t stamp, lfs_stream_t stream);
/*   lfs_gut_train_step_pipelined (round 6): lfs_gut_train_step - same arguments, same results - with the step's HBM-bound SH kernels on a side stream of the library,
 *      UNDER the latency- / VALU-bound front end of the NEXT step: projection(k+1) (records without colours) | tile lists | culling run on `stream` while SH Adam(k)
 *      (1.1 GB of read-modify-write at 1 M Gaussians) and the SH colours(k+1) run beside them; `stream` waits for the colours in front of the forward kernel. The SH
 *      backward is split for it: a direction pass on `stream` (dL/d(dirs) + a 32-byte hand-over row per Gaussian) and the Adam pass on the side stream.
 *      CONTRACT: between two pipelined calls sh0, shN and their Adam moments belong to the side stream - lfs_gut_pipeline_join(stream) makes `stream` wait (device-side,
 *      no host wait) for the last update before anything else touches them (every other lfs_gut_* entry point joins by itself). All other tensors stay ordered on `stream`.
 *   lfs_gut_pipeline_join: 1 = there was a pending update and `stream` now waits for it, 0 = nothing pending, < 0 = error. */
LFS_API int lfs_gut_train_step_pipelined(const lfs_gut_step_args* args, int64_t capacity, int64_t assumed_longest, void* workspace, size_t workspace_bytes,
                                         int64_t* host_counts, int64_t stamp, lfs_stream_t stream);
LFS_API int lfs_gut_pipeline_join(lfs_stream_t stream);

/* Extension: the all-inline training step of the fused 3DGUT path (one camera, global shutter, 3 channels, ONE view per step on one rank - the
 * reference's training configuration). No parameter gradient is materialised:
 *   lfs_rasterize_to_pixels_from_world_3dgs_bwd_prepared_mse_acc : ..._bwd_prepared_mse without its last kernel - the per-Gaussian sums (rows of 16
 *       floats: dL/dA 9 | -dL/dg 3 | dL/dopacity | dL/dcolour 3) stay in the workspace at lfs_rasterize_workspace_acc_offset(C, N);
 *   lfs_sh_model_bwd_adam_all : SH backward reading dL/dcolour from those rows, Adam on sh0 AND shN in place, dL/d(dirs) written to v_dirs [N,3];
 *   lfs_gut_finish_adam       : rows -> dL/d(means, quats, scales, opacity) (raster_finish) -> raw-parameter gradients (normalize / exp / sigmoid vjp +
 *       the regularisers) -> Adam on means, raw_scales, raw_quats, raw_opacities, in one pass; *loss = the fused MSE (stored, not added).
 * Element for element the operations of the separate kernels (lfs_..._bwd_prepared_mse, lfs_sh_model_bwd_adam, lfs_activations_bwd, lfs_adam_step_multi). */
LFS_API size_t lfs_rasterize_workspace_acc_offset(uint32_t C, uint32_t N);
LFS_API int lfs_rasterize_to_pixels_from_world_3dgs_bwd_prepared_mse_acc(
    uint32_t N, const float* means, const float* quats, const float* scales, const float* colors, const float* opacities,
    const float* backgrounds, const lfs_cameras* cams, uint32_t tile_size, const int32_t* tile_offsets, const int32_t* flatten_ids,
    int64_t n_isects, const float* render_colors, const float* render_alphas, const int32_t* last_ids, const float* target_chw, float weight,
    void* workspace, size_t workspace_bytes, lfs_stream_t stream);
LFS_API int lfs_sh_model_bwd_adam_all(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* means, const float* viewmat, float* sh0, float* shN,
    const int32_t* radii, const float* colors, const float* acc_rows, float* v_dirs /* [n,3], written */,
    float* sh0_exp_avg, float* sh0_exp_avg_sq, const float* sh0_scalars /* host: lr, beta1, beta2, eps, bc1_rcp, bc2_sqrt_rcp */,
    float* shN_exp_avg, float* shN_exp_avg_sq, const float* shN_scalars, lfs_stream_t stream);
/* For steps that need the gradient tensors (multi-GPU all-reduce, several views per step, any loss): lfs_..._bwd_prepared_acc = lfs_..._bwd_prepared without
 * its last kernel (3 channels, one camera, no masks), then lfs_gut_finish_grads turns the accumulator rows into dL/d(means [rasterizer part], raw_scales,
 * raw_quats, raw_opacities) - written, or added to when accumulate != 0 - and dL/dcolour [N,3] in ONE pass (raster_finish + lfs_activations_bwd + the copy of
 * dL/dmeans; the regularisers as in lfs_activations_bwd); loss (nullable): += the fused MSE of lfs_..._bwd_prepared_mse_acc. */
LFS_API int lfs_rasterize_to_pixels_from_world_3dgs_bwd_prepared_acc(
    uint32_t N, const float* means, const float* quats, const float* scales, const float* colors, const float* opacities,
    const float* backgrounds, const lfs_cameras* cams, uint32_t tile_size, const int32_t* tile_offsets, const int32_t* flatten_ids,
    int64_t n_isects, const float* render_alphas, const int32_t* last_ids, const float* v_render_colors, const float* v_render_alphas,
    void* workspace, size_t workspace_bytes, lfs_stream_t stream);
LFS_API int lfs_gut_finish_grads(
    uint32_t N, const float* means, const float* raw_quats, const float* quats, const float* scales, const float* opacities, float scale_reg, float opacity_reg,
    int accumulate, float* g_means, float* g_raw_scales, float* g_raw_quats, float* g_raw_opacities, float* v_colors, float* loss,
    void* workspace, size_t workspace_bytes, lfs_stream_t stream);
LFS_API int lfs_gut_finish_adam(
    uint32_t N, float* means, float* raw_scales, float* raw_quats, float* raw_opacities, const float* quats, const float* scales, const float* opacities,
    const float* v_dirs /* [N,3] from lfs_sh_model_bwd_adam_all */, float* const* exp_avg /* [4] host: means, raw_scales, raw_quats, raw_opacities */,
    float* const* exp_avg_sq, const float* scalars /* [4][6] host */, float scale_reg, float opacity_reg, float* loss, void* workspace, size_t workspace_bytes,
    lfs_stream_t stream);

/* Extension ("device-side index ops with no host syncs"): MCMC::relocate_gs of the reference
 * (src/training/strategies/mcmc.cpp:113-194) as ONE enqueue without a host round trip. The reference finds the dead Gaussians with nonzero()
 * (a device->host sync for the count), draws as many sources from the alive ones with torch::multinomial(opacity), calls gsplat::relocation on
 * them, copies every parameter row source -> dead and zeroes the sources' Adam moments. Here, all on `stream`:
 *   dead_i   = sigmoid(raw_opacities_i) <= min_opacity  ||  |raw_quats_i|^2 < 1e-8
 *   source_i = inverse-CDF sample over the alive opacities with the caller's uniform number uniforms[i] in [0,1) (dead i only; fp64 prefix sums in a
 *              fixed order: the same uniforms give the same sources on every rank), count_j = how often j was drawn
 *   for every drawn j: (opacity, scale)_j <- relocation(opacity_j, scale_j, min(count_j + 1, n_max)), opacity clamped to [min_opacity, 1 - 1e-7]
 *                      (mcmc.cpp:149-164), raw values written back; Adam moments of ALL rows[] of j zeroed (mcmc.cpp:87-111)
 *   for every dead i : every parameter row of rows[] <- the (updated) row of source_i
 * rows[k] = {param, exp_avg, exp_avg_sq, width} for the six parameter tensors in the order means, sh0, shN, raw_scales, raw_quats, raw_opacities
 * (exp_avg / exp_avg_sq may be NULL). *n_dead (device, optional) receives the number of dead Gaussians. Nothing happens when no Gaussian is alive. */
typedef struct lfs_param_rows { float* param; float* exp_avg; float* exp_avg_sq; uint32_t width; } lfs_param_rows;
LFS_API size_t lfs_mcmc_relocate_workspace_bytes(uint32_t N);
LFS_API int lfs_mcmc_relocate(
    uint32_t N, const lfs_param_rows* rows /* [6], host */, const double* uniforms /* [N] device */, const float* binoms, int32_t n_max,
    float min_opacity, int32_t* n_dead, void* workspace, size_t workspace_bytes, lfs_stream_t stream);

/* ---- fast_gs::optimizer::adam_step (fastgs/optimizer/include/adam.h:9-20, adam_kernels.cuh:13-36).
 *      bias_correction1_rcp = 1/(1-beta1^t), bias_correction2_sqrt_rcp = 1/sqrt(1-beta2^t)
 *      (fused_adam.cpp:78-79). In place on param / exp_avg / exp_avg_sq. Unlike the reference
 *      (legacy default stream, adam.cu:22) the launch goes to `stream`. */
LFS_API int lfs_adam_step(
    float* param, float* exp_avg, float* exp_avg_sq, const float* param_grad, int64_t n_elements,
    float lr, float beta1, float beta2, float eps, float bias_correction1_rcp, float bias_correction2_sqrt_rcp,
    lfs_stream_t stream);

/* One launch for several parameter tensors (the 6 Gaussian parameter groups of
 * FusedAdam::step, fused_adam.cpp:22-95): same arithmetic as lfs_adam_step per tensor. */
typedef struct lfs_adam_tensor {
    float* param; float* exp_avg; float* exp_avg_sq; const float* grad;
    int64_t n_elements;
    float lr, beta1, beta2, eps, bias_correction1_rcp, bias_correction2_sqrt_rcp;
} lfs_adam_tensor;
#define LFS_ADAM_MAX_TENSORS 8
LFS_API int lfs_adam_step_multi(const lfs_adam_tensor* tensors /* host array */, int32_t n_tensors, lfs_stream_t stream);

/* ---- instrumentation (not in the reference): per-kernel HIP-event timing on the launch stream.
 *      lfs_profile_enable(1) makes every entry point bracket its principal kernel(s) with events;
 *      lfs_profile_collect waits for them, sums by kernel name (names: max_entries x 64 chars)
 *      and clears the log. Returns the number of distinct names written. */
LFS_API int lfs_profile_enable(int on);
LFS_API int lfs_profile_filter(const char* name); /* only time the scopes called `name` (NULL / "" = all) */
LFS_API int lfs_profile_collect(int max_entries, char* names, float* total_ms, int* counts);

/* Library identification: returns "lfs_gsplat gfx950 <abi-version>" */
LFS_API const char* lfs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LFS_GSPLAT_H */
