// Internal (not exported) entry points shared between the operator translation units and the C++ training-step driver (gut_step.hip).
// The "guarded" forms run BEFORE the host knows the intersection count: buffers are sized for `capacity` intersections, the per-tile sort
// classes are launched for lists of up to `assumed_longest` entries, the count stays on the device (n_isects = -1 in the calls below), and
// tile_scan_kernel raises *abort_flag - and empties every list - when either assumption fails (lfs_tilelists.cuh).
#pragma once
#include "../../include/lfs_gsplat.h"
#include <hip/hip_runtime.h>

namespace lfs {

struct IsectGuard { int64_t capacity; int64_t assumed_longest; int32_t* abort_flag; };

// the longest tile list the sort classes launched for `assumed_longest` can order (intersect.hip: <= 1024, <= 4096, <= 16384, global)
inline uint32_t sort_class_limit(int64_t assumed_longest) {
    if (assumed_longest <= 1024) return 1024u;
    if (assumed_longest <= 4096) return 4096u;
    if (assumed_longest <= 16384) return 16384u;
    return 0xFFFFFFFFu;
}

int isect_count_impl(uint32_t C, uint32_t N, const float* means2d, const int32_t* radii, uint32_t tile_size, uint32_t tile_width, uint32_t tile_height,
                     int32_t* tiles_per_gauss, int64_t* n_isects, int64_t* max_tile_isects, int32_t* tile_offsets, uint32_t flags, int64_t* stamp_out, int64_t stamp,
                     void* workspace, size_t workspace_bytes, hipStream_t s, const IsectGuard* guard);
// guard != nullptr: n_isects is ignored (read from the workspace offsets on the device), scratch is required
int isect_emit_impl(uint32_t C, uint32_t N, const float* means2d, const int32_t* radii, const float* depths, uint32_t tile_size, uint32_t tile_width,
                    uint32_t tile_height, int sort, int64_t n_isects, const int32_t* tiles_per_gauss, int64_t* isect_ids, int32_t* flatten_ids,
                    int32_t* tile_offsets, int64_t* scratch, int64_t max_tile_isects, void* workspace, size_t workspace_bytes, hipStream_t s, const IsectGuard* guard);
bool isect_two_pass_supported(uint32_t C, uint32_t N, uint32_t tile_width, uint32_t tile_height);
// the [T + 1] offsets inside an intersection workspace (offsets[T] = n_isects): what the guarded rasterizer calls take as tile_offsets
const int32_t* isect_workspace_offsets(void* workspace, uint32_t C, uint32_t N, uint32_t tile_width, uint32_t tile_height);

// rasterizer, guarded: n_isects = -1 (the kernels read offsets[T]; tile_offsets must be the [T + 1] array above), cell lists sized for `capacity`
int raster_fwd_guarded(uint32_t N, const float* means, const float* quats, const float* scales, const float* colors, const float* opacities,
                       const float* backgrounds, const lfs_cameras* cams, uint32_t tile_size, const int32_t* tile_offsets, const int32_t* flatten_ids,
                       int64_t capacity, float* render_colors, float* render_alphas, int32_t* last_ids, void* workspace, size_t workspace_bytes, hipStream_t s,
                       bool cams_ready = false, bool records_ready = false, // records_ready: the projection kernel wrote the records + culling records (no raster_pack launch)
                       hipEvent_t wait_before_fwd = nullptr);               // the stream waits for this event between the culling kernel and the forward kernel (pipelined step)
void raster_workspace_parts(void* workspace, uint32_t N, void** cams_dev, void** recs, void** cull);
int sh_model_fwd_impl(uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* means, const float* viewmat, const float* sh0, const float* shN,
                      const int32_t* radii /* NULL: every Gaussian */, float* colors, hipStream_t s);
int raster_bwd_mse_acc_guarded(uint32_t N, const float* means, const float* quats, const float* scales, const float* colors, const float* opacities,
                               const float* backgrounds, const lfs_cameras* cams, uint32_t tile_size, const int32_t* tile_offsets, const int32_t* flatten_ids,
                               int64_t capacity, const float* render_colors, const float* render_alphas, const int32_t* last_ids, const float* target_chw,
                               float weight, void* workspace, size_t workspace_bytes, hipStream_t s);
int raster_bwd_acc_guarded(uint32_t N, const float* means, const float* quats, const float* scales, const float* colors, const float* opacities,
                           const float* backgrounds, const lfs_cameras* cams, uint32_t tile_size, const int32_t* tile_offsets, const int32_t* flatten_ids,
                           int64_t capacity, const float* render_alphas, const int32_t* last_ids, const float* v_render_colors, void* workspace,
                           size_t workspace_bytes, hipStream_t s);
size_t raster_workspace_bytes_for(uint32_t N, uint32_t image_width, uint32_t image_height, uint32_t tile_size, int64_t capacity);
int gut_finish_adam_impl(uint32_t N, float* means, float* raw_scales, float* raw_quats, float* raw_opacities, const float* quats, const float* scales,
                         const float* opacities, const float* v_dirs, float* const* exp_avg, float* const* exp_avg_sq, const float* scalars, float scale_reg,
                         float opacity_reg, float* loss, void* workspace, size_t workspace_bytes, hipStream_t s, const int32_t* abort_flag);
int sh_model_bwd_adam_all_impl(uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* means, const float* viewmat, float* sh0, float* shN,
                               const int32_t* radii, const float* colors, const float* acc_rows, float* v_dirs, float* sh0_exp_avg, float* sh0_exp_avg_sq,
                               const float* sh0_scalars, float* shN_exp_avg, float* shN_exp_avg_sq, const float* shN_scalars, hipStream_t s,
                               const int32_t* abort_flag);

// lfs_activations_project_ut with two riders for the training step (both save a launch of a few microseconds each): `zero_words` [zero_n] is cleared (the
// intersection stage's per-tile totals - the count kernel then runs with LFS_ISECT_COUNTERS_ZERO, no memset), and the device-side camera state the rasterizer
// kernels read is written to `cams_out` (what cam_prep_kernel computes; raster_fwd_guarded(..., cams_ready = true) then skips that kernel)
int activations_project_ut_impl(uint32_t N, const float* means, const float* raw_quats, const float* raw_scales, const float* raw_opacities, const lfs_cameras* cams,
                                float eps2d, float near_plane, float far_plane, float radius_clip, const lfs_ut_params* ut_params, float* quats, float* scales,
                                float* opacities, int32_t* radii, float* means2d, float* depths, uint32_t* zero_words, uint32_t zero_n, void* cams_out, hipStream_t s,
                                void* recs_out = nullptr, void* cull_out = nullptr, const float* pack_colors = nullptr); // (given: the rasterizer's records - with these colours [N,3] - and culling records from the same pass, lfs_raster_pack.cuh)
uint32_t* isect_workspace_totals(void* workspace, uint32_t C, uint32_t N, uint32_t tile_width, uint32_t tile_height);
int sh_model_bwd_rows_impl(uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* means, const float* viewmat, const float* sh0, const float* shN,
                           const int32_t* radii, const float* colors, const float* acc_rows, int accumulate, float* v_sh0, float* v_shN, float* v_dirs, hipStream_t s,
                           float* shN_exp_avg = nullptr, float* shN_exp_avg_sq = nullptr, const float* shN_scalars = nullptr); // given: shN's Adam step inline, v_shN unused
// lfs_gut_finish_grads with dL/d(dirs) [N,3] (nullable) added to the means gradient and no dL/dcolour output (the SH backward has run already)
int gut_finish_grads_impl(uint32_t N, const float* means, const float* raw_quats, const float* quats, const float* scales, const float* opacities, float scale_reg,
                          float opacity_reg, int accumulate, float* g_means, float* g_raw_scales, float* g_raw_quats, float* g_raw_opacities, float* v_colors,
                          const float* v_dirs, float* loss, void* workspace, size_t workspace_bytes, hipStream_t s);

// the fused tail of the all-inline step (raster.hip: gut_tail_kernel): SH backward + the six Adam updates in one pass; next_viewmat (nullable): also the NEXT view's SH
// colours -> colors [N,3], every Gaussian. exp_avg / exp_avg_sq / scalars in FusedAdam's group order (lfs_gut_step_args). LFS_E_UNSUPPORTED for K > 16.
int gut_tail_impl(uint32_t N, uint32_t K, uint32_t degrees_to_use, float* means, float* sh0, float* shN, float* raw_scales, float* raw_quats, float* raw_opacities,
                  const float* quats, const float* scales, const float* opacities, const float* viewmat, const float* next_viewmat, const int32_t* radii, float* colors,
                  float* const* exp_avg, float* const* exp_avg_sq, const float (*scalars)[6], float scale_reg, float opacity_reg, float* loss, void* workspace,
                  size_t workspace_bytes, hipStream_t s, const int32_t* abort_flag);

// pipelined training step (gut_step.hip): the SH backward in two kernels on two streams (sh.hip), and the SH colours written into the rasterizer's records
int sh_pipe_dirs_impl(uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* means, const float* viewmat, const float* shN, const int32_t* radii,
                      const float* colors, const float* acc_rows, float* v_dirs, void* handover /* 32 B per Gaussian */, const int32_t* abort_flag, int32_t* abort_snapshot,
                      hipStream_t s);
int sh_pipe_adam_impl(uint32_t n, uint32_t K, uint32_t degrees_to_use, float* sh0, float* shN, const void* handover, float* sh0_exp_avg, float* sh0_exp_avg_sq,
                      const float* sh0_scalars, float* shN_exp_avg, float* shN_exp_avg_sq, const float* shN_scalars, const int32_t* abort_snapshot, hipStream_t s);
int sh_model_fwd_records_impl(uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* means, const float* viewmat, const float* sh0, const float* shN,
                              const int32_t* radii, float* colors, float* rec_rgb, uint32_t rec_stride, hipStream_t s);

} // namespace lfs
