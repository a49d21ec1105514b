// The --gut training step as ONE host call (C++; replaces the reference's Trainer::train_step -> rasterize() -> Ops.h sequence of
// src/training/trainer.cpp:579-770 and rasterization/rasterizer.cpp:200-344 for the path BASELINE.json's metric is quoted on: one camera per step,
// pinhole, global shutter, RGB, MSE, fused Adam).
//
// What was Python + ctypes in rounds 1-2 (fused.py: ~16 launches, each behind an interpreter round trip, and one BLOCKING host read of the
// intersection count in the middle of the step: 0.41 ms of the driver's 2.05 ms per step was not GPU time) is a straight-line enqueue here:
//
//   activations + 3DGUT projection -> tile count + scan -> SH colours -> row / tile binning -> per-tile sort -> pack -> cull -> forward ->
//   backward (MSE folded in) -> SH backward + Adam(sh0, shN) -> finish + activation backward + Adam(means, scales, quats, opacities)
//
// with NO host read on the critical path. The reference synchronises for n_isects (gsplat/Intersect.cpp:75-76) because it allocates its key / value
// arrays to that size. Here the arrays are sized for a CAPACITY the caller picked (last step's count plus a margin), the count stays on the
// device (offsets[T]; every kernel that needs it reads it there), and tile_scan_kernel guards the assumption: a count above the capacity - or a tile
// list longer than the sort classes that were launched - raises a device flag, empties all lists, and makes the two Adam kernels return without
// touching a parameter. The host looks at the (pinned) counts only AFTER it has enqueued the whole step, when they have long been written; in the rare
// overflow case it enlarges the workspace and runs the same step again - nothing was updated by the failed attempt. The GPU queue never drains.
#include "lfs_step_internal.h"
#include "lfs_prof.h"
#include <chrono>
#include <cstdlib>
#include <thread>

namespace lfs {
namespace {

inline size_t a256(size_t v) { return (v + 255) & ~size_t(255); }

struct StepWs {
    float *quats, *scales, *opacities, *means2d, *depths, *colors, *v_dirs, *render, *alpha;
    int32_t *radii, *tiles_per_gauss, *flatten_ids, *last_ids, *abort_flag, *abort_snapshot;
    void* handover;   // pipelined step: 32 bytes per Gaussian from the SH direction kernel (main stream) to the SH Adam kernel (side stream)
    int64_t *isect_ids, *binned, *dev_counts;
    void *isect_ws, *raster_ws;
    size_t isect_ws_bytes, raster_ws_bytes, bytes;
};

bool step_ws(void* base, uint32_t N, uint32_t W, uint32_t H, uint32_t tile, int64_t capacity, StepWs& w, lfs_gut_step_layout* lay) {
    if (tile == 0 || W == 0 || H == 0 || capacity <= 0) return false;
    const uint32_t tw = (W + tile - 1) / tile, th = (H + tile - 1) / tile;
    char* p = static_cast<char*>(base);
    size_t o = 0;
    auto take = [&](size_t nbytes) { const size_t at = o; o += a256(nbytes); return at; };
    const size_t n = N, P = size_t(W) * H, cap = size_t(capacity);
    const size_t o_quats = take(16 * n), o_scales = take(12 * n), o_opac = take(4 * n), o_radii = take(8 * n), o_m2d = take(8 * n), o_depths = take(4 * n);
    const size_t o_tpg = take(4 * n), o_colors = take(12 * n), o_vdirs = take(12 * n);
    w.isect_ws_bytes = lfs_intersect_tile_workspace_bytes(1, N, tw, th);
    const size_t o_iws = take(w.isect_ws_bytes);
    const size_t o_ids = take(8 * cap), o_flat = take(4 * cap), o_binned = take(8 * cap);
    w.raster_ws_bytes = raster_workspace_bytes_for(N, W, H, tile, capacity);
    if (w.raster_ws_bytes == 0) return false;
    const size_t o_rws = take(w.raster_ws_bytes);
    const size_t o_render = take(12 * P), o_alpha = take(4 * P), o_last = take(4 * P), o_flag = take(4), o_counts = take(32);
    const size_t o_hand = take(32 * n), o_snap = take(4);   // (appended in round 6: every earlier offset is where it was)
    w.bytes = o;
    w.handover = p + o_hand; w.abort_snapshot = (int32_t*)(p + o_snap);
    w.quats = (float*)(p + o_quats); w.scales = (float*)(p + o_scales); w.opacities = (float*)(p + o_opac); w.radii = (int32_t*)(p + o_radii);
    w.means2d = (float*)(p + o_m2d); w.depths = (float*)(p + o_depths); w.tiles_per_gauss = (int32_t*)(p + o_tpg); w.colors = (float*)(p + o_colors);
    w.v_dirs = (float*)(p + o_vdirs); w.isect_ws = p + o_iws; w.isect_ids = (int64_t*)(p + o_ids); w.flatten_ids = (int32_t*)(p + o_flat);
    w.binned = (int64_t*)(p + o_binned); w.raster_ws = p + o_rws; w.render = (float*)(p + o_render); w.alpha = (float*)(p + o_alpha);
    w.last_ids = (int32_t*)(p + o_last); w.abort_flag = (int32_t*)(p + o_flag); w.dev_counts = (int64_t*)(p + o_counts);
    if (lay) {
        lay->bytes = o; lay->quats = o_quats; lay->scales = o_scales; lay->opacities = o_opac; lay->radii = o_radii; lay->means2d = o_m2d; lay->depths = o_depths;
        lay->colors = o_colors; lay->isect_ids = o_ids; lay->flatten_ids = o_flat; lay->render = o_render; lay->alpha = o_alpha; lay->last_ids = o_last;
        lay->abort_flag = o_flag; lay->counts = o_counts;
        lay->tile_offsets = o_iws + size_t(reinterpret_cast<const char*>(isect_workspace_offsets(nullptr, 1, N, tw, th)) - static_cast<const char*>(nullptr));
    }
    return true;
}

struct Front { lfs_cameras cams; const int32_t* offsets; };

// ---- the pipelined step's side stream -------------------------------------------------------------------------------------------------------------------
// One per process (one process per GPU). The SH colour kernel of step k + 1 and the SH Adam kernel of step k live on it; three events tie it to the caller's
// stream (see lfs_gut_train_step_pipelined below). `done` is what lfs_gut_pipeline_join makes a stream wait for.
struct Pipeline {
    hipStream_t side = nullptr;
    hipEvent_t projected = nullptr, colours = nullptr, dirs = nullptr, done = nullptr;
    bool pending = false;   // a side-stream update has been enqueued since the last join
    bool start_after_finish = false;
    int init() {
        if (side != nullptr) return LFS_OK;
        hipError_t e = hipSuccess;
#ifndef LFS_EMULATE
        // measurement knobs (tools/r6_lease10.sh): LFS_PIPE_PRIO = low | high (side stream priority), LFS_PIPE_CUMASK = n (side stream confined to every n-th CU),
        // LFS_PIPE_START = finish (the SH Adam pass starts behind the finish pass instead of beside it)
        const char* prio = getenv("LFS_PIPE_PRIO"); const char* cum = getenv("LFS_PIPE_CUMASK"); const char* st = getenv("LFS_PIPE_START");
        start_after_finish = st != nullptr && st[0] == 'f';
        if (cum != nullptr && atoi(cum) > 1) {
            const int every = atoi(cum);
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int cu = 0; cu < 256; ++cu) if (cu % every == 0) mask[cu >> 5] |= 1u << (cu & 31);
            e = hipExtStreamCreateWithCUMask(&side, 8, mask);
        } else if (prio != nullptr) {
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // (numerically: lo = least urgent, hi = most urgent)
            e = hipStreamCreateWithPriority(&side, hipStreamNonBlocking, prio[0] == 'l' ? lo : hi);
        } else
#endif
        e = hipStreamCreateWithFlags(&side, hipStreamNonBlocking);   // (non-blocking: no implicit ordering against the legacy default stream torch may hand in)
        if (e != hipSuccess) { side = nullptr; return (int)e; }
        for (hipEvent_t* ev : {&projected, &colours, &dirs, &done}) {
            e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
            if (e != hipSuccess) return (int)e;
        }
        return LFS_OK;
    }
};
Pipeline g_pipe;
// the other step forms read / write sh0 and shN on the caller's stream: they wait for a pending side-stream update first
int auto_join(hipStream_t s) {
    if (!g_pipe.pending || g_pipe.side == nullptr) return LFS_OK;
    if (hipStreamWaitEvent(s, g_pipe.done, 0) != hipSuccess) return LFS_E_INVALID;
    g_pipe.pending = false;
    return LFS_OK;
}

// everything up to and including the rasterizer forward; shared by the Adam-inline step and the gradient-tensor step
int enqueue_forward(const lfs_gut_step_args* a, const StepWs& w, int64_t capacity, int64_t assumed_longest, int64_t* host_counts, int64_t stamp, hipStream_t s, Front& f,
                    Pipeline* pipe = nullptr, bool colors_ready = false) {   // colors_ready: w.colors holds this view's SH colours already (the previous step's fused tail wrote them)
    const uint32_t N = a->N, W = a->image_width, H = a->image_height, tile = a->tile_size;
    const uint32_t tw = (W + tile - 1) / tile, th = (H + tile - 1) / tile;
    lfs_cameras& cams = f.cams;
    cams = lfs_cameras{};
    cams.C = 1; cams.image_width = W; cams.image_height = H; cams.camera_model = LFS_CAMERA_PINHOLE; cams.rs_type = LFS_SHUTTER_GLOBAL;
    cams.viewmats0 = a->viewmat; cams.Ks = a->Kmat;
    const lfs_ut_params ut{0.1f, 2.f, 0.f, 0.1f, 1};   // Cameras.h:27-61 defaults, as the trainer passes them (rasterizer_autograd.cpp:223-234)
    // trainer constants of rasterizer.cpp:176-181: eps2d 0.3, near 0.01, far 1e4, radius_clip 0
    // the projection kernel also clears the intersection stage's per-tile totals and writes the rasterizer's camera state (first block of the raster workspace):
    // two launches of a few microseconds each that a step does not need (the 32 KB memset and cam_prep_kernel)
    // Round 4: the SH colours are evaluated FIRST (for every Gaussian - visibility is not known yet, 6 % more coefficient rows on SYN-B) so that the projection
    // kernel, which has the activated quaternion / scale / opacity in registers, can write the rasterizer's 64-byte record and the 32-byte culling record of every
    // visible Gaussian itself: raster_pack_kernel's second pass over the Gaussians (0.040 ms, 152 MB re-read) is gone. Debug bit 6: the round-3 order (A/B, tests).
    const bool pack_here = pipe != nullptr || !(lfs_get_debug_flags() & 64u);
    void *recs = nullptr, *cull = nullptr;
    raster_workspace_parts(w.raster_ws, N, nullptr, &recs, &cull);
    int rc = LFS_OK;
    if (pipe != nullptr) {
        // Pipelined: the projection goes FIRST and packs the records without colours; the SH colours (visible Gaussians only - the radii exist now) follow on the side
        // stream, behind the previous step's SH Adam kernel, into `colors` and into the rgb slots of the records; the main stream carries on with the tile lists and
        // waits for them in front of the forward kernel.
        rc = activations_project_ut_impl(N, a->means, a->raw_quats, a->raw_scales, a->raw_opacities, &cams, 0.3f, 0.01f, 10000.f, 0.f, &ut, w.quats, w.scales,
                                         w.opacities, w.radii, w.means2d, w.depths, isect_workspace_totals(w.isect_ws, 1, N, tw, th), tw * th, w.raster_ws, s, recs, cull, nullptr);
        if (rc) return rc;
        hipError_t e = hipEventRecord(pipe->projected, s);
        if (e == hipSuccess) e = hipStreamWaitEvent(pipe->side, pipe->projected, 0);
        if (e != hipSuccess) return (int)e;
        rc = sh_model_fwd_records_impl(N, a->K, a->sh_degree, a->means, a->viewmat, a->sh0, a->shN, w.radii, w.colors, static_cast<float*>(recs) + 13, 16, pipe->side);
        if (rc) return rc;
        e = hipEventRecord(pipe->colours, pipe->side);
        if (e != hipSuccess) return (int)e;
        const IsectGuard guard{capacity, assumed_longest, w.abort_flag};
        int64_t* counts = host_counts ? host_counts : w.dev_counts;
        rc = isect_count_impl(1, N, w.means2d, w.radii, tile, tw, th, w.tiles_per_gauss, counts, counts + 1, nullptr, LFS_ISECT_COUNTERS_ZERO, counts + 2, stamp, w.isect_ws,
                              w.isect_ws_bytes, s, &guard);
        if (rc) return rc;
        rc = isect_emit_impl(1, N, w.means2d, w.radii, w.depths, tile, tw, th, 1, -1, w.tiles_per_gauss, w.isect_ids, w.flatten_ids, nullptr, w.binned, -1, w.isect_ws,
                             w.isect_ws_bytes, s, &guard);
        if (rc) return rc;
        f.offsets = isect_workspace_offsets(w.isect_ws, 1, N, tw, th);
        return raster_fwd_guarded(N, a->means, w.quats, w.scales, w.colors, w.opacities, a->background, &cams, tile, f.offsets, w.flatten_ids, capacity, w.render, w.alpha,
                                  w.last_ids, w.raster_ws, w.raster_ws_bytes, s, /*cams_ready=*/true, /*records_ready=*/true, /*wait_before_fwd=*/pipe->colours);
    }
    if (pack_here && !colors_ready) {
        rc = sh_model_fwd_impl(N, a->K, a->sh_degree, a->means, a->viewmat, a->sh0, a->shN, nullptr, w.colors, s);
        if (rc) return rc;
    }
    rc = activations_project_ut_impl(N, a->means, a->raw_quats, a->raw_scales, a->raw_opacities, &cams, 0.3f, 0.01f, 10000.f, 0.f, &ut, w.quats, w.scales,
                                     w.opacities, w.radii, w.means2d, w.depths, isect_workspace_totals(w.isect_ws, 1, N, tw, th), tw * th, w.raster_ws, s,
                                     pack_here ? recs : nullptr, pack_here ? cull : nullptr, pack_here ? w.colors : nullptr);
    if (rc) return rc;
    const IsectGuard guard{capacity, assumed_longest, w.abort_flag};
    int64_t* counts = host_counts ? host_counts : w.dev_counts;   // [n_isects, longest tile list, stamp]
    rc = isect_count_impl(1, N, w.means2d, w.radii, tile, tw, th, w.tiles_per_gauss, counts, counts + 1, nullptr, LFS_ISECT_COUNTERS_ZERO, counts + 2, stamp, w.isect_ws,
                          w.isect_ws_bytes, s, &guard);
    if (rc) return rc;
    if (!pack_here && !colors_ready) { // the SH colours need the projection's radii only: enqueued between the count and the binning passes, as the Python step did with its `overlap` hook
        rc = lfs_sh_model_fwd(N, a->K, a->sh_degree, a->means, a->viewmat, a->sh0, a->shN, w.radii, w.colors, s);
        if (rc) return rc;
    }
    rc = isect_emit_impl(1, N, w.means2d, w.radii, w.depths, tile, tw, th, 1, -1, w.tiles_per_gauss, w.isect_ids, w.flatten_ids, nullptr, w.binned, -1, w.isect_ws,
                         w.isect_ws_bytes, s, &guard);
    if (rc) return rc;
    f.offsets = isect_workspace_offsets(w.isect_ws, 1, N, tw, th);
    return raster_fwd_guarded(N, a->means, w.quats, w.scales, w.colors, w.opacities, a->background, &cams, tile, f.offsets, w.flatten_ids, capacity, w.render, w.alpha,
                              w.last_ids, w.raster_ws, w.raster_ws_bytes, s, /*cams_ready=*/true, /*records_ready=*/pack_here);
}

int check_args(const lfs_gut_step_args* a, bool need_adam) {
    if (!a || !a->means || !a->sh0 || !a->raw_scales || !a->raw_quats || !a->raw_opacities || !a->viewmat || !a->Kmat) return LFS_E_INVALID;
    if (a->N == 0 || a->K == 0 || a->K > 32 || (a->K > 1 && !a->shN)) return LFS_E_INVALID;
    if ((a->sh_degree + 1) * (a->sh_degree + 1) > a->K || a->sh_degree > 4) return LFS_E_INVALID;
    if (need_adam) {
        if (a->K < 2 || !a->target_chw || !a->loss) return LFS_E_INVALID;   // (degree-0-only models take the gradient-tensor step: there is no shN to update)
        for (int k = 0; k < 6; ++k) if (!a->exp_avg[k] || !a->exp_avg_sq[k]) return LFS_E_INVALID;
    }
    return LFS_OK;
}

} // namespace
} // namespace lfs

using namespace lfs;

extern "C" int lfs_gut_step_layout_for(uint32_t N, uint32_t image_width, uint32_t image_height, uint32_t tile_size, int64_t capacity, lfs_gut_step_layout* out) {
    if (!out) return LFS_E_INVALID;
    StepWs w;
    if (!step_ws(nullptr, N, image_width, image_height, tile_size, capacity, w, out)) return LFS_E_INVALID;
    return LFS_OK;
}

// 1: the speculative step takes this problem shape (its binning needs the two-pass scatter: at most 512 tile rows, Gaussian index + tile column in 32 bits, debug
// bit 5 off; tile sizes the rasterizer has cell kernels for). 0: lfs_gut_* would return LFS_E_UNSUPPORTED - the caller enqueues the operators one by one instead.
extern "C" int lfs_gut_step_supported(uint32_t N, uint32_t image_width, uint32_t image_height, uint32_t tile_size) {
    if (N == 0 || tile_size < 8 || tile_size > 64 || (tile_size & 7) || image_width == 0 || image_height == 0) return 0;
    const uint32_t tw = (image_width + tile_size - 1) / tile_size, th = (image_height + tile_size - 1) / tile_size;
    return isect_two_pass_supported(1, N, tw, th) ? 1 : 0;
}

extern "C" int lfs_gut_step_fits(int64_t n_isects, int64_t longest, int64_t capacity, int64_t assumed_longest) {
    return (n_isects <= capacity && uint64_t(longest) <= uint64_t(sort_class_limit(assumed_longest))) ? 1 : 0;
}

extern "C" int lfs_gut_train_step(const lfs_gut_step_args* a, int64_t capacity, int64_t assumed_longest, void* workspace, size_t workspace_bytes,
                                  int64_t* host_counts, int64_t stamp, lfs_stream_t stream) {
    int rc = check_args(a, true);
    if (rc) return rc;
    if (!workspace) return LFS_E_INVALID;
    StepWs w;
    if (!step_ws(workspace, a->N, a->image_width, a->image_height, a->tile_size, capacity, w, nullptr)) return LFS_E_INVALID;
    if (workspace_bytes < w.bytes) return LFS_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    rc = auto_join(s);
    if (rc) return rc;
    Front f;
    rc = enqueue_forward(a, w, capacity, assumed_longest, host_counts, stamp, s, f);
    if (rc) return rc;
    rc = raster_bwd_mse_acc_guarded(a->N, a->means, w.quats, w.scales, w.colors, w.opacities, a->background, &f.cams, a->tile_size, f.offsets, w.flatten_ids, capacity,
                                    w.render, w.alpha, w.last_ids, a->target_chw, a->loss_weight, w.raster_ws, w.raster_ws_bytes, s);
    if (rc) return rc;
    const float* acc_rows = reinterpret_cast<const float*>(static_cast<const char*>(w.raster_ws) + lfs_rasterize_workspace_acc_offset(1, a->N));
    rc = sh_model_bwd_adam_all_impl(a->N, a->K, a->sh_degree, a->means, a->viewmat, a->sh0, a->shN, w.radii, w.colors, acc_rows, w.v_dirs, a->exp_avg[1], a->exp_avg_sq[1],
                                    a->adam[1], a->exp_avg[2], a->exp_avg_sq[2], a->adam[2], s, w.abort_flag);
    if (rc) return rc;
    // lfs_gut_finish_adam's order: means, raw_scales, raw_quats, raw_opacities = FusedAdam groups 0, 3, 4, 5
    float* const m[4] = {a->exp_avg[0], a->exp_avg[3], a->exp_avg[4], a->exp_avg[5]};
    float* const v[4] = {a->exp_avg_sq[0], a->exp_avg_sq[3], a->exp_avg_sq[4], a->exp_avg_sq[5]};
    float sc[24];
    const int grp[4] = {0, 3, 4, 5};
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 6; ++j) sc[6 * k + j] = a->adam[grp[k]][j];
    return gut_finish_adam_impl(a->N, a->means, a->raw_scales, a->raw_quats, a->raw_opacities, w.quats, w.scales, w.opacities, w.v_dirs, m, v, sc, a->scale_reg,
                                a->opacity_reg, a->loss, w.raster_ws, w.raster_ws_bytes, s, w.abort_flag);
}

// lfs_gut_train_step with its three per-Gaussian tail passes as ONE (round 6; raster.hip: gut_tail_kernel): SH backward + Adam(sh0, shN) + finish + Adam(means, scales,
// quaternions, opacities) in one launch, dL/d(dirs) handed over in registers - and, when the caller names the NEXT step's view (next_viewmat, device [4,4]), that view's SH
// colours for every Gaussian from the coefficient rows as they leave their Adam update: the next call then passes colors_ready = 1 and its SH colour kernel is not launched.
//   one-call step :  ... backward | SH backward + Adam (0.26 ms) | finish + Adam (0.10) | [next step] SH colours (0.07) | projection ...
//   this form     :  ... backward | tail (SH backward + six Adam updates + next colours) | [next step] projection ...
// colors_ready = 1 is the caller's statement that (a) the previous call on this workspace was this entry point with next_viewmat pointing at the matrix args->viewmat holds
// now, (b) with the same N, K and sh_degree, (c) it fitted its buffers (lfs_gut_step_fits), and (d) nothing has written means / sh0 / shN since. gut_step.GutStep keeps that
// book. Same results as lfs_gut_train_step, bit for bit in the deterministic accumulation mode. K > 16 (SH degree 4): the three separate passes run, as there.
extern "C" int lfs_gut_train_step_ex(const lfs_gut_step_args* a, const float* next_viewmat, int colors_ready, int64_t capacity, int64_t assumed_longest, void* workspace,
                                     size_t workspace_bytes, int64_t* host_counts, int64_t stamp, lfs_stream_t stream) {
    int rc = check_args(a, true);
    if (rc) return rc;
    if (!workspace) return LFS_E_INVALID;
    StepWs w;
    if (!step_ws(workspace, a->N, a->image_width, a->image_height, a->tile_size, capacity, w, nullptr)) return LFS_E_INVALID;
    if (workspace_bytes < w.bytes) return LFS_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    rc = auto_join(s);
    if (rc) return rc;
    Front f;
    rc = enqueue_forward(a, w, capacity, assumed_longest, host_counts, stamp, s, f, nullptr, colors_ready != 0);
    if (rc) return rc;
    rc = raster_bwd_mse_acc_guarded(a->N, a->means, w.quats, w.scales, w.colors, w.opacities, a->background, &f.cams, a->tile_size, f.offsets, w.flatten_ids, capacity,
                                    w.render, w.alpha, w.last_ids, a->target_chw, a->loss_weight, w.raster_ws, w.raster_ws_bytes, s);
    if (rc) return rc;
    rc = gut_tail_impl(a->N, a->K, a->sh_degree, a->means, a->sh0, a->shN, a->raw_scales, a->raw_quats, a->raw_opacities, w.quats, w.scales, w.opacities, a->viewmat,
                       next_viewmat, w.radii, w.colors, a->exp_avg, a->exp_avg_sq, a->adam, a->scale_reg, a->opacity_reg, a->loss, w.raster_ws, w.raster_ws_bytes, s, w.abort_flag);
    if (rc != LFS_E_UNSUPPORTED) return rc;
    // K > 16: the separate passes of lfs_gut_train_step (no colours for the next step: the caller's next call must pass colors_ready = 0 - GutStep checks K)
    const float* acc_rows = reinterpret_cast<const float*>(static_cast<const char*>(w.raster_ws) + lfs_rasterize_workspace_acc_offset(1, a->N));
    rc = sh_model_bwd_adam_all_impl(a->N, a->K, a->sh_degree, a->means, a->viewmat, a->sh0, a->shN, w.radii, w.colors, acc_rows, w.v_dirs, a->exp_avg[1], a->exp_avg_sq[1],
                                    a->adam[1], a->exp_avg[2], a->exp_avg_sq[2], a->adam[2], s, w.abort_flag);
    if (rc) return rc;
    float* const m[4] = {a->exp_avg[0], a->exp_avg[3], a->exp_avg[4], a->exp_avg[5]};
    float* const v[4] = {a->exp_avg_sq[0], a->exp_avg_sq[3], a->exp_avg_sq[4], a->exp_avg_sq[5]};
    float sc[24];
    const int grp[4] = {0, 3, 4, 5};
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 6; ++j) sc[6 * k + j] = a->adam[grp[k]][j];
    return gut_finish_adam_impl(a->N, a->means, a->raw_scales, a->raw_quats, a->raw_opacities, w.quats, w.scales, w.opacities, w.v_dirs, m, v, sc, a->scale_reg,
                                a->opacity_reg, a->loss, w.raster_ws, w.raster_ws_bytes, s, w.abort_flag);
}

// lfs_gut_train_step with the step's HBM-bound SH kernels moved UNDER its latency- and VALU-bound front end (round 6). Same arguments, same results (bit for bit in
// the deterministic accumulation mode); what changes is the ORDER across two streams:
//
//   stream (caller's)                                                         side stream (the library's)
//   projection(k) -> records without colours ----- event `projected` ------>  SH colours(k): visible Gaussians, -> colors [N,3] + the rgb slots of the records
//   tile count + scan, row / tile binning, sort, culling                      (behind SH Adam(k - 1): the coefficients it reads are the updated ones)
//   <------------------------------------------------ event `colours` ------
//   forward, backward (MSE folded in)
//   SH direction pass(k): dL/d(dirs) + 32-byte hand-over rows --- `dirs` -->  SH Adam(k): sh0 / shN read-modify-write, 1.1 GB at 1 M Gaussians
//   finish + Adam(means, scales, quaternions, opacities)                      |  runs under finish(k), projection(k + 1), binning(k + 1), sort, culling:
//   [next call] projection(k + 1) ...                                         v  kernels that leave the HBM idle
//
// In lfs_gut_train_step everything is one chain, and 0.33 ms of it (SH backward + Adam, SH colours) is pure HBM time during which no other kernel can run, while the
// front end (0.25 ms) is latency / LDS / VALU bound and moves < 1 TB/s. Contract: between two pipelined calls sh0, shN and their moments belong to the side stream -
// call lfs_gut_pipeline_join(stream) before anything else reads or writes them on `stream` (another step form, a strategy, evaluation, a checkpoint). Every other
// tensor is ordered on `stream` as before. The price: the direction pass re-reads the coefficient rows of the Gaussians that received a gradient (<= 180 MB).
extern "C" int lfs_gut_train_step_pipelined(const lfs_gut_step_args* a, int64_t capacity, int64_t assumed_longest, void* workspace, size_t workspace_bytes,
                                            int64_t* host_counts, int64_t stamp, lfs_stream_t stream) {
    int rc = check_args(a, true);
    if (rc) return rc;
    if (!workspace) return LFS_E_INVALID;
    StepWs w;
    if (!step_ws(workspace, a->N, a->image_width, a->image_height, a->tile_size, capacity, w, nullptr)) return LFS_E_INVALID;
    if (workspace_bytes < w.bytes) return LFS_E_WORKSPACE;
    rc = g_pipe.init();
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    Front f;
    rc = enqueue_forward(a, w, capacity, assumed_longest, host_counts, stamp, s, f, &g_pipe);
    if (rc) return rc;
    rc = raster_bwd_mse_acc_guarded(a->N, a->means, w.quats, w.scales, w.colors, w.opacities, a->background, &f.cams, a->tile_size, f.offsets, w.flatten_ids, capacity,
                                    w.render, w.alpha, w.last_ids, a->target_chw, a->loss_weight, w.raster_ws, w.raster_ws_bytes, s);
    if (rc) return rc;
    const float* acc_rows = reinterpret_cast<const float*>(static_cast<const char*>(w.raster_ws) + lfs_rasterize_workspace_acc_offset(1, a->N));
    rc = sh_pipe_dirs_impl(a->N, a->K, a->sh_degree, a->means, a->viewmat, a->shN, w.radii, w.colors, acc_rows, w.v_dirs, w.handover, w.abort_flag, w.abort_snapshot, s);
    if (rc) return rc;
    float* const m[4] = {a->exp_avg[0], a->exp_avg[3], a->exp_avg[4], a->exp_avg[5]};
    float* const v[4] = {a->exp_avg_sq[0], a->exp_avg_sq[3], a->exp_avg_sq[4], a->exp_avg_sq[5]};
    float sc[24];
    const int grp[4] = {0, 3, 4, 5};
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 6; ++j) sc[6 * k + j] = a->adam[grp[k]][j];
    if (g_pipe.start_after_finish) {
        rc = gut_finish_adam_impl(a->N, a->means, a->raw_scales, a->raw_quats, a->raw_opacities, w.quats, w.scales, w.opacities, w.v_dirs, m, v, sc, a->scale_reg,
                                  a->opacity_reg, a->loss, w.raster_ws, w.raster_ws_bytes, s, w.abort_flag);
        if (rc) return rc;
    }
    hipError_t e = hipEventRecord(g_pipe.dirs, s);
    if (e == hipSuccess) e = hipStreamWaitEvent(g_pipe.side, g_pipe.dirs, 0);
    if (e != hipSuccess) return (int)e;
    rc = sh_pipe_adam_impl(a->N, a->K, a->sh_degree, a->sh0, a->shN, w.handover, a->exp_avg[1], a->exp_avg_sq[1], a->adam[1], a->exp_avg[2], a->exp_avg_sq[2], a->adam[2],
                           w.abort_snapshot, g_pipe.side);
    if (rc) return rc;
    e = hipEventRecord(g_pipe.done, g_pipe.side);
    if (e != hipSuccess) return (int)e;
    g_pipe.pending = true;
    if (g_pipe.start_after_finish) return LFS_OK;
    return gut_finish_adam_impl(a->N, a->means, a->raw_scales, a->raw_quats, a->raw_opacities, w.quats, w.scales, w.opacities, w.v_dirs, m, v, sc, a->scale_reg,
                                a->opacity_reg, a->loss, w.raster_ws, w.raster_ws_bytes, s, w.abort_flag);
}

// `stream` waits for the side stream's last SH update (no host wait). Cheap and idempotent: call it whenever sh0 / shN / their moments are about to be used outside
// lfs_gut_train_step_pipelined. Returns 1 when there was something to wait for, 0 when not, < 0 on error.
extern "C" int lfs_gut_pipeline_join(lfs_stream_t stream) {
    if (!g_pipe.pending || g_pipe.side == nullptr) return 0;
    const hipError_t e = hipStreamWaitEvent((hipStream_t)stream, g_pipe.done, 0);
    if (e != hipSuccess) return LFS_E_INVALID;
    g_pipe.pending = false;
    return 1;
}

// Backward of the view lfs_gut_view_forward left in the workspace, into GRADIENT TENSORS (data-parallel ranks, several views per step, iterations <= 1000):
// grads = means, sh0, shN, raw_scales, raw_quats, raw_opacities - written (accumulate == 0) or added to. Two halves, so that a data-parallel caller can put
// the all-reduce of the SH gradients (45 of 59 floats per Gaussian at degree 3) on the wire between them:
//   lfs_gut_view_backward_sh     : accumulator-only rasterizer backward (with target_chw the clamped MSE is folded in, otherwise v_render [H,W,3] is the
//                                  caller's dL/d(render)), then the SH backward straight from the accumulator rows -> grads[1], grads[2] final for this view
//   lfs_gut_view_backward_finish : rows + dL/d(dirs) -> grads[0], grads[3..5] (raster_finish + activation backward + regularisers); *loss += the fused MSE
// An attempt that did not fit its buffers (lfs_gut_step_fits) rendered EMPTY lists: the caller checks the counts of the forward before it calls these.
static int view_setup(const lfs_gut_step_args* a, int64_t capacity, void* workspace, size_t workspace_bytes, StepWs& w, lfs_cameras& cams, const int32_t*& offsets) {
    int rc = check_args(a, false);
    if (rc) return rc;
    if (!workspace) return LFS_E_INVALID;
    if (!step_ws(workspace, a->N, a->image_width, a->image_height, a->tile_size, capacity, w, nullptr)) return LFS_E_INVALID;
    if (workspace_bytes < w.bytes) return LFS_E_WORKSPACE;
    const uint32_t tile = a->tile_size, tw = (a->image_width + tile - 1) / tile, th = (a->image_height + tile - 1) / tile;
    cams = lfs_cameras{};
    cams.C = 1; cams.image_width = a->image_width; cams.image_height = a->image_height; cams.camera_model = LFS_CAMERA_PINHOLE; cams.rs_type = LFS_SHUTTER_GLOBAL;
    cams.viewmats0 = a->viewmat; cams.Ks = a->Kmat;
    offsets = isect_workspace_offsets(w.isect_ws, 1, a->N, tw, th);
    return LFS_OK;
}

extern "C" int lfs_gut_view_backward_sh(const lfs_gut_step_args* a, int64_t capacity, const float* v_render, float* const* grads /* [6] host */, int accumulate,
                                        void* workspace, size_t workspace_bytes, lfs_stream_t stream) {
    StepWs w; lfs_cameras cams; const int32_t* offsets;
    int rc = view_setup(a, capacity, workspace, workspace_bytes, w, cams, offsets);
    if (rc) return rc;
    // args->exp_avg[2] given: shN's Adam step runs inside the SH backward (one view per step, one rank - the reference's MCMC / L1+D-SSIM / bilateral-grid
    // steps, whose other five tensors go through gradient tensors and FusedAdam); grads[2] is then neither read nor written
    const bool inline_shN = a->K > 1 && a->exp_avg[2] != nullptr;
    if (inline_shN && (accumulate || !a->exp_avg_sq[2])) return LFS_E_INVALID;
    if (!grads || !grads[1] || (a->K > 1 && !grads[2] && !inline_shN) || (!a->target_chw && !v_render)) return LFS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    rc = auto_join(s);
    if (rc) return rc;
    if (a->target_chw)
        rc = raster_bwd_mse_acc_guarded(a->N, a->means, w.quats, w.scales, w.colors, w.opacities, a->background, &cams, a->tile_size, offsets, w.flatten_ids, capacity,
                                        w.render, w.alpha, w.last_ids, a->target_chw, a->loss_weight, w.raster_ws, w.raster_ws_bytes, s);
    else
        rc = raster_bwd_acc_guarded(a->N, a->means, w.quats, w.scales, w.colors, w.opacities, a->background, &cams, a->tile_size, offsets, w.flatten_ids, capacity, w.alpha,
                                    w.last_ids, v_render, w.raster_ws, w.raster_ws_bytes, s);
    if (rc) return rc;
    const float* acc_rows = reinterpret_cast<const float*>(static_cast<const char*>(w.raster_ws) + lfs_rasterize_workspace_acc_offset(1, a->N));
    return sh_model_bwd_rows_impl(a->N, a->K, a->sh_degree, a->means, a->viewmat, a->sh0, a->shN, w.radii, w.colors, acc_rows, accumulate, grads[1], grads[2], w.v_dirs, s,
                                  inline_shN ? a->exp_avg[2] : nullptr, inline_shN ? a->exp_avg_sq[2] : nullptr, inline_shN ? a->adam[2] : nullptr);
}

extern "C" int lfs_gut_view_backward_finish(const lfs_gut_step_args* a, int64_t capacity, float* const* grads /* [6] host */, int accumulate,
                                            void* workspace, size_t workspace_bytes, lfs_stream_t stream) {
    StepWs w; lfs_cameras cams; const int32_t* offsets;
    int rc = view_setup(a, capacity, workspace, workspace_bytes, w, cams, offsets);
    if (rc) return rc;
    if (!grads || !grads[0] || !grads[3] || !grads[4] || !grads[5] || (a->target_chw && !a->loss)) return LFS_E_INVALID;
    return gut_finish_grads_impl(a->N, a->means, a->raw_quats, w.quats, w.scales, w.opacities, a->scale_reg, a->opacity_reg, accumulate, grads[0], grads[3], grads[4],
                                 grads[5], nullptr, w.v_dirs, a->target_chw ? a->loss : nullptr, w.raster_ws, w.raster_ws_bytes, (hipStream_t)stream);
}

// The view's backward for the FACTORED gradient exchange of the replicated data-parallel layout (dist.ColorGradExchange): rasterizer backward, then the finish pass -
// grads[0], grads[3..5] (means WITHOUT the SH direction term, scales, quaternions, opacities), written or added to - and dL/dcolour [N,3] -> v_colors_out. No SH
// backward here: per view the gradient of the SH coefficients is the outer product basis(direction) x dL/dcolour, every rank knows every rank's camera, so the ranks
// exchange the 3-float rows and each evaluates the multi-view SH backward (lfs_sh_model_bwd_views) over ALL views itself.
extern "C" int lfs_gut_view_backward_rows(const lfs_gut_step_args* a, int64_t capacity, const float* v_render, float* const* grads /* [6] host */, int accumulate,
                                          float* v_colors_out, void* workspace, size_t workspace_bytes, lfs_stream_t stream) {
    StepWs w; lfs_cameras cams; const int32_t* offsets;
    int rc = view_setup(a, capacity, workspace, workspace_bytes, w, cams, offsets);
    if (rc) return rc;
    if (!grads || !grads[0] || !grads[3] || !grads[4] || !grads[5] || !v_colors_out || (!a->target_chw && !v_render) || (a->target_chw && !a->loss)) return LFS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (a->target_chw)
        rc = raster_bwd_mse_acc_guarded(a->N, a->means, w.quats, w.scales, w.colors, w.opacities, a->background, &cams, a->tile_size, offsets, w.flatten_ids, capacity,
                                        w.render, w.alpha, w.last_ids, a->target_chw, a->loss_weight, w.raster_ws, w.raster_ws_bytes, s);
    else
        rc = raster_bwd_acc_guarded(a->N, a->means, w.quats, w.scales, w.colors, w.opacities, a->background, &cams, a->tile_size, offsets, w.flatten_ids, capacity, w.alpha,
                                    w.last_ids, v_render, w.raster_ws, w.raster_ws_bytes, s);
    if (rc) return rc;
    return gut_finish_grads_impl(a->N, a->means, a->raw_quats, w.quats, w.scales, w.opacities, a->scale_reg, a->opacity_reg, accumulate, grads[0], grads[3], grads[4],
                                 grads[5], v_colors_out, nullptr, a->target_chw ? a->loss : nullptr, w.raster_ws, w.raster_ws_bytes, s);
}

extern "C" int lfs_gut_view_backward(const lfs_gut_step_args* a, int64_t capacity, const float* v_render, float* const* grads /* [6] host */, int accumulate,
                                     void* workspace, size_t workspace_bytes, lfs_stream_t stream) {
    const int rc = lfs_gut_view_backward_sh(a, capacity, v_render, grads, accumulate, workspace, workspace_bytes, stream);
    return rc ? rc : lfs_gut_view_backward_finish(a, capacity, grads, accumulate, workspace, workspace_bytes, stream);
}

extern "C" int lfs_gut_view_forward(const lfs_gut_step_args* a, int64_t capacity, int64_t assumed_longest, void* workspace, size_t workspace_bytes,
                                    int64_t* host_counts, int64_t stamp, lfs_stream_t stream) {
    int rc = check_args(a, false);
    if (rc) return rc;
    if (!workspace) return LFS_E_INVALID;
    StepWs w;
    if (!step_ws(workspace, a->N, a->image_width, a->image_height, a->tile_size, capacity, w, nullptr)) return LFS_E_INVALID;
    if (workspace_bytes < w.bytes) return LFS_E_WORKSPACE;
    rc = auto_join((hipStream_t)stream);
    if (rc) return rc;
    Front f;
    return enqueue_forward(a, w, capacity, assumed_longest, host_counts, stamp, (hipStream_t)stream, f);
}

// Wait until the counts of the call stamped `stamp` have arrived in pinned host memory (they were written early in the step; by the time the host has
// enqueued the rest this returns at once). 0 = ok, LFS_E_INVALID on timeout.
extern "C" int lfs_gut_step_wait(const int64_t* host_counts, int64_t stamp, double timeout_s, int64_t* n_isects, int64_t* longest) {
    if (!host_counts) return LFS_E_INVALID;
    const volatile int64_t* c = host_counts;
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (c[2] != stamp) {
        if (++spins > 64) {
            std::this_thread::yield();
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return LFS_E_INVALID;
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (n_isects) *n_isects = c[0];
    if (longest) *longest = c[1];
    return LFS_OK;
}
