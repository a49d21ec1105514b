// libtorch wrappers restoring the reference's C++ operator signatures (gsplat/Ops.h,
// fastgs/optimizer adam_api.h / adam.h) over the C ABI of liblfs_gsplat.so. This is the layer a
// LichtFeld-Studio build links instead of `gsplat_backend` + `fastgs_backend`
// (gsplat/CMakeLists.txt:42, fastgs/CMakeLists.txt:26): same checks (CHECK_INPUT = device tensor +
// contiguous), same output allocation (dtype / shape / device from the inputs), current-stream
// launches, c10::Error on failure. PyTorch-ROCm tensors report is_cuda() == true.
#include "../../include/lfs_gsplat_torch.hpp"
#include "../../include/lfs_gsplat.h"

#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <c10/hip/HIPStream.h>
#include <c10/hip/HIPCachingAllocator.h>

#include <atomic>
#include <mutex>
#include <vector>

#define LFS_CHECK_INPUT(x)                                        \
    TORCH_CHECK((x).is_cuda(), #x " must be a CUDA tensor");      \
    TORCH_CHECK((x).is_contiguous(), #x " must be contiguous")
#define LFS_DEVICE_GUARD(t)                                       \
    TORCH_CHECK((t).is_cuda(), #t " must be a CUDA tensor");      \
    const at::hip::OptionalHIPGuardMasqueradingAsCUDA device_guard(device_of(t))

namespace {
lfs_stream_t cur_stream() { return (lfs_stream_t)c10::hip::getCurrentHIPStream().stream(); }
void check_rc(int rc, const char* what) {
    TORCH_CHECK(rc == 0, what, " failed (", rc == LFS_E_INVALID ? "invalid argument" : rc == LFS_E_UNSUPPORTED ? "unsupported configuration"
                                             : rc == LFS_E_WORKSPACE ? "workspace too small" : "HIP error", ", code ", rc, ")");
}
// optional tensor -> pointer; an empty {0} tensor counts as absent (rasterizer.cpp:300-303 passes one for "no background")
template <class T> const T* opt_ptr(const gsplat::OptT& t) { return (t.has_value() && t->defined() && t->numel() > 0) ? t->data_ptr<T>() : nullptr; }
bool present(const gsplat::OptT& t) { return t.has_value() && t->defined() && t->numel() > 0; }

lfs_cameras make_cams(const at::Tensor& viewmats0, const gsplat::OptT& viewmats1, const at::Tensor& Ks, uint32_t W, uint32_t H,
                      gsplat::CameraModelType model, ShutterType rs, const gsplat::OptT& radial, const gsplat::OptT& tangential, const gsplat::OptT& thin) {
    lfs_cameras c{};
    c.C = (uint32_t)Ks.size(0); c.image_width = W; c.image_height = H;
    c.camera_model = (int32_t)model; c.rs_type = (int32_t)rs;
    c.viewmats0 = viewmats0.data_ptr<float>(); c.viewmats1 = opt_ptr<float>(viewmats1); c.Ks = Ks.data_ptr<float>();
    c.radial_coeffs = opt_ptr<float>(radial); c.n_radial = present(radial) ? (int32_t)radial->size(-1) : 0;
    c.tangential_coeffs = opt_ptr<float>(tangential);
    c.thin_prism_coeffs = opt_ptr<float>(thin); c.n_thin_prism = present(thin) ? (int32_t)thin->size(-1) : 0;
    return c;
}
lfs_ut_params make_ut(const UnscentedTransformParameters& u) {
    return lfs_ut_params{u.alpha, u.beta, u.kappa, u.in_image_margin_factor, u.require_all_sigma_points_valid ? 1 : 0};
}
at::Tensor scratch(size_t bytes, const at::Tensor& like) {
    return at::empty({(int64_t)std::max<size_t>(bytes, 256)}, like.options().dtype(at::kByte)); // caching allocator, like CUB_WRAPPER (Common.h:23-30)
}

// Ops.h has no argument through which the forward could hand its staging (camera state, 64-byte records, per-cell lists) to the backward: the backward wrapper
// used to rebuild all of it (raster_pack + raster_cull again: 0.11 ms of a 2.3 ms step on SYN-B). The last forward's workspace is therefore kept, together with
// the identity of every tensor it was built from - storage address AND autograd version counter (an in-place update bumps it; the Adam / add_noise wrappers below
// bump it for their raw-pointer writes) - and a backward called with exactly those tensors, on the same stream, takes the "prepared" entry point. Anything else:
// the self-contained path, as before. One slot per process (a backward matches the forward that directly preceded it: the training step).
// Round 5 (review of round 4): (a) a storage freed and handed out again at the same address with the same version and size between the forward and the backward (the
// ABA hit a caller with short-lived same-sized temporaries could otherwise produce) must miss - round 5 held the keyed tensors, round 6 watches their storages (below); (b) the slot is
// guarded by a mutex, and a backward MOVES the workspace out under the lock: the reference's viewer thread may render through these wrappers
// (rendering_pipeline.cpp:79) while the training thread is between its forward and its backward (render_mutex_ only covers post_backward / step, trainer.cpp:741), and
// libtorch runs a C++ autograd Function's backward on the engine's device thread, not on the thread that ran the forward - which is also why the slot cannot be
// thread_local. A forward of another thread in between simply replaces the slot: the training backward then misses and rebuilds its staging (correct, 0.1 ms slower).
struct TensorId {
    const void* ptr = nullptr; uint32_t version = 0; int64_t numel = -1;
    bool operator==(const TensorId& o) const { return ptr == o.ptr && version == o.version && numel == o.numel; }
};
TensorId tid(const at::Tensor& t) { return t.defined() ? TensorId{t.data_ptr(), (uint32_t)t._version(), t.numel()} : TensorId{}; }
TensorId tid(const gsplat::OptT& t) { return (t.has_value() && t->defined()) ? tid(*t) : TensorId{}; }
struct RasterKey {
    TensorId t[15]; uint32_t W = 0, H = 0, tile = 0; int cam = 0, shutter = 0; lfs_stream_t stream = nullptr;
    bool operator==(const RasterKey& o) const {
        for (int i = 0; i < 15; ++i) if (!(t[i] == o.t[i])) return false;
        return W == o.W && H == o.H && tile == o.tile && cam == o.cam && shutter == o.shutter && stream == o.stream;
    }
};
// Round 6 (review of round 5 / ADVICE): the slot no longer HOLDS its keyed tensors - it watches their storages through weak references. A storage that died between the
// forward and the backward (whose address the allocator may have handed out again: the ABA case of round 4) makes the key stale: miss. Nothing a forward-only caller
// passed in - a model's activated copies, an autograd graph hanging off `colors` - outlives the caller's own references any more; the slot owns the workspace only.
typedef c10::weak_intrusive_ptr<c10::StorageImpl> WeakStorage;
struct RasterCache {
    std::mutex mu; bool valid = false; RasterKey key; at::Tensor ws; std::vector<WeakStorage> watched;
    uint64_t n_store = 0, n_hit = 0, n_miss = 0, n_skipped = 0;
    void store(RasterKey k, at::Tensor w, std::vector<WeakStorage> h) {
        at::Tensor old_ws;   // (released outside the lock)
        { std::lock_guard<std::mutex> g(mu); old_ws = std::move(ws); valid = true; key = k; ws = std::move(w); watched = std::move(h); ++n_store; }
    }
    // -> the forward's workspace if `k` is the stored key and every watched storage is still alive (the slot is emptied either way: one backward per forward)
    at::Tensor take(const RasterKey& k, bool count = true) {
        at::Tensor out, old_ws;
        {
            std::lock_guard<std::mutex> g(mu);
            bool hit = valid && key == k;
            for (const WeakStorage& w : watched) hit = hit && !w.expired();
            if (hit) out = std::move(ws); else old_ws = std::move(ws);
            ws = at::Tensor(); valid = false; watched.clear();
            if (count) { if (hit) ++n_hit; else ++n_miss; }
        }
        return out;
    }
    void clear() { (void)take(RasterKey{}, false); }   // (an all-default key never matches a stored one: the workspace is released outside the lock)
};
// never destroyed: a static destructor would release device memory after the HIP runtime may already be gone (process exit reclaims it)
RasterCache& raster_cache() { static RasterCache* c = new RasterCache; return *c; }
// Callers whose tensors do not carry requires_grad although a backward follows (raw-pointer style bindings: the pybind test module's explicit forward + backward
// pairs) switch the requires_grad test of the forward wrapper off: lfs::torch_keep_raster_staging(true).
std::atomic<bool> g_keep_staging_always{false};
std::vector<WeakStorage> raster_watched(const at::Tensor& means, const at::Tensor& quats, const at::Tensor& scales, const at::Tensor& colors, const at::Tensor& opacities,
                                        const gsplat::OptT& backgrounds, const gsplat::OptT& masks, const at::Tensor& viewmats0, const gsplat::OptT& viewmats1, const at::Tensor& Ks,
                                        const gsplat::OptT& radial, const gsplat::OptT& tangential, const gsplat::OptT& thin, const at::Tensor& tile_offsets, const at::Tensor& flatten_ids) {
    std::vector<WeakStorage> h;
    auto watch = [&](const at::Tensor& t) { if (t.defined() && t.has_storage()) h.push_back(t.storage().getWeakStorageImpl()); };
    for (const at::Tensor* t : {&means, &quats, &scales, &colors, &opacities, &viewmats0, &Ks, &tile_offsets, &flatten_ids}) watch(*t);
    for (const gsplat::OptT* o : {&backgrounds, &masks, &viewmats1, &radial, &tangential, &thin}) if (o->has_value()) watch(**o);
    return h;
}
// Can a backward follow this forward at all? Not in inference mode, and not when none of the five differentiable operands requires a gradient (an evaluation or viewer
// render of detached tensors): such a forward parks nothing and releases what an earlier one parked. The reference's rasterizer_autograd.cpp calls the wrapper from
// inside its autograd Function's forward(), where grad mode is switched off but the operands are the caller's variables and still say requires_grad(): the training
// step is recognised by that, not by the grad mode. (A caller that renders parameter tensors under NoGradGuard cannot be told apart from the training step from in
// here; its workspace is replaced by the next forward and can be dropped at once with lfs::torch_raster_staging_clear().)
bool backward_may_follow(const at::Tensor& means, const at::Tensor& quats, const at::Tensor& scales, const at::Tensor& colors, const at::Tensor& opacities) {
    if (c10::InferenceMode::is_enabled()) return false;
    return means.requires_grad() || quats.requires_grad() || scales.requires_grad() || colors.requires_grad() || opacities.requires_grad();
}
RasterKey raster_key(const at::Tensor& means, const at::Tensor& quats, const at::Tensor& scales, const at::Tensor& colors, const at::Tensor& opacities,
                     const gsplat::OptT& backgrounds, const gsplat::OptT& masks, uint32_t W, uint32_t H, uint32_t tile, const at::Tensor& viewmats0,
                     const gsplat::OptT& viewmats1, const at::Tensor& Ks, int cam, int shutter, const gsplat::OptT& radial, const gsplat::OptT& tangential,
                     const gsplat::OptT& thin, const at::Tensor& tile_offsets, const at::Tensor& flatten_ids) {
    RasterKey k;
    // round 6 (review of round 5): the thin-prism coefficients are part of the key like every other operand - two forwards that differ only in them are two keys
    const TensorId ids[15] = {tid(means), tid(quats), tid(scales), tid(colors), tid(opacities), tid(backgrounds), tid(masks), tid(viewmats0), tid(viewmats1), tid(Ks),
                              tid(radial), tid(tangential), tid(thin), tid(tile_offsets), tid(flatten_ids)};
    for (int i = 0; i < 15; ++i) k.t[i] = ids[i];
    k.W = W; k.H = H; k.tile = tile; k.cam = cam; k.shutter = shutter; k.stream = cur_stream();
    return k;
}
void bump(at::Tensor& t) { if (t.defined()) t.unsafeGetTensorImpl()->bump_version(); } // an in-place update through a raw pointer: visible to autograd and to the cache above
} // namespace

torch::Tensor UnscentedTransformParameters::to_tensor() const {
    return torch::tensor({alpha, beta, kappa, in_image_margin_factor, static_cast<float>(require_all_sigma_points_valid)},
                         torch::TensorOptions().dtype(torch::kFloat32));
}
UnscentedTransformParameters UnscentedTransformParameters::from_tensor(const torch::Tensor& t) {
    TORCH_CHECK(t.dim() == 1 && t.size(0) == 5, "UnscentedTransformParameters must be a 1D tensor of size 5");
    UnscentedTransformParameters u;
    u.alpha = t[0].item<float>(); u.beta = t[1].item<float>(); u.kappa = t[2].item<float>();
    u.in_image_margin_factor = t[3].item<float>(); u.require_all_sigma_points_valid = t[4].item<bool>();
    return u;
}

namespace gsplat {

at::Tensor spherical_harmonics_fwd(const uint32_t degrees_to_use, const at::Tensor dirs, const at::Tensor coeffs, const OptT masks) {
    LFS_DEVICE_GUARD(dirs);
    LFS_CHECK_INPUT(dirs); LFS_CHECK_INPUT(coeffs);
    if (present(masks)) { LFS_CHECK_INPUT(masks.value()); }
    TORCH_CHECK(coeffs.size(-1) == 3, "coeffs must have last dimension 3");
    TORCH_CHECK(dirs.size(-1) == 3, "dirs must have last dimension 3");
    at::Tensor colors = at::empty_like(dirs);
    check_rc(lfs_spherical_harmonics_fwd((uint32_t)(dirs.numel() / 3), (uint32_t)coeffs.size(-2), degrees_to_use, dirs.data_ptr<float>(),
                                         coeffs.data_ptr<float>(), (const uint8_t*)opt_ptr<bool>(masks), colors.data_ptr<float>(), cur_stream()),
             "spherical_harmonics_fwd");
    return colors;
}

std::tuple<at::Tensor, at::Tensor> spherical_harmonics_bwd(const uint32_t K, const uint32_t degrees_to_use, const at::Tensor dirs,
                                                           const at::Tensor coeffs, const OptT masks, const at::Tensor v_colors, bool compute_v_dirs) {
    LFS_DEVICE_GUARD(dirs);
    LFS_CHECK_INPUT(dirs); LFS_CHECK_INPUT(coeffs); LFS_CHECK_INPUT(v_colors);
    if (present(masks)) { LFS_CHECK_INPUT(masks.value()); }
    TORCH_CHECK(v_colors.size(-1) == 3, "v_colors must have last dimension 3");
    TORCH_CHECK(coeffs.size(-1) == 3, "coeffs must have last dimension 3");
    TORCH_CHECK(dirs.size(-1) == 3, "dirs must have last dimension 3");
    TORCH_CHECK(coeffs.size(-2) == (int64_t)K, "K does not match coeffs");
    at::Tensor v_coeffs = at::empty_like(coeffs); // fully written by the kernel
    at::Tensor v_dirs;
    if (compute_v_dirs) v_dirs = at::empty_like(dirs);
    check_rc(lfs_spherical_harmonics_bwd((uint32_t)(dirs.numel() / 3), K, degrees_to_use, dirs.data_ptr<float>(), coeffs.data_ptr<float>(),
                                         (const uint8_t*)opt_ptr<bool>(masks), v_colors.data_ptr<float>(), v_coeffs.data_ptr<float>(),
                                         compute_v_dirs ? v_dirs.data_ptr<float>() : nullptr, cur_stream()),
             "spherical_harmonics_bwd");
    return std::make_tuple(v_coeffs, v_dirs);
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> intersect_tile(const at::Tensor means2d, const at::Tensor radii, const at::Tensor depths,
                                                              const OptT camera_ids, const OptT gaussian_ids, const uint32_t C,
                                                              const uint32_t tile_size, const uint32_t tile_width, const uint32_t tile_height,
                                                              const bool sort) {
    LFS_DEVICE_GUARD(means2d);
    LFS_CHECK_INPUT(means2d); LFS_CHECK_INPUT(radii); LFS_CHECK_INPUT(depths);
    TORCH_CHECK(means2d.dim() == 3, "packed mode is not supported (the reference trainer never uses it, rasterizer.cpp:56)");
    (void)camera_ids; (void)gaussian_ids;
    const uint32_t N = (uint32_t)means2d.size(1);
    at::Tensor tiles_per_gauss = at::empty_like(depths, depths.options().dtype(at::kInt));
    const size_t ws_bytes = lfs_intersect_tile_workspace_bytes(C, N, tile_width, tile_height);
    at::Tensor ws = scratch(ws_bytes, depths);
    // The one host sync of the path, at the place the reference has it (Intersect.cpp:75-76: it allocates to the count). Round 6: the scan kernel writes
    // {n_isects, longest tile list, stamp} straight into pinned host memory and the host spins on its own stamp - no device-to-host copy, no stream
    // synchronisation behind it (was n_dev.item<int64_t>(): a copy kernel + hipStreamSynchronize in the bubble the GPU idles through). The longest list also tells
    // the per-tile sort which size classes occur. One slot per host thread, never freed (a static tensor's destructor would run after the HIP runtime's).
    static thread_local at::Tensor* const counts = new at::Tensor(at::zeros({3}, at::TensorOptions().dtype(at::kLong).pinned_memory(true)));
    static thread_local int64_t stamp = 0;
    stamp += 1;
    int64_t* const host = counts->data_ptr<int64_t>();
    check_rc(lfs_intersect_tile_count_ex(C, N, means2d.data_ptr<float>(), radii.data_ptr<int32_t>(), tile_size, tile_width, tile_height,
                                         tiles_per_gauss.data_ptr<int32_t>(), host, host + 1, nullptr, 0u, host + 2, stamp, ws.data_ptr(), (size_t)ws.numel(), cur_stream()),
             "intersect_tile(count)");
    int64_t n_isects = 0, longest = -1;
    check_rc(lfs_gut_step_wait(host, stamp, 30.0, &n_isects, &longest), "intersect_tile(count read-back)");
    at::Tensor isect_ids = at::empty({n_isects}, depths.options().dtype(at::kLong));
    at::Tensor flatten_ids = at::empty({n_isects}, depths.options().dtype(at::kInt));
    at::Tensor binned = at::empty({sort ? n_isects : 0}, depths.options().dtype(at::kLong)); // the two-pass scatter's intermediate (row-binned) array
    check_rc(lfs_intersect_tile_emit_ex(C, N, means2d.data_ptr<float>(), radii.data_ptr<int32_t>(), depths.data_ptr<float>(), tile_size, tile_width,
                                        tile_height, sort ? 1 : 0, n_isects, tiles_per_gauss.data_ptr<int32_t>(),
                                        n_isects ? isect_ids.data_ptr<int64_t>() : nullptr, n_isects ? flatten_ids.data_ptr<int32_t>() : nullptr, nullptr,
                                        (sort && n_isects) ? binned.data_ptr<int64_t>() : nullptr, longest, ws.data_ptr(), (size_t)ws.numel(), cur_stream()),
             "intersect_tile(emit)");
    return std::make_tuple(tiles_per_gauss, isect_ids, flatten_ids);
}

at::Tensor intersect_offset(const at::Tensor isect_ids, const uint32_t C, const uint32_t tile_width, const uint32_t tile_height) {
    LFS_DEVICE_GUARD(isect_ids);
    LFS_CHECK_INPUT(isect_ids);
    at::Tensor offsets = at::empty({(int64_t)C, (int64_t)tile_height, (int64_t)tile_width}, isect_ids.options().dtype(at::kInt));
    check_rc(lfs_intersect_offset(isect_ids.size(0), isect_ids.numel() ? isect_ids.data_ptr<int64_t>() : nullptr, C, tile_width, tile_height,
                                  offsets.data_ptr<int32_t>(), cur_stream()), "intersect_offset");
    return offsets;
}

at::Tensor quats_to_rotmats(const at::Tensor quats) {
    LFS_DEVICE_GUARD(quats);
    LFS_CHECK_INPUT(quats);
    at::Tensor rotmats = at::empty({quats.size(0), 3, 3}, quats.options());
    check_rc(lfs_quats_to_rotmats((uint32_t)quats.size(0), quats.data_ptr<float>(), rotmats.data_ptr<float>(), cur_stream()), "quats_to_rotmats");
    return rotmats;
}

std::tuple<at::Tensor, at::Tensor> relocation(at::Tensor opacities, at::Tensor scales, at::Tensor ratios, at::Tensor binoms, const int n_max) {
    LFS_DEVICE_GUARD(opacities);
    LFS_CHECK_INPUT(opacities); LFS_CHECK_INPUT(scales); LFS_CHECK_INPUT(ratios); LFS_CHECK_INPUT(binoms);
    at::Tensor new_opacities = at::empty_like(opacities), new_scales = at::empty_like(scales);
    check_rc(lfs_relocation((uint32_t)opacities.size(0), opacities.data_ptr<float>(), scales.data_ptr<float>(), ratios.data_ptr<int32_t>(),
                            binoms.data_ptr<float>(), n_max, new_opacities.data_ptr<float>(), new_scales.data_ptr<float>(), cur_stream()), "relocation");
    return std::make_tuple(new_opacities, new_scales);
}

void add_noise(at::Tensor raw_opacities, at::Tensor raw_scales, at::Tensor raw_quats, at::Tensor noise, at::Tensor means, const float current_lr) {
    LFS_DEVICE_GUARD(raw_opacities);
    LFS_CHECK_INPUT(raw_opacities); LFS_CHECK_INPUT(raw_scales); LFS_CHECK_INPUT(raw_quats); LFS_CHECK_INPUT(noise); LFS_CHECK_INPUT(means);
    check_rc(lfs_add_noise((uint32_t)raw_opacities.size(0), raw_opacities.data_ptr<float>(), raw_scales.data_ptr<float>(), raw_quats.data_ptr<float>(),
                           noise.data_ptr<float>(), means.data_ptr<float>(), current_lr, cur_stream()), "add_noise");
    bump(means);
}

std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> projection_ut_3dgs_fused(
    const at::Tensor means, const at::Tensor quats, const at::Tensor scales, const OptT opacities, const at::Tensor viewmats0,
    const OptT viewmats1, const at::Tensor Ks, const uint32_t image_width, const uint32_t image_height, const float eps2d,
    const float near_plane, const float far_plane, const float radius_clip, const bool calc_compensations,
    const CameraModelType camera_model, const UnscentedTransformParameters ut_params, ShutterType rs_type,
    const OptT radial_coeffs, const OptT tangential_coeffs, const OptT thin_prism_coeffs) {
    LFS_DEVICE_GUARD(means);
    LFS_CHECK_INPUT(means); LFS_CHECK_INPUT(quats); LFS_CHECK_INPUT(scales); LFS_CHECK_INPUT(viewmats0); LFS_CHECK_INPUT(Ks);
    for (const OptT* o : {&opacities, &viewmats1, &radial_coeffs, &tangential_coeffs, &thin_prism_coeffs})
        if (present(*o)) { LFS_CHECK_INPUT(o->value()); }
    const int64_t N = means.size(0), C = Ks.size(0);
    at::Tensor radii = at::empty({C, N, 2}, means.options().dtype(at::kInt));
    at::Tensor means2d = at::empty({C, N, 2}, means.options());
    at::Tensor depths = at::empty({C, N}, means.options());
    at::Tensor conics = at::empty({C, N, 3}, means.options());
    at::Tensor compensations;
    if (calc_compensations) compensations = at::zeros({C, N}, means.options());
    const lfs_cameras cams = make_cams(viewmats0, viewmats1, Ks, image_width, image_height, camera_model, rs_type, radial_coeffs, tangential_coeffs, thin_prism_coeffs);
    const lfs_ut_params ut = make_ut(ut_params);
    check_rc(lfs_projection_ut_3dgs_fused((uint32_t)N, means.data_ptr<float>(), quats.data_ptr<float>(), scales.data_ptr<float>(), opt_ptr<float>(opacities),
                                          &cams, eps2d, near_plane, far_plane, radius_clip, &ut, radii.data_ptr<int32_t>(), means2d.data_ptr<float>(),
                                          depths.data_ptr<float>(), conics.data_ptr<float>(), calc_compensations ? compensations.data_ptr<float>() : nullptr,
                                          cur_stream()), "projection_ut_3dgs_fused");
    return std::make_tuple(radii, means2d, depths, conics, compensations);
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> rasterize_to_pixels_from_world_3dgs_fwd(
    const at::Tensor means, const at::Tensor quats, const at::Tensor scales, const at::Tensor colors, const at::Tensor opacities,
    const OptT backgrounds, const OptT masks, const uint32_t image_width, const uint32_t image_height, const uint32_t tile_size,
    const at::Tensor viewmats0, const OptT viewmats1, const at::Tensor Ks, const CameraModelType camera_model,
    const UnscentedTransformParameters ut_params, ShutterType rs_type, const OptT radial_coeffs, const OptT tangential_coeffs,
    const OptT thin_prism_coeffs, const at::Tensor tile_offsets, const at::Tensor flatten_ids) {
    LFS_DEVICE_GUARD(means);
    LFS_CHECK_INPUT(means); LFS_CHECK_INPUT(quats); LFS_CHECK_INPUT(scales); LFS_CHECK_INPUT(colors); LFS_CHECK_INPUT(opacities);
    LFS_CHECK_INPUT(tile_offsets); LFS_CHECK_INPUT(flatten_ids);
    if (present(backgrounds)) { LFS_CHECK_INPUT(backgrounds.value()); }
    if (present(masks)) { LFS_CHECK_INPUT(masks.value()); }
    TORCH_CHECK(opacities.dim() == 2, "packed mode is not supported");
    const int64_t C = tile_offsets.size(0), N = means.size(0), channels = colors.size(-1);
    at::Tensor renders = at::empty({C, (int64_t)image_height, (int64_t)image_width, channels}, means.options());
    at::Tensor alphas = at::empty({C, (int64_t)image_height, (int64_t)image_width, 1}, means.options());
    at::Tensor last_ids = at::empty({C, (int64_t)image_height, (int64_t)image_width}, means.options().dtype(at::kInt));
    const lfs_cameras cams = make_cams(viewmats0, viewmats1, Ks, image_width, image_height, camera_model, rs_type, radial_coeffs, tangential_coeffs, thin_prism_coeffs);
    const lfs_ut_params ut = make_ut(ut_params);
    at::Tensor ws = scratch(lfs_rasterize_workspace_bytes((uint32_t)C, (uint32_t)N, (uint32_t)channels, (uint32_t)image_width, (uint32_t)image_height, (uint32_t)tile_size, flatten_ids.numel()), means);
    const int rc = lfs_rasterize_to_pixels_from_world_3dgs_fwd(
        (uint32_t)N, (uint32_t)channels, means.data_ptr<float>(), quats.data_ptr<float>(), scales.data_ptr<float>(), colors.data_ptr<float>(),
        opacities.data_ptr<float>(), opt_ptr<float>(backgrounds), (const uint8_t*)opt_ptr<bool>(masks), &cams, tile_size, &ut,
        tile_offsets.data_ptr<int32_t>(), flatten_ids.numel() ? flatten_ids.data_ptr<int32_t>() : nullptr, flatten_ids.size(0),
        renders.data_ptr<float>(), alphas.data_ptr<float>(), last_ids.data_ptr<int32_t>(), ws.data_ptr(), (size_t)ws.numel(), cur_stream());
    TORCH_CHECK(rc != LFS_E_UNSUPPORTED, "Unsupported number of channels: ", channels); // Rasterization.cpp:127
    check_rc(rc, "rasterize_to_pixels_from_world_3dgs_fwd");
    // what the backward of THIS forward may reuse (see RasterCache)
    if (g_keep_staging_always.load() || backward_may_follow(means, quats, scales, colors, opacities))
        raster_cache().store(raster_key(means, quats, scales, colors, opacities, backgrounds, masks, image_width, image_height, tile_size, viewmats0, viewmats1, Ks,
                                        (int)camera_model, (int)rs_type, radial_coeffs, tangential_coeffs, thin_prism_coeffs, tile_offsets, flatten_ids),
                             ws, raster_watched(means, quats, scales, colors, opacities, backgrounds, masks, viewmats0, viewmats1, Ks, radial_coeffs, tangential_coeffs, thin_prism_coeffs, tile_offsets, flatten_ids));
    else {
        raster_cache().clear();   // a forward no backward can follow: nothing is parked, and what an earlier forward parked is released
        std::lock_guard<std::mutex> g(raster_cache().mu); ++raster_cache().n_skipped;
    }
    return std::make_tuple(renders, alphas, last_ids);
}

std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> rasterize_to_pixels_from_world_3dgs_bwd(
    const at::Tensor means, const at::Tensor quats, const at::Tensor scales, const at::Tensor colors, const at::Tensor opacities,
    const OptT backgrounds, const OptT masks, const uint32_t image_width, const uint32_t image_height, const uint32_t tile_size,
    const at::Tensor viewmats0, const OptT viewmats1, const at::Tensor Ks, const CameraModelType camera_model,
    const UnscentedTransformParameters ut_params, ShutterType rs_type, const OptT radial_coeffs, const OptT tangential_coeffs,
    const OptT thin_prism_coeffs, const at::Tensor tile_offsets, const at::Tensor flatten_ids, const at::Tensor render_alphas,
    const at::Tensor last_ids, const at::Tensor v_render_colors, const at::Tensor v_render_alphas) {
    LFS_DEVICE_GUARD(means);
    LFS_CHECK_INPUT(means); LFS_CHECK_INPUT(quats); LFS_CHECK_INPUT(scales); LFS_CHECK_INPUT(colors); LFS_CHECK_INPUT(opacities);
    LFS_CHECK_INPUT(tile_offsets); LFS_CHECK_INPUT(flatten_ids); LFS_CHECK_INPUT(render_alphas); LFS_CHECK_INPUT(last_ids);
    LFS_CHECK_INPUT(v_render_colors); LFS_CHECK_INPUT(v_render_alphas);
    if (present(backgrounds)) { LFS_CHECK_INPUT(backgrounds.value()); }
    if (present(masks)) { LFS_CHECK_INPUT(masks.value()); }
    const int64_t C = tile_offsets.size(0), N = means.size(0), channels = colors.size(-1);
    at::Tensor v_means = at::empty_like(means), v_quats = at::empty_like(quats), v_scales = at::empty_like(scales);
    at::Tensor v_colors = at::empty_like(colors), v_opacities = at::empty_like(opacities);
    const lfs_cameras cams = make_cams(viewmats0, viewmats1, Ks, image_width, image_height, camera_model, rs_type, radial_coeffs, tangential_coeffs, thin_prism_coeffs);
    const lfs_ut_params ut = make_ut(ut_params);
    at::Tensor ws = raster_cache().take(raster_key(means, quats, scales, colors, opacities, backgrounds, masks, image_width, image_height, tile_size, viewmats0, viewmats1,
                                                   Ks, (int)camera_model, (int)rs_type, radial_coeffs, tangential_coeffs, thin_prism_coeffs, tile_offsets, flatten_ids));
    const bool hit = ws.defined();   // (the slot is empty now - workspace moved out, the held operands released: one backward per forward)
    if (!hit) ws = scratch(lfs_rasterize_workspace_bytes((uint32_t)C, (uint32_t)N, (uint32_t)channels, (uint32_t)image_width, (uint32_t)image_height, (uint32_t)tile_size, flatten_ids.numel()), means);
    const int rc = (hit ? lfs_rasterize_to_pixels_from_world_3dgs_bwd_prepared : lfs_rasterize_to_pixels_from_world_3dgs_bwd)(
        (uint32_t)N, (uint32_t)channels, means.data_ptr<float>(), quats.data_ptr<float>(), scales.data_ptr<float>(), colors.data_ptr<float>(),
        opacities.data_ptr<float>(), opt_ptr<float>(backgrounds), (const uint8_t*)opt_ptr<bool>(masks), &cams, tile_size, &ut,
        tile_offsets.data_ptr<int32_t>(), flatten_ids.numel() ? flatten_ids.data_ptr<int32_t>() : nullptr, flatten_ids.size(0),
        render_alphas.data_ptr<float>(), last_ids.data_ptr<int32_t>(), v_render_colors.data_ptr<float>(), v_render_alphas.data_ptr<float>(),
        v_means.data_ptr<float>(), v_quats.data_ptr<float>(), v_scales.data_ptr<float>(), v_colors.data_ptr<float>(), v_opacities.data_ptr<float>(),
        ws.data_ptr(), (size_t)ws.numel(), cur_stream());
    TORCH_CHECK(rc != LFS_E_UNSUPPORTED, "Unsupported number of channels: ", channels);
    check_rc(rc, "rasterize_to_pixels_from_world_3dgs_bwd");
    return std::make_tuple(v_means, v_quats, v_scales, v_colors, v_opacities);
}

} // namespace gsplat

namespace fast_gs::optimizer {
void adam_step(float* param, float* exp_avg, float* exp_avg_sq, const float* param_grad, const int n_elements, const float lr,
               const float beta1, const float beta2, const float eps, const float bias_correction1_rcp, const float bias_correction2_sqrt_rcp) {
    check_rc(lfs_adam_step(param, exp_avg, exp_avg_sq, param_grad, n_elements, lr, beta1, beta2, eps, bias_correction1_rcp, bias_correction2_sqrt_rcp,
                           cur_stream()), "adam_step");
}
void adam_step_wrapper(torch::Tensor& param, torch::Tensor& exp_avg, torch::Tensor& exp_avg_sq, const torch::Tensor& param_grad,
                       const float lr, const float beta1, const float beta2, const float eps, const float bias_correction1_rcp,
                       const float bias_correction2_sqrt_rcp) {
    adam_step(param.data_ptr<float>(), exp_avg.data_ptr<float>(), exp_avg_sq.data_ptr<float>(), param_grad.data_ptr<float>(), (int)param.numel(),
              lr, beta1, beta2, eps, bias_correction1_rcp, bias_correction2_sqrt_rcp);
    bump(param); bump(exp_avg); bump(exp_avg_sq);
}
} // namespace fast_gs::optimizer

// ---------------------------------------------------------------------------------------------------------
// lfs::GutTrainStep: the C++ training step (csrc/gut_step.hip) for a libtorch caller
// ---------------------------------------------------------------------------------------------------------
namespace lfs {
void torch_raster_staging_clear() { raster_cache().clear(); }
void torch_keep_raster_staging(bool always) { g_keep_staging_always.store(always); }
RasterStagingStats torch_raster_staging_stats() {
    RasterCache& c = raster_cache();
    std::lock_guard<std::mutex> g(c.mu);
    return RasterStagingStats{c.n_store, c.n_hit, c.n_miss, c.n_skipped, c.ws.defined() ? (uint64_t)c.ws.numel() : 0u};
}
GutTrainStep::GutTrainStep(uint32_t tile_size, int64_t initial_capacity) : tile_(tile_size), capacity_(initial_capacity) {}

void GutTrainStep::ensure(uint32_t N, uint32_t W, uint32_t H, const torch::Tensor& like) {
    if (capacity_ <= 0) capacity_ = std::max<int64_t>(4 * int64_t(N), 1 << 16);   // first guess; the first step corrects it
    const uint32_t flags = lfs_get_debug_flags();
    if (ws_.defined() && N == N_ && W == W_ && H == H_ && cap_built_ == capacity_ && flags == flags_) return;
    colours_for_.valid = false;   // a new layout (or block): whatever colours the last tail prepared are not where the next step would look for them
    lfs_gut_step_layout lay{};
    check_rc(lfs_gut_step_layout_for(N, W, H, tile_, capacity_, &lay), "gut_step_layout_for");
    if (!ws_.defined() || (size_t)ws_.numel() < lay.bytes || ws_.device() != like.device()) {
        ws_ = at::Tensor();   // release the old block before the larger one is requested
        ws_ = at::empty({(int64_t)lay.bytes}, like.options().dtype(at::kByte));
    }
    if (!counts_.defined()) counts_ = at::zeros({3}, at::TensorOptions().dtype(at::kLong).pinned_memory(true));
    N_ = N; W_ = W; H_ = H; cap_built_ = capacity_; flags_ = flags;
    off_render_ = lay.render; off_alpha_ = lay.alpha; off_radii_ = lay.radii; ws_bytes_ = lay.bytes;
}

int64_t GutTrainStep::step(torch::Tensor& means, torch::Tensor& sh0, torch::Tensor& shN, torch::Tensor& raw_scales, torch::Tensor& raw_quats,
                           torch::Tensor& raw_opacities, const std::array<AdamGroupState, 6>& adam, uint32_t sh_degree, const torch::Tensor& viewmat,
                           const torch::Tensor& K, uint32_t image_width, uint32_t image_height, const at::optional<torch::Tensor>& background,
                           const torch::Tensor& target_chw, float loss_weight, torch::Tensor& loss, float scale_reg, float opacity_reg,
                           const at::optional<torch::Tensor>& next_viewmat) {
    LFS_DEVICE_GUARD(means);
    LFS_CHECK_INPUT(means); LFS_CHECK_INPUT(sh0); LFS_CHECK_INPUT(shN); LFS_CHECK_INPUT(raw_scales); LFS_CHECK_INPUT(raw_quats); LFS_CHECK_INPUT(raw_opacities);
    LFS_CHECK_INPUT(viewmat); LFS_CHECK_INPUT(K); LFS_CHECK_INPUT(target_chw); LFS_CHECK_INPUT(loss);
    TORCH_CHECK(shN.dim() == 3 && shN.size(1) > 0, "GutTrainStep needs higher-degree SH coefficients (shN [N,K-1,3], K > 1)");
    TORCH_CHECK(target_chw.numel() == 3 * int64_t(image_width) * image_height, "target must be [3,H,W]");
    const uint32_t N = (uint32_t)means.size(0);
    lfs_gut_step_args a{};
    a.N = N; a.K = 1 + (uint32_t)shN.size(1); a.sh_degree = sh_degree; a.image_width = image_width; a.image_height = image_height; a.tile_size = tile_;
    a.means = means.data_ptr<float>(); a.sh0 = sh0.data_ptr<float>(); a.shN = shN.data_ptr<float>();
    a.raw_scales = raw_scales.data_ptr<float>(); a.raw_quats = raw_quats.data_ptr<float>(); a.raw_opacities = raw_opacities.data_ptr<float>();
    for (int k = 0; k < 6; ++k) {
        LFS_CHECK_INPUT(adam[k].exp_avg); LFS_CHECK_INPUT(adam[k].exp_avg_sq);
        a.exp_avg[k] = adam[k].exp_avg.data_ptr<float>(); a.exp_avg_sq[k] = adam[k].exp_avg_sq.data_ptr<float>();
        const float sc[6] = {adam[k].lr, adam[k].beta1, adam[k].beta2, adam[k].eps, adam[k].bias_correction1_rcp, adam[k].bias_correction2_sqrt_rcp};
        for (int j = 0; j < 6; ++j) a.adam[k][j] = sc[j];
    }
    a.viewmat = viewmat.data_ptr<float>(); a.Kmat = K.data_ptr<float>();
    a.background = opt_ptr<float>(background); a.target_chw = target_chw.data_ptr<float>();
    a.loss_weight = loss_weight; a.scale_reg = scale_reg; a.opacity_reg = opacity_reg; a.loss = loss.data_ptr<float>();
    // Round 6: the fused tail (lfs_gut_train_step_ex). The colours in the workspace are "ready" for this call when the previous call's tail evaluated them for exactly
    // this view tensor from exactly these parameter tensors and nothing has written to either since (storage address + autograd version: every in-place torch op and the
    // raw-pointer wrappers of this file bump the version; this step's own writes are recorded AFTER the call, below).
    const float* next_vm = nullptr;
    if (next_viewmat.has_value() && next_viewmat->defined() && a.K <= 16) {   // (K > 16: the tail falls back to the separate passes and prepares nothing)
        LFS_CHECK_INPUT(next_viewmat.value());
        TORCH_CHECK(next_viewmat->numel() == 16, "next_viewmat must be [4,4]");
        next_vm = next_viewmat->data_ptr<float>();
    }
    const torch::Tensor* const watched[3] = {&means, &sh0, &shN};
    auto describes = [&](const ColoursFor& c, const torch::Tensor& vm) {
        bool same = c.valid && c.viewmat == vm.data_ptr() && c.viewmat_version == (uint32_t)vm._version() && c.ws == ws_.data_ptr() && c.N == N && c.K == a.K && c.degree == sh_degree;
        for (int k = 0; k < 3; ++k) same = same && c.param[k] == watched[k]->data_ptr() && c.param_version[k] == (uint32_t)watched[k]->_version();
        return same;
    };
    for (int attempt = 0; attempt < 4; ++attempt) {
        ensure(N, image_width, image_height, means);
        ++stamp_;
        int64_t* counts = counts_.data_ptr<int64_t>();
        const bool ready = describes(colours_for_, viewmat);
        colours_for_.valid = false;   // whatever happens below, the colours of THIS view are consumed / overwritten
        check_rc(lfs_gut_train_step_ex(&a, next_vm, ready ? 1 : 0, capacity_, assumed_longest_, ws_.data_ptr(), (size_t)ws_.numel(), counts, stamp_, cur_stream()), "gut_train_step_ex");
        // the counts were written by the scan kernel early in the step: by now they have long arrived (no GPU idle time behind this wait)
        check_rc(lfs_gut_step_wait(counts, stamp_, 30.0, &n_isects_, &longest_), "gut_step_wait");
        if (lfs_gut_step_fits(n_isects_, longest_, capacity_, assumed_longest_)) {
            colours_saved_ += ready ? 1 : 0;
            for (torch::Tensor* t : {&means, &sh0, &shN, &raw_scales, &raw_quats, &raw_opacities}) bump(*t);   // in-place updates through raw pointers: visible to autograd / the staging slot
            if (next_vm != nullptr) {
                ColoursFor& c = colours_for_;
                c.valid = true; c.viewmat = next_viewmat->data_ptr(); c.viewmat_version = (uint32_t)next_viewmat->_version(); c.ws = ws_.data_ptr();
                c.N = N; c.K = a.K; c.degree = sh_degree;
                for (int k = 0; k < 3; ++k) { c.param[k] = watched[k]->data_ptr(); c.param_version[k] = (uint32_t)watched[k]->_version(); }
            }
            if (double(n_isects_) > 0.92 * double(capacity_)) capacity_ = int64_t(double(n_isects_) * 1.25) + 1024;   // stay ahead of a growing scene
            const int64_t limit = assumed_longest_ <= 1024 ? 1024 : assumed_longest_ <= 4096 ? 4096 : assumed_longest_ <= 16384 ? 16384 : (int64_t(1) << 62);
            if (double(longest_) > 0.92 * double(limit)) assumed_longest_ = std::max<int64_t>(assumed_longest_, int64_t(double(longest_) * 1.25));
            return n_isects_;
        }
        // the attempt did not fit: its tail returned without writing anything, so colours that were ready still are - unless ensure() replaces the workspace (the key holds its address)
        if (ready) { colours_for_.valid = true; }
        ++retries_;
        capacity_ = std::max<int64_t>(capacity_, int64_t(double(n_isects_) * 1.25) + 1024);
        assumed_longest_ = std::max<int64_t>(assumed_longest_, int64_t(double(longest_) * 1.25));
    }
    TORCH_CHECK(false, "GutTrainStep: the step did not fit its workspace after 4 attempts");
    return -1;
}

torch::Tensor GutTrainStep::render() const {
    TORCH_CHECK(ws_.defined(), "no step has run");
    return ws_.narrow(0, (int64_t)off_render_, int64_t(12) * W_ * H_).view(at::kFloat).view({(int64_t)H_, (int64_t)W_, 3});
}
torch::Tensor GutTrainStep::alpha() const {
    TORCH_CHECK(ws_.defined(), "no step has run");
    return ws_.narrow(0, (int64_t)off_alpha_, int64_t(4) * W_ * H_).view(at::kFloat).view({(int64_t)H_, (int64_t)W_});
}
torch::Tensor GutTrainStep::radii() const {
    TORCH_CHECK(ws_.defined(), "no step has run");
    return ws_.narrow(0, (int64_t)off_radii_, int64_t(8) * N_).view(at::kInt).view({(int64_t)N_, 2});
}
} // namespace lfs

// ---------------------------------------------------------------------------------------------------------
// fast_gs::rasterization (rasterization_api.h:27-75, src/rasterization_api.cu) and fusedssim (ssim.cuh:11-30)
// ---------------------------------------------------------------------------------------------------------
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, int, int, int, int, int>
fast_gs::rasterization::forward_wrapper(
    const torch::Tensor& means, const torch::Tensor& scales_raw, const torch::Tensor& rotations_raw, const torch::Tensor& opacities_raw,
    const torch::Tensor& sh_coefficients_0, const torch::Tensor& sh_coefficients_rest, const torch::Tensor& w2c, const torch::Tensor& cam_position,
    const int active_sh_bases, const int width, const int height, const float focal_x, const float focal_y, const float center_x,
    const float center_y, const float near_plane, const float far_plane) {
    LFS_DEVICE_GUARD(means);
    LFS_CHECK_INPUT(means); LFS_CHECK_INPUT(scales_raw); LFS_CHECK_INPUT(rotations_raw); LFS_CHECK_INPUT(opacities_raw);
    LFS_CHECK_INPUT(sh_coefficients_0); LFS_CHECK_INPUT(sh_coefficients_rest);
    const uint32_t N = (uint32_t)means.size(0), total_rest = (uint32_t)sh_coefficients_rest.size(1);
    const at::Tensor w2c_c = w2c.contiguous(), cam_c = cam_position.contiguous();
    const auto fopt = means.options().dtype(at::kFloat), bopt = means.options().dtype(at::kByte);
    at::Tensor image = at::empty({3, height, width}, fopt), alpha = at::empty({1, height, width}, fopt);
    at::Tensor prim = at::empty({(int64_t)lfs_fastgs_primitive_workspace_bytes(N, (uint32_t)width, (uint32_t)height)}, bopt);
    at::Tensor n_inst_dev = at::zeros({1}, means.options().dtype(at::kLong));
    check_rc(lfs_fastgs_preprocess(N, means.data_ptr<float>(), scales_raw.data_ptr<float>(), rotations_raw.data_ptr<float>(), opacities_raw.data_ptr<float>(),
                                   sh_coefficients_0.data_ptr<float>(), total_rest ? sh_coefficients_rest.data_ptr<float>() : nullptr, total_rest,
                                   w2c_c.data_ptr<float>(), cam_c.data_ptr<float>(), (uint32_t)active_sh_bases, (uint32_t)width, (uint32_t)height,
                                   focal_x, focal_y, center_x, center_y, near_plane, far_plane, n_inst_dev.data_ptr<int64_t>(), prim.data_ptr(),
                                   (size_t)prim.numel(), cur_stream()), "fast_gs::rasterization::forward (preprocess)");
    const int64_t n_instances = n_inst_dev.item<int64_t>(); // the host sync of forward.cu:114-117
    at::Tensor inst = at::empty({(int64_t)std::max<size_t>(256, lfs_fastgs_instance_workspace_bytes((uint32_t)width, (uint32_t)height, n_instances))}, bopt);
    check_rc(lfs_fastgs_render(N, (uint32_t)width, (uint32_t)height, n_instances, prim.data_ptr(), (size_t)prim.numel(), inst.data_ptr(), (size_t)inst.numel(),
                               image.data_ptr<float>(), alpha.data_ptr<float>(), cur_stream()), "fast_gs::rasterization::forward (render)");
    at::Tensor none = at::empty({0}, bopt);
    return {image, alpha, prim, none, inst, none.clone(), 0, (int)n_instances, 0, 0, 0};
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
fast_gs::rasterization::backward_wrapper(
    torch::Tensor& densification_info, const torch::Tensor& grad_image, const torch::Tensor& grad_alpha, const torch::Tensor& image,
    const torch::Tensor& alpha, const torch::Tensor& means, const torch::Tensor& scales_raw, const torch::Tensor& rotations_raw,
    const torch::Tensor& sh_coefficients_rest, const torch::Tensor& per_primitive_buffers, const torch::Tensor& per_tile_buffers,
    const torch::Tensor& per_instance_buffers, const torch::Tensor& per_bucket_buffers, const torch::Tensor& w2c, const torch::Tensor& cam_position,
    const int active_sh_bases, const int width, const int height, const float focal_x, const float focal_y, const float center_x,
    const float center_y, const float near_plane, const float far_plane, const int n_visible_primitives, const int n_instances,
    const int n_buckets, const int primitive_primitive_indices_selector, const int instance_primitive_indices_selector) {
    (void)image; (void)per_tile_buffers; (void)per_bucket_buffers; (void)n_visible_primitives; (void)n_buckets;
    (void)primitive_primitive_indices_selector; (void)instance_primitive_indices_selector;
    LFS_DEVICE_GUARD(means);
    TORCH_CHECK(!w2c.requires_grad(), "pose optimisation (grad_w2c) is not implemented by this backend");
    const uint32_t N = (uint32_t)means.size(0), total_rest = (uint32_t)sh_coefficients_rest.size(1);
    const auto fopt = means.options().dtype(at::kFloat);
    at::Tensor g_means = at::empty({N, 3}, fopt), g_scales = at::empty({N, 3}, fopt), g_rot = at::empty({N, 4}, fopt), g_opac = at::empty({N, 1}, fopt);
    at::Tensor g_sh0 = at::empty({N, 1, 3}, fopt), g_shr = at::empty({N, (int64_t)total_rest, 3}, fopt);
    const at::Tensor gi = grad_image.contiguous(), ga = grad_alpha.contiguous(), al = alpha.contiguous(), w2c_c = w2c.contiguous(), cam_c = cam_position.contiguous();
    const bool dens = densification_info.defined() && densification_info.dim() > 0 && densification_info.size(0) > 0;
    check_rc(lfs_fastgs_backward(N, means.data_ptr<float>(), scales_raw.data_ptr<float>(), rotations_raw.data_ptr<float>(), nullptr,
                                 total_rest ? sh_coefficients_rest.data_ptr<float>() : nullptr, total_rest, w2c_c.data_ptr<float>(), cam_c.data_ptr<float>(),
                                 (uint32_t)active_sh_bases, (uint32_t)width, (uint32_t)height, focal_x, focal_y, center_x, center_y, near_plane, far_plane,
                                 (int64_t)n_instances, per_primitive_buffers.data_ptr(), (size_t)per_primitive_buffers.numel(), per_instance_buffers.data_ptr(),
                                 (size_t)per_instance_buffers.numel(), gi.data_ptr<float>(), ga.data_ptr<float>(), al.data_ptr<float>(),
                                 dens ? densification_info.data_ptr<float>() : nullptr, g_means.data_ptr<float>(), g_scales.data_ptr<float>(),
                                 g_rot.data_ptr<float>(), g_opac.data_ptr<float>(), g_sh0.data_ptr<float>(), total_rest ? g_shr.data_ptr<float>() : nullptr,
                                 cur_stream()), "fast_gs::rasterization::backward");
    return {g_means, g_scales, g_rot, g_opac, g_sh0, g_shr, torch::Tensor()};
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> fusedssim(float C1, float C2, torch::Tensor& img1, torch::Tensor& img2, bool train) {
    LFS_DEVICE_GUARD(img1);
    const at::Tensor a = img1.contiguous(), b = img2.contiguous();
    TORCH_CHECK(a.dim() == 4 && a.sizes() == b.sizes(), "fusedssim expects two [B,CH,H,W] tensors of the same shape");
    at::Tensor map = at::empty_like(a);
    at::Tensor d1 = train ? at::empty_like(a) : at::empty({0}, a.options()), d2 = train ? at::empty_like(a) : at::empty({0}, a.options()),
               d3 = train ? at::empty_like(a) : at::empty({0}, a.options());
    check_rc(lfs_fused_ssim_fwd((uint32_t)a.size(0), (uint32_t)a.size(1), (uint32_t)a.size(2), (uint32_t)a.size(3), C1, C2, a.data_ptr<float>(), b.data_ptr<float>(),
                                map.data_ptr<float>(), train ? d1.data_ptr<float>() : nullptr, train ? d2.data_ptr<float>() : nullptr,
                                train ? d3.data_ptr<float>() : nullptr, cur_stream()), "fusedssim");
    return {map, d1, d2, d3};
}

torch::Tensor fusedssim_backward(float C1, float C2, torch::Tensor& img1, torch::Tensor& img2, torch::Tensor& dL_dmap, torch::Tensor& dm_dmu1,
                                 torch::Tensor& dm_dsigma1_sq, torch::Tensor& dm_dsigma12) {
    LFS_DEVICE_GUARD(img1);
    const at::Tensor a = img1.contiguous(), b = img2.contiguous(), g = dL_dmap.contiguous(), d1 = dm_dmu1.contiguous(), d2 = dm_dsigma1_sq.contiguous(),
                     d3 = dm_dsigma12.contiguous();
    at::Tensor out = at::empty_like(a);
    check_rc(lfs_fused_ssim_bwd((uint32_t)a.size(0), (uint32_t)a.size(1), (uint32_t)a.size(2), (uint32_t)a.size(3), C1, C2, a.data_ptr<float>(), b.data_ptr<float>(),
                                g.data_ptr<float>(), d1.data_ptr<float>(), d2.data_ptr<float>(), d3.data_ptr<float>(), out.data_ptr<float>(), cur_stream()),
             "fusedssim_backward");
    return out;
}

// ---------------------------------------------------------------------------------------------------------
// gs::bilateral_grid (include/kernels/bilateral_grid.cuh:12-33)
// ---------------------------------------------------------------------------------------------------------
void gs::bilateral_grid::slice_forward_cuda(const torch::Tensor& grid, const torch::Tensor& rgb, torch::Tensor& output, bool use_uniform_coords) {
    (void)use_uniform_coords; // the reference ignores it as well: only uniform coordinates exist (bilateral_grid_forward.cu:96-115)
    LFS_DEVICE_GUARD(grid);
    LFS_CHECK_INPUT(grid); LFS_CHECK_INPUT(rgb); LFS_CHECK_INPUT(output);
    TORCH_CHECK(grid.dim() == 4 && grid.size(0) == 12, "Grid must be [12, L, H, W]");
    TORCH_CHECK(rgb.dim() == 3 && rgb.size(2) == 3 && output.sizes() == rgb.sizes(), "RGB / output must be [H, W, 3]");
    check_rc(lfs_bilateral_slice_fwd((uint32_t)grid.size(1), (uint32_t)grid.size(2), (uint32_t)grid.size(3), (uint32_t)rgb.size(0), (uint32_t)rgb.size(1),
                                     grid.data_ptr<float>(), rgb.data_ptr<float>(), 0, 0, output.data_ptr<float>(), cur_stream()), "bilateral_grid::slice_forward_cuda");
}

std::tuple<torch::Tensor, torch::Tensor> gs::bilateral_grid::slice_backward_cuda(const torch::Tensor& grid, const torch::Tensor& rgb, const torch::Tensor& grad_output) {
    LFS_DEVICE_GUARD(grid);
    LFS_CHECK_INPUT(grid); LFS_CHECK_INPUT(rgb); LFS_CHECK_INPUT(grad_output);
    TORCH_CHECK(grid.dim() == 4 && grid.size(0) == 12, "Grid must be [12, L, H, W]");
    TORCH_CHECK(rgb.dim() == 3 && rgb.size(2) == 3 && grad_output.sizes() == rgb.sizes(), "RGB / grad_output must be [H, W, 3]");
    at::Tensor grad_grid = at::zeros_like(grid), grad_rgb = at::empty_like(rgb);
    check_rc(lfs_bilateral_slice_bwd((uint32_t)grid.size(1), (uint32_t)grid.size(2), (uint32_t)grid.size(3), (uint32_t)rgb.size(0), (uint32_t)rgb.size(1),
                                     grid.data_ptr<float>(), rgb.data_ptr<float>(), grad_output.data_ptr<float>(), 0, 0, grad_grid.data_ptr<float>(),
                                     grad_rgb.data_ptr<float>(), cur_stream()), "bilateral_grid::slice_backward_cuda");
    return {grad_grid, grad_rgb};
}

torch::Tensor gs::bilateral_grid::tv_loss_forward_cuda(const torch::Tensor& grids) {
    LFS_DEVICE_GUARD(grids);
    LFS_CHECK_INPUT(grids);
    TORCH_CHECK(grids.dim() == 5 && grids.size(1) == 12, "Grids must be [N, 12, L, H, W]");
    at::Tensor tv = at::zeros({}, grids.options());
    check_rc(lfs_bilateral_tv_loss_fwd((uint32_t)grids.size(0), (uint32_t)grids.size(2), (uint32_t)grids.size(3), (uint32_t)grids.size(4), grids.data_ptr<float>(), 1.f,
                                       tv.data_ptr<float>(), cur_stream()), "bilateral_grid::tv_loss_forward_cuda");
    return tv;
}

torch::Tensor gs::bilateral_grid::tv_loss_backward_cuda(const torch::Tensor& grids, const torch::Tensor& grad_output) {
    LFS_DEVICE_GUARD(grids);
    LFS_CHECK_INPUT(grids);
    TORCH_CHECK(grids.dim() == 5 && grids.size(1) == 12, "Grids must be [N, 12, L, H, W]");
    at::Tensor grad = at::empty_like(grids);
    check_rc(lfs_bilateral_tv_loss_bwd((uint32_t)grids.size(0), (uint32_t)grids.size(2), (uint32_t)grids.size(3), (uint32_t)grids.size(4), grids.data_ptr<float>(),
                                       grad_output.item<float>(), 0, grad.data_ptr<float>(), cur_stream()), "bilateral_grid::tv_loss_backward_cuda");
    return grad;
}

// ---------------------------------------------------------------------------------------------------------
// gs::loader (src/loader/formats/colmap.cpp:907-957, transforms.cpp:73-265) over liblfs_io.so
// ---------------------------------------------------------------------------------------------------------
#include "../../include/lfs_io.h"
namespace {
std::tuple<std::vector<gs::loader::CameraData>, torch::Tensor> views_to_torch(lfs_colmap_scene* sc) {
    std::vector<gs::loader::CameraData> out;
    const auto f32 = torch::TensorOptions().dtype(torch::kFloat32);
    const uint64_t n = lfs_colmap_num_views(sc);
    out.reserve(n);
    for (uint64_t i = 0; i < n; ++i) {
        lfs_colmap_view v;
        lfs_colmap_view_at(sc, i, &v);
        gs::loader::CameraData c;
        c._camera_ID = v.camera_id; c._camera_model = v.colmap_model; c._camera_model_type = v.camera_model_type;
        c._width = v.width; c._height = v.height;
        c._focal_x = v.focal_x; c._focal_y = v.focal_y; c._center_x = v.center_x; c._center_y = v.center_y;
        c._R = torch::from_blob(v.R, {3, 3}, f32).clone();
        c._T = torch::from_blob(v.T, {3}, f32).clone();
        c._radial_distortion = torch::from_blob(v.radial, {v.n_radial}, f32).clone();
        c._tangential_distortion = torch::from_blob(v.tangential, {v.n_tangential}, f32).clone();
        c._params = torch::from_blob(v.params, {v.n_params}, f32).clone();
        c._image_name = lfs_colmap_image_name(sc, i);
        c._image_path = lfs_colmap_image_path(sc, i);
        out.push_back(std::move(c));
    }
    float center[3];
    lfs_colmap_scene_center(sc, center);
    torch::Tensor ctr = torch::from_blob(center, {3}, f32).clone();
    lfs_colmap_close(sc);
    return {std::move(out), ctr};
}
std::tuple<std::vector<gs::loader::CameraData>, torch::Tensor> open_colmap(const std::filesystem::path& base, const std::string& folder, int format) {
    lfs_colmap_scene* sc = nullptr;
    if (lfs_colmap_open(base.string().c_str(), folder.c_str(), format, &sc) != LFS_IO_OK) throw std::runtime_error(lfs_io_last_error());
    return views_to_torch(sc);
}
gs::loader::PointCloud open_points(const std::filesystem::path& base, int format) {
    lfs_point_cloud* pc = nullptr;
    if (lfs_colmap_points_open(base.string().c_str(), format, &pc) != LFS_IO_OK) throw std::runtime_error(lfs_io_last_error());
    const int64_t n = (int64_t)lfs_point_cloud_size(pc);
    gs::loader::PointCloud out{torch::empty({n, 3}, torch::kFloat32), torch::empty({n, 3}, torch::kUInt8)};
    lfs_point_cloud_copy(pc, out.means.data_ptr<float>(), out.colors.data_ptr<uint8_t>());
    lfs_point_cloud_close(pc);
    return out;
}
} // namespace
std::tuple<std::vector<gs::loader::CameraData>, torch::Tensor> gs::loader::read_colmap_cameras_and_images(const std::filesystem::path& base, const std::string& f) { return open_colmap(base, f, 0); }
std::tuple<std::vector<gs::loader::CameraData>, torch::Tensor> gs::loader::read_colmap_cameras_and_images_text(const std::filesystem::path& base, const std::string& f) { return open_colmap(base, f, 1); }
std::tuple<std::vector<gs::loader::CameraData>, torch::Tensor> gs::loader::read_transforms_cameras_and_images(const std::filesystem::path& trans_path) {
    lfs_colmap_scene* sc = nullptr;
    if (lfs_transforms_open(trans_path.string().c_str(), &sc) != LFS_IO_OK) throw std::runtime_error(lfs_io_last_error());
    return views_to_torch(sc);
}
gs::loader::PointCloud gs::loader::read_colmap_point_cloud(const std::filesystem::path& base) { return open_points(base, 0); }
gs::loader::PointCloud gs::loader::read_colmap_point_cloud_text(const std::filesystem::path& base) { return open_points(base, 1); }

void gs::loader::save_ply(const std::filesystem::path& path, const torch::Tensor& means, const torch::Tensor& sh0, const torch::Tensor& shN, const torch::Tensor& scaling,
                          const torch::Tensor& rotation, const torch::Tensor& opacity) {
    auto host = [](const torch::Tensor& t) { return t.detach().to(torch::kCPU, torch::kFloat32).contiguous(); };
    const int64_t N = means.size(0);
    TORCH_CHECK(means.dim() == 2 && means.size(1) == 3 && sh0.size(0) == N && shN.size(0) == N && scaling.size(0) == N && rotation.size(0) == N && opacity.size(0) == N,
                "save_ply: inconsistent shapes");
    const torch::Tensor m = host(means), dc = host(sh0.transpose(1, 2).reshape({N, -1})), rest = host(shN.transpose(1, 2).reshape({N, -1})), sc = host(scaling), ro = host(torch::nn::functional::normalize(rotation, torch::nn::functional::NormalizeFuncOptions().dim(-1))),
                        op = host(opacity.reshape({N}));
    if (lfs_ply_write_splat(path.string().c_str(), (uint64_t)N, (uint32_t)dc.size(1), (uint32_t)rest.size(1), m.data_ptr<float>(), nullptr, dc.data_ptr<float>(),
                            rest.size(1) ? rest.data_ptr<float>() : nullptr, op.data_ptr<float>(), sc.data_ptr<float>(), ro.data_ptr<float>()) != LFS_IO_OK)
        throw std::runtime_error(lfs_io_last_error());
}

gs::loader::SplatTensors gs::loader::load_ply(const std::filesystem::path& path) {
    lfs_ply* ply = nullptr;
    if (lfs_ply_open(path.string().c_str(), &ply) != LFS_IO_OK) throw std::runtime_error(lfs_io_last_error());
    const int64_t N = (int64_t)lfs_ply_num_vertices(ply), P = (int64_t)lfs_ply_num_properties(ply);
    std::vector<std::string> names;
    for (int64_t i = 0; i < P; ++i) names.emplace_back(lfs_ply_property_name(ply, (uint32_t)i));
    torch::Tensor all = torch::empty({N, P}, torch::kFloat32);
    const int rc = lfs_ply_read(ply, all.data_ptr<float>());
    lfs_ply_close(ply);
    if (rc != LFS_IO_OK) throw std::runtime_error(lfs_io_last_error());
    auto columns = [&](const std::string& prefix, bool exact) {      // property columns "prefix" or "prefix<k>", k ascending
        std::vector<std::pair<int, int64_t>> cols;
        for (int64_t i = 0; i < P; ++i) {
            if (exact) { if (names[i] == prefix) cols.push_back({0, i}); }
            else if (names[i].rfind(prefix, 0) == 0) cols.push_back({std::stoi(names[i].substr(prefix.size())), i});
        }
        std::sort(cols.begin(), cols.end());
        std::vector<int64_t> idx;
        for (auto& c : cols) idx.push_back(c.second);
        return all.index_select(1, torch::tensor(idx, torch::kLong));
    };
    gs::loader::SplatTensors out;
    out.means = torch::cat({columns("x", true), columns("y", true), columns("z", true)}, 1);
    if (out.means.size(1) != 3) throw std::runtime_error("Only binary PLY with position supported");
    // the reference's defaults for missing columns (ply.cpp:531-600; held to its reader in tests/test_loader_reference.py): sh0 zeros [N,1,3], shN zeros
    // [N,15,3], opacity 0, log-scale -5, the identity quaternion
    const torch::Tensor dc = columns("f_dc_", false), rest = columns("f_rest_", false), sc = columns("scale_", false), ro = columns("rot_", false), op = columns("opacity", true);
    out.sh0 = (dc.size(1) && dc.size(1) % 3 == 0) ? dc.reshape({N, 3, -1}).transpose(1, 2).contiguous() : torch::zeros({N, 1, 3}, torch::kFloat32);
    out.shN = (rest.size(1) && rest.size(1) % 3 == 0) ? rest.reshape({N, 3, -1}).transpose(1, 2).contiguous() : torch::zeros({N, 15, 3}, torch::kFloat32);
    out.scaling = sc.size(1) == 3 ? sc.contiguous() : torch::full({N, 3}, -5.0f, torch::kFloat32);
    if (ro.size(1) == 4) out.rotation = ro.contiguous();
    else {
        out.rotation = torch::zeros({N, 4}, torch::kFloat32);
        out.rotation.select(1, 0).fill_(1.0f);
    }
    out.opacity = op.size(1) == 1 ? op.reshape({N}).contiguous() : torch::zeros({N}, torch::kFloat32);
    return out;
}
