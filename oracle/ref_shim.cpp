// ORACLE/_ref — TEST INFRASTRUCTURE ONLY.
// C entry points over the reference's OWN CPU code: /root/reference/tests/torch_impl.cpp
// (namespace reference::), compiled in place by oracle/Makefile (`make ref`). Used to
// (1) validate oracle_ops.hpp where the two overlap (quat->rotmat,
// spherical harmonics, tile intersection) and (2) time the reference's CPU path as
// bench.py's cpu_baseline ("kind": "reference"). Nothing here is product code.
#include "torch_impl.hpp" // resolved from $(REF)/tests by the Makefile include path
#include <cstring>

#define REF_API extern "C" __attribute__((visibility("default")))

static torch::Tensor f32(const float* p, std::vector<int64_t> shape) {
    return torch::from_blob(const_cast<float*>(p), shape, torch::kFloat32).clone();
}

REF_API void ref_quat_to_rotmat(int64_t N, const float* quats, float* rotmats) {
    auto R = reference::quat_to_rotmat(f32(quats, {N, 4})).contiguous();
    std::memcpy(rotmats, R.data_ptr<float>(), sizeof(float) * 9 * N);
}

// dirs [N,3], coeffs [N,K,3] -> colors [N,3]
REF_API void ref_spherical_harmonics(int64_t N, int K, int degree, const float* dirs, const float* coeffs, float* colors) {
    auto c = reference::spherical_harmonics(degree, f32(dirs, {N, 3}), f32(coeffs, {N, K, 3})).contiguous();
    std::memcpy(colors, c.data_ptr<float>(), sizeof(float) * 3 * N);
}

// means2d [C,N,2], radii int32 [C,N,2], depths [C,N]. Returns n_isects; the
// caller passes buffers of capacity `cap` (call with cap=0 first to size).
REF_API int64_t ref_isect_tiles(int64_t C, int64_t N, const float* means2d, const int32_t* radii, const float* depths,
                                int tile_size, int tile_width, int tile_height, int sort,
                                int32_t* tiles_per_gauss, int64_t* isect_ids, int32_t* flatten_ids, int64_t cap) {
    auto r = torch::from_blob(const_cast<int32_t*>(radii), {C, N, 2}, torch::kInt32).clone();
    auto [tpg, ids, flat] = reference::isect_tiles(f32(means2d, {C, N, 2}), r, f32(depths, {C, N}),
                                                   tile_size, tile_width, tile_height, sort != 0);
    tpg = tpg.to(torch::kInt32).contiguous(); ids = ids.to(torch::kInt64).contiguous(); flat = flat.to(torch::kInt32).contiguous();
    const int64_t n = ids.numel();
    if (tiles_per_gauss) std::memcpy(tiles_per_gauss, tpg.data_ptr<int32_t>(), sizeof(int32_t) * C * N);
    if (n <= cap) {
        std::memcpy(isect_ids, ids.data_ptr<int64_t>(), sizeof(int64_t) * n);
        std::memcpy(flatten_ids, flat.data_ptr<int32_t>(), sizeof(int32_t) * n);
    }
    return n;
}

// EWA pinhole projection of the reference (tests/torch_impl.cpp:146-217). NOT the UT
// kernel's semantics (fixed 3.33 sigma radii, det clamp) — used only as an indicative
// cross-check of means2d / depths / conics for small Gaussians.
REF_API void ref_fully_fused_projection(int64_t N, const float* means, const float* quats, const float* scales,
                                        const float* viewmat, const float* Kmat, int width, int height, float eps2d,
                                        float near_plane, float far_plane,
                                        int32_t* radii, float* means2d, float* depths, float* conics) {
    auto [covars, precis] = reference::quat_scale_to_covar_preci(f32(quats, {N, 4}), f32(scales, {N, 3}), true, false, false);
    auto [r, m2, d, c, comp] = reference::fully_fused_projection(f32(means, {N, 3}), covars, f32(viewmat, {1, 4, 4}), f32(Kmat, {1, 3, 3}),
                                                                 width, height, eps2d, near_plane, far_plane, false, "pinhole");
    r = r.to(torch::kInt32).contiguous(); m2 = m2.contiguous(); d = d.contiguous(); c = c.contiguous();
    std::memcpy(radii, r.data_ptr<int32_t>(), sizeof(int32_t) * 2 * N);
    std::memcpy(means2d, m2.data_ptr<float>(), sizeof(float) * 2 * N);
    std::memcpy(depths, d.data_ptr<float>(), sizeof(float) * N);
    std::memcpy(conics, c.data_ptr<float>(), sizeof(float) * 3 * N);
}

// Timed CPU baseline of the reference's projection-side stages on its own
// config-1 shape: covar + EWA projection + SH + isect, fwd only.
REF_API int64_t ref_cpu_stage_pipeline(int64_t N, int K, int degree, const float* means, const float* quats, const float* scales,
                                       const float* coeffs, const float* viewmat, const float* Kmat, int width, int height,
                                       int tile_size, int with_isect) {
    auto m = f32(means, {N, 3}), q = f32(quats, {N, 4}), s = f32(scales, {N, 3});
    auto [covars, precis] = reference::quat_scale_to_covar_preci(q, s, true, false, false);
    auto vm = f32(viewmat, {1, 4, 4}), Kt = f32(Kmat, {1, 3, 3});
    auto [radii, means2d, depths, conics, comp] = reference::fully_fused_projection(m, covars, vm, Kt, width, height);
    auto campos = torch::inverse(vm).index({torch::indexing::Slice(), torch::indexing::Slice(0, 3), 3});
    auto dirs = m - campos;
    auto colors = reference::spherical_harmonics(degree, dirs, f32(coeffs, {N, K, 3}));
    int64_t n = 0;
    if (with_isect) {
        int tw = (width + tile_size - 1) / tile_size, th = (height + tile_size - 1) / tile_size;
        auto [tpg, ids, flat] = reference::isect_tiles(means2d, radii.to(torch::kInt32), depths, tile_size, tw, th, true);
        n = ids.numel();
    }
    return n + (colors.numel() > 0 ? 0 : 0);
}

// SURVEY.md 8(d) "CPU baseline timing": quat_scale_to_covar_preci + fully_fused_projection + spherical_harmonics (+ isect_tiles) of the reference's
// tests/torch_impl.cpp (:38, :147, :296, :324), FORWARD + AUTOGRAD BACKWARD (the gradient of a fixed linear functional of means2d, depths, conics and colours
// with respect to means, quats, scales and the SH coefficients), `repeats` times on `threads` libtorch intra-op threads; seconds[r] = wall time of repeat r.
// Returns the number of intersections (0 without isect). The per-element `.item()` loop of isect_tiles (torch_impl.cpp:370-397) makes it impractical at 1 M.
#include <chrono>
REF_API int64_t ref_cpu_stage_fwd_bwd(int64_t N, int K, int degree, const float* means, const float* quats, const float* scales, const float* coeffs,
                                      const float* viewmat, const float* Kmat, int width, int height, int tile_size, int with_isect, int threads, int repeats,
                                      double* seconds) {
    at::set_num_threads(threads);
    auto req = [](torch::Tensor t) { return t.set_requires_grad(true); };
    auto m = req(f32(means, {N, 3})), q = req(f32(quats, {N, 4})), s = req(f32(scales, {N, 3})), c = req(f32(coeffs, {N, K, 3}));
    auto vm = f32(viewmat, {1, 4, 4}), Kt = f32(Kmat, {1, 3, 3});
    torch::manual_seed(1);
    auto w2 = torch::randn({1, N, 2}), wd = torch::randn({1, N}), wc = torch::randn({1, N, 3}), wcol = torch::randn({N, 3});
    int64_t n = 0;
    for (int r = 0; r < repeats; ++r) {
        for (auto* t : {&m, &q, &s, &c}) if (t->grad().defined()) t->mutable_grad().reset();
        const auto t0 = std::chrono::steady_clock::now();
        auto [covars, precis] = reference::quat_scale_to_covar_preci(q, s, true, false, false);
        auto [radii, means2d, depths, conics, comp] = reference::fully_fused_projection(m, covars, vm, Kt, width, height);
        auto campos = torch::inverse(vm).index({torch::indexing::Slice(), torch::indexing::Slice(0, 3), 3});
        auto colors = reference::spherical_harmonics(degree, m - campos, c);
        if (with_isect) {
            torch::NoGradGuard ng;
            int tw = (width + tile_size - 1) / tile_size, th = (height + tile_size - 1) / tile_size;
            auto [tpg, ids, flat] = reference::isect_tiles(means2d.detach(), radii.to(torch::kInt32), depths.detach(), tile_size, tw, th, true);
            n = ids.numel();
        }
        auto loss = (means2d * w2).sum() + (depths * wd).sum() + (conics * wc).sum() + (colors * wcol).sum();
        loss.backward();
        seconds[r] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return n;
}
