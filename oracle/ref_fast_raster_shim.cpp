// ORACLE/_ref - TEST INFRASTRUCTURE ONLY.
// The reference's DEFAULT training render path (fastgs / EWA), end to end, on the CPU (oracle/Makefile, `make reffast` -> oracle/_ref/libref_fast_raster.so):
//   src/training/rasterization/fast_rasterizer.cpp, fast_rasterizer_autograd.cpp    fast_rasterize(): settings from the camera, FastGSRasterize autograd Function,
//                                                                                   background compositing image + (1 - alpha) * bg        (whole files)
//   fastgs/rasterization/src/rasterization_api.cu                                   forward_wrapper / backward_wrapper: buffers through resize callbacks, zeroed
//                                                                                   gradients, optional densification_info (whole file; sed: torch::kCUDA -> kCPU)
//   fastgs/rasterization/src/forward.cu, backward.cu + the kernel headers           under oracle/ref_emul/ (ref_fastgs_all.cpp; the edits of `make refk_fastgs`)
//   src/core/camera.cpp, SplatData's constructors / getters                         as in `make refraster`
// Nothing of this path is restated. Used by oracle/make_golden_refk_fastgs.py ("composed/..." entries of tests/golden/refk_fastgs.npz), which
// tests/test_gpu_refk_golden.py holds the product's fastgs.fast_rasterize + backward to; the same library regenerates the file's wrapper-level cases through the
// reference's own forward_wrapper / backward_wrapper (reffast_wrappers below), which validates the restated wrapper of ref_kernels_fastgs.cpp.
#include "core/camera.hpp"
#include "core/image_io.hpp"
#include "core/splat_data.hpp"
#include "fast_rasterizer.hpp"
#include "kernels/fused_ssim.cuh" // the reference's fused-SSIM autograd wrapper over its own ssim.cu (compiled whole under the emulator): reffast_train_loss_backward
#include "rasterization_api.h"
#include <cstring>

#define REF_API extern "C" __attribute__((visibility("default")))

std::tuple<unsigned char*, int, int, int> load_image(std::filesystem::path, int, int) { throw std::runtime_error("ref_fast_raster_shim: no image loading"); }
void free_image(unsigned char*) {}
std::tuple<int, int, int> get_image_info(std::filesystem::path) { throw std::runtime_error("ref_fast_raster_shim: no image loading"); }

static torch::Tensor f32(const float* p, std::vector<int64_t> shape) { return torch::from_blob(const_cast<float*>(p), shape, torch::kFloat32).clone(); }
static void put(const torch::Tensor& t, float* dst) {
    if (!dst || !t.defined()) return;
    auto c = t.detach().to(torch::kFloat32).contiguous();
    std::memcpy(dst, c.data_ptr<float>(), sizeof(float) * c.numel());
}

// fast_rasterize(camera, model, bg) + backward of sum(image * v_image) + sum(alpha * v_alpha); densification_info [2,N] in/out (null: the empty tensor the MCMC
// strategy keeps). Raw parameters as SplatData holds them (opacity [N,1]).
REF_API int reffast_render_backward(int64_t N, int64_t K1, int sh_degree, int active_sh_degree, const float* means, const float* sh0, const float* shN,
                                    const float* scaling, const float* rotation, const float* opacity, const float* R, const float* T, float fx, float fy, float cx,
                                    float cy, int width, int height, const float* bg, const float* v_image, const float* v_alpha, float* densification_info,
                                    float* image, float* alpha, float* g_means, float* g_sh0, float* g_shN, float* g_scaling, float* g_rotation, float* g_opacity) {
    try {
        auto req = [](torch::Tensor t) { return t.set_requires_grad(true); };
        gs::SplatData model(sh_degree, req(f32(means, {N, 3})), req(f32(sh0, {N, 1, 3})), req(f32(shN, {N, K1, 3})), req(f32(scaling, {N, 3})), req(f32(rotation, {N, 4})),
                            req(f32(opacity, {N, 1})), 1.0f);
        model.set_active_sh_degree(active_sh_degree);
        if (densification_info) model._densification_info = f32(densification_info, {2, N});
        gs::Camera cam(f32(R, {3, 3}), f32(T, {3}), fx, fy, cx, cy, torch::empty({0}, torch::kFloat32), torch::empty({0}, torch::kFloat32), gsplat::CameraModelType::PINHOLE,
                       "view", "", width, height, 0);
        auto bgc = f32(bg, {3});
        auto out = gs::training::fast_rasterize(cam, model, bgc);
        put(out.image, image), put(out.alpha, alpha);
        auto loss = (out.image * f32(v_image, {3, height, width})).sum() + (out.alpha * f32(v_alpha, {1, height, width})).sum();
        loss.backward();
        auto g = [](const torch::Tensor& p) { return p.grad().defined() ? p.grad() : torch::zeros_like(p); };
        put(g(model.means()), g_means), put(g(model.sh0()), g_sh0), put(g(model.shN()), g_shN), put(g(model.scaling_raw()), g_scaling);
        put(g(model.rotation_raw()), g_rotation), put(g(model.opacity_raw()), g_opacity);
        if (densification_info) put(model._densification_info, densification_info);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "reffast_render_backward: %s\n", e.what());
        return 1;
    }
}

// The wrapper level (the shape of refk_fastgs_fwd_bwd in ref_kernels_fastgs.cpp) through the reference's own forward_wrapper / backward_wrapper
REF_API int reffast_wrappers(int n, int active_sh_bases, int total_bases_sh_rest, int width, int height, float fx, float fy, float cx, float cy, float near_plane,
                             float far_plane, const float* means, const float* scales_raw, const float* rotations_raw, const float* opacities_raw, const float* sh0,
                             const float* sh_rest, const float* w2c, const float* cam_position, const float* grad_image, const float* grad_alpha, float* image, float* alpha,
                             float* grad_means, float* grad_scales_raw, float* grad_rotations_raw, float* grad_opacities_raw, float* grad_sh0, float* grad_sh_rest,
                             float* densification_info, int32_t* counts) {
    try {
        using namespace fast_gs::rasterization;
        auto m = f32(means, {n, 3}), s = f32(scales_raw, {n, 3}), r = f32(rotations_raw, {n, 4}), o = f32(opacities_raw, {n, 1}), c0 = f32(sh0, {n, 1, 3}),
             cr = f32(sh_rest, {n, total_bases_sh_rest, 3}), w = f32(w2c, {1, 4, 4}), cp = f32(cam_position, {3});
        auto fw = forward_wrapper(m, s, r, o, c0, cr, w, cp, active_sh_bases, width, height, fx, fy, cx, cy, near_plane, far_plane);
        put(std::get<0>(fw), image), put(std::get<1>(fw), alpha);
        counts[0] = std::get<6>(fw), counts[1] = std::get<7>(fw), counts[2] = std::get<8>(fw);
        if (grad_image) {
            torch::Tensor dens = densification_info ? f32(densification_info, {2, n}) : torch::empty({0});
            auto gi = f32(grad_image, {3, height, width}), ga = f32(grad_alpha, {1, height, width});
            auto bw = backward_wrapper(dens, gi, ga, std::get<0>(fw), std::get<1>(fw), m, s, r, cr, std::get<2>(fw), std::get<3>(fw), std::get<4>(fw), std::get<5>(fw), w, cp,
                                       active_sh_bases, width, height, fx, fy, cx, cy, near_plane, far_plane, std::get<6>(fw), std::get<7>(fw), std::get<8>(fw),
                                       std::get<9>(fw), std::get<10>(fw));
            put(std::get<0>(bw), grad_means), put(std::get<1>(bw), grad_scales_raw), put(std::get<2>(bw), grad_rotations_raw), put(std::get<3>(bw), grad_opacities_raw);
            put(std::get<4>(bw), grad_sh0), put(std::get<5>(bw), grad_sh_rest);
            if (densification_info) put(dens, densification_info);
        }
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "reffast_wrappers: %s\n", e.what());
        return 1;
    }
}

// The photometric loss of one training step on the default path as Trainer::train_step composes it (trainer.cpp:656, 668-677): fast_rasterize() (the image is NOT
// clamped on this path) -> compute_photometric_loss (:103-130, restated line by line: l1_loss, 1 - fused_ssim(..., "valid", train), the lambda mix) -> backward().
REF_API int reffast_train_loss_backward(int64_t N, int64_t K1, int sh_degree, int active_sh_degree, const float* means, const float* sh0, const float* shN,
                                        const float* scaling, const float* rotation, const float* opacity, const float* R, const float* T, float fx, float fy, float cx,
                                        float cy, int width, int height, const float* bg, const float* gt_image, float lambda_dssim, float* densification_info,
                                        float* loss_out, float* g_means, float* g_sh0, float* g_shN, float* g_scaling, float* g_rotation, float* g_opacity) {
    try {
        auto req = [](torch::Tensor t) { return t.set_requires_grad(true); };
        gs::SplatData model(sh_degree, req(f32(means, {N, 3})), req(f32(sh0, {N, 1, 3})), req(f32(shN, {N, K1, 3})), req(f32(scaling, {N, 3})), req(f32(rotation, {N, 4})),
                            req(f32(opacity, {N, 1})), 1.0f);
        model.set_active_sh_degree(active_sh_degree);
        if (densification_info) model._densification_info = f32(densification_info, {2, N});
        gs::Camera cam(f32(R, {3, 3}), f32(T, {3}), fx, fy, cx, cy, torch::empty({0}, torch::kFloat32), torch::empty({0}, torch::kFloat32), gsplat::CameraModelType::PINHOLE,
                       "view", "", width, height, 0);
        auto bgc = f32(bg, {3});
        auto out = gs::training::fast_rasterize(cam, model, bgc);
        torch::Tensor rendered = out.image.unsqueeze(0), gt = f32(gt_image, {3, height, width}).unsqueeze(0);       // :113-114
        auto l1_loss = torch::l1_loss(rendered, gt);                                                                 // :121
        auto ssim_loss = 1.f - fused_ssim(rendered, gt, "valid", /*train=*/true);                                    // :122
        torch::Tensor loss = (1.f - lambda_dssim) * l1_loss + lambda_dssim * ssim_loss;                              // :123-124
        loss.backward();
        *loss_out = loss.item<float>();
        auto g = [](const torch::Tensor& p) { return p.grad().defined() ? p.grad() : torch::zeros_like(p); };
        put(g(model.means()), g_means), put(g(model.sh0()), g_sh0), put(g(model.shN()), g_shN), put(g(model.scaling_raw()), g_scaling);
        put(g(model.rotation_raw()), g_rotation), put(g(model.opacity_raw()), g_opacity);
        if (densification_info) put(model._densification_info, densification_info);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "reffast_train_loss_backward: %s\n", e.what());
        return 1;
    }
}
