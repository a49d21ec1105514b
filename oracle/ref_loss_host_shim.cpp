// ORACLE/_ref - TEST INFRASTRUCTURE ONLY.
// The reference's OWN loss-side host code on the CPU (oracle/Makefile, `make reflosshost`):
//   include/kernels/fused_ssim.cuh                    the header-only autograd wrapper of the fused SSIM ("valid" crop, mean)   (real header)
//   src/training/components/bilateral_grid.cpp        BilateralGrid (identity initialisation, apply(): clamp / permute / slice, tv_loss) and its two autograd Functions
//                                                     (whole file; sed: torch::kCUDA -> torch::kCPU and the two-line TORCH_CHECK(... is_cuda() ...) dropped)
//   src/training/optimizers/scheduler.cpp, fused_adam.cpp   WarmupExponentialLR (the bilateral grid's schedule) over the optimizer base class
// compiled in place against CPU libtorch. The host functions the wrappers call - fusedssim / fusedssim_backward (ssim.cu:426-520) and the four
// gs::bilateral_grid::*_cuda launchers (bilateral_grid_forward.cu:96-115, _backward.cu:155-183, _tv.cu:137-188) - are the reference's own too: the four .cu files
// are compiled whole under the emulator (ref_loss_all.cpp, through ref_cu_prep.py). Trainer::compute_photometric_loss (trainer.cpp:103-130) is a member of the Trainer class;
// its three lines are restated in reflh_photometric, cited. Used by oracle/make_golden_refk_loss.py -> tests/golden/refk_loss.npz ("host/..." entries).
#include "kernels/bilateral_grid.cuh"
#include "kernels/fused_ssim.cuh"
#include "bilateral_grid.hpp"
#include "adam_api.h"
#include "optimizers/fused_adam.hpp"
#include "optimizers/scheduler.hpp"
#include <cstring>

#define REF_API extern "C" __attribute__((visibility("default")))

#ifndef REF_REAL_LOSS // (default build: -DREF_REAL_LOSS - these host functions come from the reference's own .cu files, ref_loss_all.cpp; the restated form below
// is kept for reference and builds against libref_loss.so without the define)
extern "C" {
void refk_fusedssim(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2, int train, float* ssim_map, float* dm_dmu1,
                    float* dm_dsigma1_sq, float* dm_dsigma12);
void refk_fusedssim_backward(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2, const float* dL_dmap, const float* dm_dmu1,
                             const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg1);
void refk_bilateral_slice_forward(const float* grid, const float* rgb, float* out, int L, int H, int W, int h, int w);
void refk_bilateral_slice_backward(const float* grid, const float* rgb, const float* grad_output, float* grad_grid, float* grad_rgb, int L, int H, int W, int h, int w);
void refk_bilateral_tv_forward(const float* grids, float* loss, int N, int L, int H, int W);
void refk_bilateral_tv_backward(const float* grids, float grad_output, float* grad_grids, int N, int L, int H, int W);
}

// ssim.cu:426-474: the map and, when training, the three derivative maps (zeros_like), otherwise empty tensors
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> fusedssim(float C1, float C2, torch::Tensor& img1, torch::Tensor& img2, bool train) {
    auto a = img1.contiguous(), b = img2.contiguous();
    const int B = a.size(0), CH = a.size(1), H = a.size(2), W = a.size(3);
    auto map = torch::zeros_like(a);
    auto d1 = train ? torch::zeros_like(a) : torch::empty({0}, a.options()), d2 = train ? torch::zeros_like(a) : torch::empty({0}, a.options()),
         d3 = train ? torch::zeros_like(a) : torch::empty({0}, a.options());
    refk_fusedssim(B, CH, H, W, C1, C2, a.data_ptr<float>(), b.data_ptr<float>(), train ? 1 : 0, map.data_ptr<float>(), train ? d1.data_ptr<float>() : nullptr,
                   train ? d2.data_ptr<float>() : nullptr, train ? d3.data_ptr<float>() : nullptr);
    return {map, d1, d2, d3};
}
// ssim.cu:476-520
torch::Tensor fusedssim_backward(float C1, float C2, torch::Tensor& img1, torch::Tensor& img2, torch::Tensor& dL_dmap, torch::Tensor& dm_dmu1,
                                 torch::Tensor& dm_dsigma1_sq, torch::Tensor& dm_dsigma12) {
    auto a = img1.contiguous(), b = img2.contiguous(), g = dL_dmap.contiguous(), d1 = dm_dmu1.contiguous(), d2 = dm_dsigma1_sq.contiguous(), d3 = dm_dsigma12.contiguous();
    auto out = torch::zeros_like(a);
    refk_fusedssim_backward(a.size(0), a.size(1), a.size(2), a.size(3), C1, C2, a.data_ptr<float>(), b.data_ptr<float>(), g.data_ptr<float>(), d1.data_ptr<float>(),
                            d2.data_ptr<float>(), d3.data_ptr<float>(), out.data_ptr<float>());
    return out;
}

namespace gs::bilateral_grid {
    void slice_forward_cuda(const torch::Tensor& grid, const torch::Tensor& rgb, torch::Tensor& output, bool) { // bilateral_grid_forward.cu:96-115
        auto g = grid.contiguous(), r = rgb.contiguous();
        TORCH_CHECK(output.is_contiguous(), "output must be contiguous");
        refk_bilateral_slice_forward(g.data_ptr<float>(), r.data_ptr<float>(), output.data_ptr<float>(), g.size(1), g.size(2), g.size(3), r.size(0), r.size(1));
    }
    std::tuple<torch::Tensor, torch::Tensor> slice_backward_cuda(const torch::Tensor& grid, const torch::Tensor& rgb, const torch::Tensor& grad_output) { // _backward.cu:155-183
        auto g = grid.contiguous(), r = rgb.contiguous(), go = grad_output.contiguous();
        auto gg = torch::zeros_like(g), gr = torch::zeros_like(r);
        refk_bilateral_slice_backward(g.data_ptr<float>(), r.data_ptr<float>(), go.data_ptr<float>(), gg.data_ptr<float>(), gr.data_ptr<float>(), g.size(1), g.size(2),
                                      g.size(3), r.size(0), r.size(1));
        return {gg, gr};
    }
    torch::Tensor tv_loss_forward_cuda(const torch::Tensor& grids) { // _tv.cu:137-160: a zero-dim result
        auto g = grids.contiguous();
        auto loss = torch::zeros({}, g.options());
        refk_bilateral_tv_forward(g.data_ptr<float>(), loss.data_ptr<float>(), g.size(0), g.size(2), g.size(3), g.size(4));
        return loss;
    }
    torch::Tensor tv_loss_backward_cuda(const torch::Tensor& grids, const torch::Tensor& grad_output) { // _tv.cu:162-188: grad_output read on the host (:181)
        auto g = grids.contiguous();
        auto gg = torch::zeros_like(g);
        refk_bilateral_tv_backward(g.data_ptr<float>(), grad_output.item<float>(), gg.data_ptr<float>(), g.size(0), g.size(2), g.size(3), g.size(4));
        return gg;
    }
} // namespace gs::bilateral_grid

#endif

namespace fast_gs::optimizer { // fused_adam.cpp links against it; no Adam step is taken here
    void adam_step_wrapper(torch::Tensor&, torch::Tensor&, torch::Tensor&, const torch::Tensor&, const float, const float, const float, const float, const float,
                           const float) { TORCH_CHECK(false, "ref_loss_host_shim: no optimizer step"); }
}

static torch::Tensor f32(const float* p, std::vector<int64_t> shape) { return torch::from_blob(const_cast<float*>(p), shape, torch::kFloat32).clone(); }
static void put(const torch::Tensor& t, float* dst) {
    if (!dst) return;
    auto c = t.detach().to(torch::kFloat32).contiguous();
    std::memcpy(dst, c.data_ptr<float>(), sizeof(float) * c.numel());
}
template <class F> static int guarded(F&& f) {
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "ref_loss_host_shim: %s\n", e.what());
        return 1;
    }
}

// fused_ssim(img1, img2, padding, train) [3,H,W] -> value; d value / d img1
REF_API int reflh_fused_ssim(int H, int W, const float* img1, const float* img2, int valid, float* value, float* grad_img1) {
    return guarded([&] {
        auto a = f32(img1, {3, H, W}).set_requires_grad(true);
        auto v = fused_ssim(a, f32(img2, {3, H, W}), valid ? "valid" : "same", true);
        v.backward();
        *value = v.item<float>();
        put(a.grad().defined() ? a.grad() : torch::zeros_like(a), grad_img1);
    });
}
// Trainer::compute_photometric_loss (trainer.cpp:110-125): 4-D views, l1_loss, 1 - fused_ssim(..., "valid", train = true), the lambda mix
REF_API int reflh_photometric(int H, int W, const float* rendered, const float* gt, float lambda_dssim, float* loss_out, float* grad_rendered) {
    return guarded([&] {
        auto r = f32(rendered, {3, H, W}).set_requires_grad(true);
        auto rendered4 = r.unsqueeze(0), gt4 = f32(gt, {3, H, W}).unsqueeze(0);                             // :113-114
        auto l1_loss = torch::l1_loss(rendered4, gt4);                                                      // :121
        auto ssim_loss = 1.f - fused_ssim(rendered4, gt4, "valid", /*train=*/true);                         // :122
        torch::Tensor loss = (1.f - lambda_dssim) * l1_loss + lambda_dssim * ssim_loss;                     // :123-124
        loss.backward();
        *loss_out = loss.item<float>();
        put(r.grad(), grad_rendered);
    });
}
// BilateralGrid(num_images, W, H, L): identity grids (returned in `identity`), then grids += delta [N,12,L,H,W]; apply(rgb [3,h,w], image_idx) -> out, and the
// gradients of sum(out * v_out) + tv_weight * tv_loss() with respect to the grids and the image
REF_API int reflh_bilateral(int num_images, int gW, int gH, int gL, int image_idx, const float* delta, int h, int w, const float* rgb, const float* v_out, float tv_weight,
                            float* identity, float* out, float* tv, float* g_grids, float* g_rgb) {
    return guarded([&] {
        gs::training::BilateralGrid bg(num_images, gW, gH, gL);
        put(bg.parameters(), identity);
        {
            torch::NoGradGuard ng;
            bg.parameters().add_(f32(delta, {num_images, 12, gL, gH, gW}));
        }
        auto image = f32(rgb, {3, h, w}).set_requires_grad(true);
        auto o = bg.apply(image, image_idx);
        auto tvl = bg.tv_loss();
        auto loss = (o * f32(v_out, {3, h, w})).sum() + tv_weight * tvl;
        loss.backward();
        put(o, out);
        *tv = tvl.item<float>();
        put(bg.parameters().grad(), g_grids), put(image.grad(), g_rgb);
    });
}
// WarmupExponentialLR(optimizer with one group at lr0, gamma, warmup_steps, warmup_start_factor).step() x n -> the learning rate after every step
REF_API int reflh_warmup_schedule(double lr0, double gamma, int warmup_steps, double warmup_start_factor, int n, double* lrs) {
    return guarded([&] {
        using gs::training::FusedAdam;
        auto p = torch::zeros({1}).set_requires_grad(true);
        std::vector<torch::optim::OptimizerParamGroup> groups;
        groups.emplace_back(std::vector<torch::Tensor>{p}, std::unique_ptr<torch::optim::OptimizerOptions>(std::make_unique<FusedAdam::Options>(lr0)));
        FusedAdam opt(std::move(groups), std::make_unique<FusedAdam::Options>(lr0));
        gs::training::WarmupExponentialLR sched(opt, gamma, warmup_steps, warmup_start_factor, -1);
        for (int i = 0; i < n; ++i) {
            sched.step();
            lrs[i] = static_cast<FusedAdam::Options&>(opt.param_groups()[0].options()).lr();
        }
    });
}
