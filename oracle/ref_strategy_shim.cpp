// ORACLE/_ref - TEST INFRASTRUCTURE ONLY.
// C entry points over the reference's OWN strategy layer and optimizer, compiled in place against libtorch on the CPU (oracle/Makefile, `make refstrategy`):
//   src/training/strategies/mcmc.cpp, default_strategy.cpp, strategy_utils.cpp, src/training/optimizers/fused_adam.cpp, scheduler.cpp   (whole files)
//   src/core/splat_data.cpp:200-287, 386-434   (SplatData's constructors, getters, increment_sh_degree)
// with their real headers (mcmc.hpp, default_strategy.hpp, istrategy.hpp, strategy_utils.hpp, fused_adam.hpp, scheduler.hpp, core/parameters.hpp,
// core/splat_data.hpp, adam_api.h). Edits made by sed on the way into the scratch directory, all documented in the Makefile: torch::kCUDA -> torch::kCPU, and the
// three random draws (torch::multinomial / randn_like / randn) routed through ref_stub/ref_hook.hpp, which calls the real function and records the result.
// Headers of absent libraries are stood in for under ref_stub/ (<expected>, nlohmann/json_fwd.hpp, glm, geometry/bounding_box.hpp, the logger, RenderOutput, Ops.h).
// The four device operators the layer calls are the reference's own kernels run on the CPU (oracle/_ref/libref_kernels.so): gsplat::relocation, add_noise,
// quats_to_rotmats and fast_gs::optimizer::adam_step_wrapper are defined below as their host launch sequences (cited).
// Used by oracle/make_golden_ref_strategy.py -> tests/golden/ref_strategy.npz, which the product's strategies are held to.
#include "Ops.h"
#include "adam_api.h"
#include "core/parameters.hpp"
#include "core/splat_data.hpp"
#include "optimizers/fused_adam.hpp"
#include "optimizers/scheduler.hpp"
#include "rasterization/rasterizer.hpp"
#include "ref_hook.hpp"
#include <cstring>
#include <functional>
#include <sstream>
#define private public // the optimizer state of the strategies is private and has no accessor; this translation unit only reads it (every other header
#include "default_strategy.hpp" // they pull in has been included above, so only the two strategy classes are affected)
#include "mcmc.hpp"
#undef private

#define REF_API extern "C" __attribute__((visibility("default")))

extern "C" {
void refk_relocation(int64_t N, float* opacities, float* scales, int* ratios, float* binoms, int n_max, float* new_opacities, float* new_scales);
void refk_add_noise(int64_t N, float* raw_opacities, float* raw_scales, float* raw_quats, float* noise, float* means, float current_lr);
void refk_quats_to_rotmats(int64_t N, const float* quats, float* rotmats);
void refk_adam_step(int64_t n, float* param, float* exp_avg, float* exp_avg_sq, const float* grad, float lr, float beta1, float beta2, float eps,
                    float bias_correction1_rcp, float bias_correction2_sqrt_rcp);
}

// ---- the host sides of the four operators (allocation + launch), over the reference's kernels --------------------------------------------------------------
namespace gsplat {
    at::Tensor quats_to_rotmats(const at::Tensor quats) { // QuatToRotmat.cpp:14-26
        auto q = quats.contiguous();
        auto R = at::empty({q.size(0), 3, 3}, q.options());
        refk_quats_to_rotmats(q.size(0), q.data_ptr<float>(), R.data_ptr<float>());
        return R;
    }
    std::tuple<at::Tensor, at::Tensor> relocation(at::Tensor opacities, at::Tensor scales, at::Tensor ratios, at::Tensor binoms, const int n_max) { // Relocation.cpp:15-32
        TORCH_CHECK(opacities.is_contiguous() && scales.is_contiguous() && ratios.is_contiguous() && binoms.is_contiguous(), "CHECK_INPUT: contiguous");
        TORCH_CHECK(ratios.scalar_type() == at::kInt, "ratios must be int32");
        auto new_o = at::empty_like(opacities), new_s = at::empty_like(scales);
        refk_relocation(opacities.numel(), opacities.data_ptr<float>(), scales.data_ptr<float>(), ratios.data_ptr<int>(), binoms.data_ptr<float>(), n_max,
                        new_o.data_ptr<float>(), new_s.data_ptr<float>());
        return {new_o, new_s};
    }
    void add_noise(at::Tensor raw_opacities, at::Tensor raw_scales, at::Tensor raw_quats, at::Tensor noise, at::Tensor means, const float current_lr) { // :34-50
        TORCH_CHECK(raw_opacities.is_contiguous() && raw_scales.is_contiguous() && raw_quats.is_contiguous() && noise.is_contiguous() && means.is_contiguous(),
                    "CHECK_INPUT: contiguous");
        refk_add_noise(means.size(0), raw_opacities.data_ptr<float>(), raw_scales.data_ptr<float>(), raw_quats.data_ptr<float>(), noise.data_ptr<float>(),
                       means.data_ptr<float>(), current_lr);
    }
} // namespace gsplat
namespace fast_gs::optimizer {
    void adam_step_wrapper(torch::Tensor& param, torch::Tensor& exp_avg, torch::Tensor& exp_avg_sq, const torch::Tensor& param_grad, const float lr, const float beta1,
                           const float beta2, const float eps, const float bias_correction1_rcp, const float bias_correction2_sqrt_rcp) { // adam_api.cu:8-33
        TORCH_CHECK(param.is_contiguous() && exp_avg.is_contiguous() && exp_avg_sq.is_contiguous() && param_grad.is_contiguous(), "adam: contiguous tensors");
        refk_adam_step(param.numel(), param.data_ptr<float>(), exp_avg.data_ptr<float>(), exp_avg_sq.data_ptr<float>(), param_grad.data_ptr<float>(), lr, beta1, beta2,
                       eps, bias_correction1_rcp, bias_correction2_sqrt_rcp);
    }
} // namespace fast_gs::optimizer

// ---- C API --------------------------------------------------------------------------------------------------------------------------------------------------
struct RefParams { // the fields of gs::param::OptimizationParameters the strategies read; refstrat_default_params fills it from the reference's defaults
    int64_t iterations, sh_degree_interval, refine_every, start_refine, stop_refine, reset_every, pause_refine_after_reset;
    float means_lr, shs_lr, opacity_lr, scaling_lr, rotation_lr, min_opacity, grad_threshold, prune_opacity, grow_scale3d, prune_scale3d, opacity_reg, scale_reg,
        lambda_dssim, init_opacity, init_scaling;
    int32_t max_cap, revised_opacity, sh_degree;
};

REF_API void refstrat_default_params(RefParams* p) {
    const gs::param::OptimizationParameters d;
    *p = RefParams{(int64_t)d.iterations, (int64_t)d.sh_degree_interval, (int64_t)d.refine_every, (int64_t)d.start_refine, (int64_t)d.stop_refine, (int64_t)d.reset_every,
                   (int64_t)d.pause_refine_after_reset, d.means_lr, d.shs_lr, d.opacity_lr, d.scaling_lr, d.rotation_lr, d.min_opacity, d.grad_threshold, d.prune_opacity,
                   d.grow_scale3d, d.prune_scale3d, d.opacity_reg, d.scale_reg, d.lambda_dssim, d.init_opacity, d.init_scaling, d.max_cap, d.revised_opacity ? 1 : 0,
                   d.sh_degree};
}

struct RefStrategy {
    std::unique_ptr<gs::training::IStrategy> s;
    gs::training::RenderOutput ro;
    int kind;
    torch::optim::Optimizer* opt() {
        return kind == 0 ? static_cast<gs::training::MCMC*>(s.get())->_optimizer.get() : static_cast<gs::training::DefaultStrategy*>(s.get())->_optimizer.get();
    }
    std::array<torch::Tensor*, 6> params() {
        auto& m = s->get_model();
        return {&m.means(), &m.sh0(), &m.shN(), &m.scaling_raw(), &m.rotation_raw(), &m.opacity_raw()};
    }
};

static thread_local std::string g_error;
REF_API const char* refstrat_last_error() { return g_error.c_str(); }
template <class F> static int guarded(F&& f) {
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return 1;
    }
}
static torch::Tensor f32(const float* p, std::vector<int64_t> shape) { return torch::from_blob(const_cast<float*>(p), shape, torch::kFloat32).clone(); }

// kind 0 = MCMC, 1 = DefaultStrategy. sh0 [N,1,3], shN [N,K,3], scaling [N,3], rotation [N,4], opacity [N,1] (raw), as SplatData holds them.
REF_API int refstrat_create(int kind, int64_t N, int64_t K, int sh_degree, const float* means, const float* sh0, const float* shN, const float* scaling,
                            const float* rotation, const float* opacity, float scene_scale, const RefParams* rp, RefStrategy** out) {
    return guarded([&] {
        gs::SplatData model(sh_degree, f32(means, {N, 3}), f32(sh0, {N, 1, 3}), f32(shN, {N, K, 3}), f32(scaling, {N, 3}), f32(rotation, {N, 4}), f32(opacity, {N, 1}),
                            scene_scale);
        gs::param::OptimizationParameters p;
        p.iterations = rp->iterations, p.sh_degree_interval = rp->sh_degree_interval, p.refine_every = rp->refine_every, p.start_refine = rp->start_refine;
        p.stop_refine = rp->stop_refine, p.reset_every = rp->reset_every, p.pause_refine_after_reset = rp->pause_refine_after_reset;
        p.means_lr = rp->means_lr, p.shs_lr = rp->shs_lr, p.opacity_lr = rp->opacity_lr, p.scaling_lr = rp->scaling_lr, p.rotation_lr = rp->rotation_lr;
        p.min_opacity = rp->min_opacity, p.grad_threshold = rp->grad_threshold, p.prune_opacity = rp->prune_opacity, p.grow_scale3d = rp->grow_scale3d;
        p.prune_scale3d = rp->prune_scale3d, p.max_cap = rp->max_cap, p.revised_opacity = rp->revised_opacity != 0, p.sh_degree = rp->sh_degree;
        auto r = std::make_unique<RefStrategy>();
        r->kind = kind;
        if (kind == 0) r->s = std::make_unique<gs::training::MCMC>(std::move(model));
        else r->s = std::make_unique<gs::training::DefaultStrategy>(std::move(model));
        r->s->initialize(p);
        *out = r.release();
    });
}
REF_API void refstrat_destroy(RefStrategy* r) { delete r; }
REF_API int64_t refstrat_size(RefStrategy* r) { return r->s->get_model().size(); }
REF_API int refstrat_active_sh_degree(RefStrategy* r) { return r->s->get_model().get_active_sh_degree(); }
REF_API int refstrat_is_refining(RefStrategy* r, int iter) { return r->s->is_refining(iter) ? 1 : 0; }
REF_API double refstrat_lr(RefStrategy* r, int group) { return static_cast<gs::training::FusedAdam::Options&>(r->opt()->param_groups()[group].options()).lr(); }

// what: 0 parameter, 1 exp_avg, 2 exp_avg_sq. Returns the element count (out may be null to size), -1 if the optimizer holds no state for the parameter yet.
REF_API int64_t refstrat_get(RefStrategy* r, int which, int what, float* out) {
    torch::Tensor t = *r->params()[which];
    if (what) {
        auto it = r->opt()->state().find(r->opt()->param_groups()[which].params()[0].unsafeGetTensorImpl());
        if (it == r->opt()->state().end()) return -1;
        auto* st = static_cast<gs::training::FusedAdam::AdamParamState*>(it->second.get());
        t = what == 1 ? st->exp_avg : st->exp_avg_sq;
    }
    t = t.detach().contiguous();
    if (out) std::memcpy(out, t.data_ptr<float>(), sizeof(float) * t.numel());
    return t.numel();
}
REF_API int64_t refstrat_step_count(RefStrategy* r, int which) {
    auto it = r->opt()->state().find(r->opt()->param_groups()[which].params()[0].unsafeGetTensorImpl());
    return it == r->opt()->state().end() ? -1 : static_cast<gs::training::FusedAdam::AdamParamState*>(it->second.get())->step_count;
}
REF_API int refstrat_set_grads(RefStrategy* r, const float* const* grads) {
    return guarded([&] {
        auto ps = r->params();
        for (int i = 0; i < 6; ++i) ps[i]->mutable_grad() = f32(grads[i], ps[i]->sizes().vec());
    });
}
REF_API int refstrat_set_densification_info(RefStrategy* r, const float* info) {
    return guarded([&] { r->s->get_model()._densification_info = f32(info, {2, r->s->get_model().size()}); });
}
REF_API int refstrat_step(RefStrategy* r, int iter) {
    return guarded([&] { r->s->step(iter); });
}
REF_API int refstrat_post_backward(RefStrategy* r, int iter, uint64_t seed) {
    return guarded([&] {
        torch::manual_seed(seed);
        ref_hook::log().clear();
        r->s->post_backward(iter, r->ro);
    });
}
REF_API int refstrat_remove_gaussians(RefStrategy* r, const uint8_t* mask) {
    return guarded([&] {
        const int64_t N = r->s->get_model().size();
        r->s->remove_gaussians(torch::from_blob(const_cast<uint8_t*>(mask), {N}, torch::kUInt8).clone().to(torch::kBool));
    });
}
// the random draws of the last post_backward, in call order
REF_API int64_t refstrat_log_count() { return (int64_t)ref_hook::log().size(); }
REF_API const char* refstrat_log_name(int64_t i) { return ref_hook::log()[i].first.c_str(); }
REF_API int64_t refstrat_log_numel(int64_t i) { return ref_hook::log()[i].second.numel(); }
REF_API int refstrat_log_is_int64(int64_t i) { return ref_hook::log()[i].second.scalar_type() == torch::kLong ? 1 : 0; }
REF_API void refstrat_log_copy(int64_t i, void* out) {
    auto t = ref_hook::log()[i].second.contiguous();
    std::memcpy(out, t.data_ptr(), t.numel() * t.element_size());
}
