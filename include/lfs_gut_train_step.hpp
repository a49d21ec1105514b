// lfs::GutTrainStep - the extension a patched reference trainer includes BESIDE its own gsplat/Ops.h (this header declares nothing Ops.h declares, so both can be
// included in one translation unit: oracle/ref_raster_shim.cpp does, under `make -C oracle reflink`, next to the reference's own FusedAdam / SplatData / Camera).
// Implemented in lichtfeld-studio_amd/csrc/torch_ops.cpp (liblfs_gsplat_torch.so) over lfs_gut_train_step_ex (include/lfs_gsplat.h, csrc/gut_step.hip).
#pragma once
#include <torch/torch.h>
#include <array>
#include <cstdint>

// ---- The --gut training step as ONE call for a libtorch caller (extension; csrc/gut_step.hip behind include/lfs_gsplat.h: lfs_gut_train_step) ----------------
// What Trainer::train_step (src/training/trainer.cpp:579-770) does between `rasterize()` and `optimizer->step()` on the --gut path with the MSE loss - projection,
// SH colours, tile intersection, compositing, loss, the whole backward and FusedAdam::step (fused_adam.cpp:22-95) for all six parameter groups - without the host
// synchronisation of gsplat/Intersect.cpp:75-76 and without gradient tensors. INTEGRATION.md shows the trainer.cpp patch. One object per model / image size:
// it owns the workspace (a byte tensor from the caching allocator) and the capacity bookkeeping of the speculative intersection lists (an attempt that did not
// fit updated nothing and is re-run with a larger workspace; `retries()` counts them - a few after start-up or a densification, none in steady state).
namespace lfs {
struct AdamGroupState {   // one FusedAdam parameter group: moments + the scalars FusedAdam::step computes for this iteration (fused_adam.cpp:78-92)
    torch::Tensor exp_avg, exp_avg_sq;
    float lr = 0.f, beta1 = 0.9f, beta2 = 0.999f, eps = 1e-15f, bias_correction1_rcp = 1.f, bias_correction2_sqrt_rcp = 1.f;
};
class GutTrainStep {
public:
    explicit GutTrainStep(uint32_t tile_size = 16, int64_t initial_capacity = 0);
    // Parameters (updated in place) and `adam` in FusedAdam's group order: means, sh0, shN, raw_scales, raw_quats, raw_opacities. viewmat [4,4] / [1,4,4],
    // K [3,3] / [1,3,3], background [3] (optional), target_chw [3,H,W]; loss: a 1-element device tensor, receives loss_weight * mse(clamp(render,0,1), target).
    // Needs at least one higher-degree SH coefficient (shN.size(1) > 0). Returns the number of tile intersections of the view.
    // next_viewmat (round 6, optional): the view the NEXT call will render, when the trainer knows it (its dataloader has the next camera queued,
    // src/training/dataloader.cpp:115-119). The step's tail (lfs_gut_train_step_ex: SH backward + the six Adam updates in one launch) then also evaluates that view's SH
    // colours from the coefficient rows as they leave their Adam update, and the next call - recognised by the same tensor (storage address and version), the same
    // parameter tensors with nothing written to means / sh0 / shN in between (their autograd versions), the same N / K / degree / workspace - skips its SH colour
    // kernel. A next call for any other view, or after a densification, simply evaluates its colours itself: same results either way.
    int64_t step(torch::Tensor& means, torch::Tensor& sh0, torch::Tensor& shN, torch::Tensor& raw_scales, torch::Tensor& raw_quats, torch::Tensor& raw_opacities,
                 const std::array<AdamGroupState, 6>& adam, uint32_t sh_degree, const torch::Tensor& viewmat, const torch::Tensor& K, uint32_t image_width,
                 uint32_t image_height, const at::optional<torch::Tensor>& background, const torch::Tensor& target_chw, float loss_weight, torch::Tensor& loss,
                 float scale_reg = 0.f, float opacity_reg = 0.f, const at::optional<torch::Tensor>& next_viewmat = at::nullopt);
    torch::Tensor render() const;   // [H,W,3] view of the last step's un-clamped image (valid until the next step)
    torch::Tensor alpha() const;    // [H,W]
    torch::Tensor radii() const;    // int32 [N,2]
    int64_t capacity() const { return capacity_; }
    int retries() const { return retries_; }
    int colour_launches_saved() const { return colours_saved_; }   // steps that found their SH colours prepared by the step before
private:
    void ensure(uint32_t N, uint32_t W, uint32_t H, const torch::Tensor& like);
    uint32_t tile_;
    int64_t capacity_, assumed_longest_ = 1024, stamp_ = 0, n_isects_ = 0, longest_ = 0;
    int retries_ = 0;
    uint32_t N_ = 0, W_ = 0, H_ = 0, flags_ = 0;
    int64_t cap_built_ = -1;
    torch::Tensor ws_, counts_;
    size_t off_render_ = 0, off_alpha_ = 0, off_radii_ = 0, ws_bytes_ = 0;
    struct ColoursFor {   // what the colours in the workspace were evaluated for (valid: by the previous call's tail, which fitted its buffers)
        bool valid = false; const void* viewmat = nullptr; uint32_t viewmat_version = 0; const void* ws = nullptr; const void* param[3] = {nullptr, nullptr, nullptr};
        uint32_t param_version[3] = {0, 0, 0}; uint32_t N = 0, K = 0, degree = 0;
    } colours_for_;
    int colours_saved_ = 0;
};
} // namespace lfs
