// ORACLE/_ref - TEST INFRASTRUCTURE ONLY. The three gsplat operators the strategy layer calls, declared as gsplat/Ops.h:46-65 declares them (the real header pulls
// in the camera model headers). ref_strategy_shim.cpp defines them over the reference's own device kernels run on the CPU (oracle/_ref/libref_kernels.so).
#pragma once
#include <ATen/core/Tensor.h>
#include <tuple>
namespace gsplat {
    at::Tensor quats_to_rotmats(const at::Tensor quats);
    std::tuple<at::Tensor, at::Tensor> relocation(at::Tensor opacities, at::Tensor scales, at::Tensor ratios, at::Tensor binoms, const int n_max);
    void add_noise(at::Tensor raw_opacities, at::Tensor raw_scales, at::Tensor raw_quats, at::Tensor noise, at::Tensor means, const float current_lr);
} // namespace gsplat
