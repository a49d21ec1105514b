// ORACLE/_ref — TEST INFRASTRUCTURE ONLY.
// The reference's OWN fastgs (EWA) rasterizer run on the CPU: /root/reference/fastgs/rasterization/src/forward.cu:14-199 and backward.cu:14-116 (the
// host sequences: preprocess -> depth sort -> instances -> tile sort -> ranges / buckets -> blend, and blend_backward -> preprocess_backward) together
// with the kernels they launch (include/kernels_forward.cuh, kernels_backward.cuh, kernel_utils.cuh, buffer_utils.h, utils/helper_math.h, utils.h) are
// compiled IN PLACE as host C++ by `make -C oracle refk_fastgs`. The recipe pipes the two .cu files through sed into a scratch directory that is
// deleted after the compile - the only edits are the launch syntax, `kernel<<<grid, block>>>(args)` -> `cuemu::launcher(kernel, grid, block)(args)`,
// which no host compiler parses otherwise, one shift by 32 in kernels_forward.cuh:278 that CUDA defines (0) and C++ does not, and one __syncwarp() in front
// of the stage refill of blend_backward_cu (kernels_backward.cuh:332), where the kernel relies on warp lock-step (see the Makefile);
// nothing of the reference is copied into the repository, oracle/_ref only receives the .so.
// CUDA's execution model: oracle/ref_emul/ (cuda_emul.h fibers and 32-lane warps, cuda_runtime.h vector types / runtime calls / warp intrinsics,
// cooperative_groups.h, cub/cub.cuh: the published semantics of the three CUB calls - CUB is a toolkit dependency, not in the reference tree).
// oracle/make_golden_refk.py runs the entry point below and writes tests/golden/refk_fastgs_*.npz; the oracle restatement (oracle_fastgs.hpp) and the
// HIP kernels (csrc/fastgs_*.hip) are both checked against those files: this pins SURVEY.md §8f row 1 to the reference itself.
#define __CUDACC__ 1   // helper_math.h: skip its host re-definitions of fminf / fmaxf / min / max / rsqrtf (math.h and cuda_emul.h provide them)
#include "cuda_runtime.h"
#include "cooperative_groups.h"
#include "k_fastgs_forward.inc"
#include "k_fastgs_backward.inc"

#include <vector>

#define REFK_API extern "C" __attribute__((visibility("default")))

// One forward + backward of fast_gs::rasterization (rasterization_api.cu:15-181 is the libtorch wrapper around exactly these two calls: buffers from
// resize callbacks, gradients zero-initialised, densification_info optional). All pointers are host memory. counts = {n_visible_primitives,
// n_instances, n_buckets}. grad_w2c is not requested (the trainer's cameras have no gradient: rasterization_api.cu:74-77).
REFK_API void refk_fastgs_fwd_bwd(
    int n_primitives, int active_sh_bases, int total_bases_sh_rest, int width, int height, float fx, float fy, float cx, float cy, float near_plane,
    float far_plane, const float* means, const float* scales_raw, const float* rotations_raw, const float* opacities_raw, const float* sh0,
    const float* sh_rest, const float* w2c, const float* cam_position, const float* grad_image, const float* grad_alpha,
    float* image, float* alpha, float* grad_means, float* grad_scales_raw, float* grad_rotations_raw, float* grad_opacities_raw, float* grad_sh0,
    float* grad_sh_rest, float* densification_info /* [2,N] or null */, int32_t* counts) {
    using namespace fast_gs::rasterization;
    std::vector<char> prim, tile, inst, bucket;
    auto resizer = [](std::vector<char>& v) { return [&v](size_t n) { v.assign(n + 256, 0); return v.data(); }; };
    auto [n_visible, n_instances, n_buckets, sel_p, sel_i] = forward(
        resizer(prim), resizer(tile), resizer(inst), resizer(bucket),
        reinterpret_cast<const float3*>(means), reinterpret_cast<const float3*>(scales_raw), reinterpret_cast<const float4*>(rotations_raw), opacities_raw,
        reinterpret_cast<const float3*>(sh0), reinterpret_cast<const float3*>(sh_rest), reinterpret_cast<const float4*>(w2c),
        reinterpret_cast<const float3*>(cam_position), image, alpha, n_primitives, active_sh_bases, total_bases_sh_rest, width, height, fx, fy, cx, cy,
        near_plane, far_plane);
    counts[0] = n_visible; counts[1] = n_instances; counts[2] = n_buckets;
    if (!grad_image) return;
    const size_t N = size_t(n_primitives);
    std::fill(grad_means, grad_means + 3 * N, 0.f); std::fill(grad_scales_raw, grad_scales_raw + 3 * N, 0.f);
    std::fill(grad_rotations_raw, grad_rotations_raw + 4 * N, 0.f); std::fill(grad_opacities_raw, grad_opacities_raw + N, 0.f);
    std::fill(grad_sh0, grad_sh0 + 3 * N, 0.f); std::fill(grad_sh_rest, grad_sh_rest + 3 * N * size_t(total_bases_sh_rest), 0.f);
    std::vector<float> mean2d_helper(2 * N, 0.f), conic_helper(3 * N, 0.f);
    backward(grad_image, grad_alpha, image, alpha, reinterpret_cast<const float3*>(means), reinterpret_cast<const float3*>(scales_raw),
             reinterpret_cast<const float4*>(rotations_raw), reinterpret_cast<const float3*>(sh_rest), reinterpret_cast<const float4*>(w2c),
             reinterpret_cast<const float3*>(cam_position), prim.data(), tile.data(), inst.data(), bucket.data(),
             reinterpret_cast<float3*>(grad_means), reinterpret_cast<float3*>(grad_scales_raw), reinterpret_cast<float4*>(grad_rotations_raw), grad_opacities_raw,
             reinterpret_cast<float3*>(grad_sh0), reinterpret_cast<float3*>(grad_sh_rest), reinterpret_cast<float2*>(mean2d_helper.data()), conic_helper.data(),
             nullptr, densification_info, n_primitives, n_visible, n_instances, n_buckets, sel_p, sel_i, active_sh_bases, total_bases_sh_rest, width, height,
             fx, fy, cx, cy);
}
