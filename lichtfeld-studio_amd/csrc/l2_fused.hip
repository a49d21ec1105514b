// Fused L2 glue (extensions, not in gsplat/Ops.h): the element-wise libtorch work the reference's trainer wraps
// around the operators, as single HBM passes.
//   * activations fwd / bwd: gs::SplatData::get_rotation / get_scaling / get_opacity
//     (src/core/splat_data.cpp:267-286: normalize, exp, sigmoid) and their autograd backward;
//   * photometric MSE of the clamped render against a CHW target with its gradient in the rasterizer's HWC
//     layout (the rasterizer-only loss of SURVEY.md §8d; rasterizer.cpp:399 clamps the image to [0,1]).
// All of it is HBM-bound streaming: one thread per Gaussian / pixel, 16-byte accesses where the layout allows.
#include "lfs_math.cuh"
#include "lfs_prof.h"
#include "../../include/lfs_gsplat.h"

namespace lfs {

__global__ void __launch_bounds__(256) activations_fwd_kernel(
    const uint32_t N, const float* __restrict__ raw_quats, const float* __restrict__ raw_scales, const float* __restrict__ raw_opacities,
    float* __restrict__ quats, float* __restrict__ scales, float* __restrict__ opacities) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    const float4 q = reinterpret_cast<const float4*>(raw_quats)[g];
    // torch::nn::functional::normalize: x / max(||x||_2, 1e-12)
    const float den = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    reinterpret_cast<float4*>(quats)[g] = make_float4(q.x / den, q.y / den, q.z / den, q.w / den); // true divisions, as ATen
#pragma unroll
    for (int k = 0; k < 3; ++k) scales[3 * g + k] = expf(raw_scales[3 * g + k]);
    opacities[g] = 1.f / (1.f + expf(-raw_opacities[g]));
}

// grads w.r.t. the raw parameters from grads w.r.t. the activated ones. ACCUM adds to the outputs.
template <bool ACCUM>
__global__ void __launch_bounds__(256) activations_bwd_kernel(
    const uint32_t N, const float* __restrict__ raw_quats, const float* __restrict__ scales, const float* __restrict__ opacities,
    const float* __restrict__ v_quats, const float* __restrict__ v_scales, const float* __restrict__ v_opacities,
    const float v_scale_reg, const float v_opacity_reg,
    float* __restrict__ g_raw_quats, float* __restrict__ g_raw_scales, float* __restrict__ g_raw_opacities) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    const float4 q = reinterpret_cast<const float4*>(raw_quats)[g];
    const float4 v = reinterpret_cast<const float4*>(v_quats)[g];
    const float nrm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    float4 gq;
    if (nrm > 1e-12f) { // y = x / n : dx = (v - (v.y) y) / n
        const float inv = 1.f / nrm;
        const float4 y = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
        const float d = v.x * y.x + v.y * y.y + v.z * y.z + v.w * y.w;
        gq = make_float4((v.x - d * y.x) * inv, (v.y - d * y.y) * inv, (v.z - d * y.z) * inv, (v.w - d * y.w) * inv);
    } else { // clamped denominator: y = x / 1e-12
        gq = make_float4(v.x * 1e12f, v.y * 1e12f, v.z * 1e12f, v.w * 1e12f);
    }
    float4* gqo = reinterpret_cast<float4*>(g_raw_quats) + g;
    if (ACCUM) { const float4 o = *gqo; gq = make_float4(o.x + gq.x, o.y + gq.y, o.z + gq.z, o.w + gq.w); }
    *gqo = gq;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float gs = (v_scales[3 * g + k] + v_scale_reg) * scales[3 * g + k];
        if (ACCUM) g_raw_scales[3 * g + k] += gs; else g_raw_scales[3 * g + k] = gs;
    }
    const float o = opacities[g];
    const float go = (v_opacities[g] + v_opacity_reg) * o * (1.f - o);
    if (ACCUM) g_raw_opacities[g] += go; else g_raw_opacities[g] = go;
}

// loss += sum((clamp(x,0,1) - t)^2) * inv_numel * weight; v_render = 2 (clamp(x) - t) * inv_numel * weight where 0 <= x <= 1
// (torch::clamp passes the gradient on the closed interval). render HWC, target CHW.
// CHW = true: render / v_render are [3,H,W] and NOT clamped (the fastgs path: fast_rasterizer.cpp hands the image on as is)
template <bool CHW>
__global__ void __launch_bounds__(256) mse_loss_kernel(
    const uint32_t H, const uint32_t W, const float* __restrict__ render, const float* __restrict__ target,
    const float scale, float* __restrict__ v_render, float* __restrict__ loss, const bool clamp = !CHW) {
    const uint32_t P = H * W;
    float acc = 0.f;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const size_t ri = CHW ? size_t(c) * P + p : 3 * size_t(p) + c;
            const float x = render[ri];
            const float d = (clamp ? fminf(fmaxf(x, 0.f), 1.f) : x) - target[size_t(c) * P + p];
            acc += d * d;
            v_render[ri] = (!clamp || (x >= 0.f && x <= 1.f)) ? 2.f * d * scale : 0.f;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (part[0] + part[1] + part[2] + part[3]) * scale);
}

} // namespace lfs

extern "C" int lfs_activations_fwd(uint32_t N, const float* raw_quats, const float* raw_scales, const float* raw_opacities,
                                   float* quats, float* scales, float* opacities, lfs_stream_t stream) {
    if (N == 0) return LFS_OK;
    if (!raw_quats || !raw_scales || !raw_opacities || !quats || !scales || !opacities) return LFS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    lfs::ProfScope prof("activations_fwd", s);
    hipLaunchKernelGGL(lfs::activations_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, raw_quats, raw_scales, raw_opacities, quats, scales, opacities);
    return (int)hipGetLastError();
}

extern "C" int lfs_activations_bwd(uint32_t N, const float* raw_quats, const float* scales, const float* opacities,
                                   const float* v_quats, const float* v_scales, const float* v_opacities, float scale_reg, float opacity_reg, int accumulate,
                                   float* g_raw_quats, float* g_raw_scales, float* g_raw_opacities, lfs_stream_t stream) {
    if (N == 0) return LFS_OK;
    if (!raw_quats || !scales || !opacities || !v_quats || !v_scales || !v_opacities || !g_raw_quats || !g_raw_scales || !g_raw_opacities) return LFS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    lfs::ProfScope prof("activations_bwd", s);
    const dim3 grid((N + 255) / 256), block(256);
    // regularisers of trainer.cpp:132-158: scale_reg * mean(scales) (mean over 3N values) and opacity_reg * mean(opacities)
    const float vs = scale_reg / (3.f * float(N)), vo = opacity_reg / float(N);
    if (accumulate) hipLaunchKernelGGL(lfs::activations_bwd_kernel<true>, grid, block, 0, s, N, raw_quats, scales, opacities, v_quats, v_scales, v_opacities, vs, vo, g_raw_quats, g_raw_scales, g_raw_opacities);
    else hipLaunchKernelGGL(lfs::activations_bwd_kernel<false>, grid, block, 0, s, N, raw_quats, scales, opacities, v_quats, v_scales, v_opacities, vs, vo, g_raw_quats, g_raw_scales, g_raw_opacities);
    return (int)hipGetLastError();
}

extern "C" int lfs_mse_loss_fwd_bwd(uint32_t H, uint32_t W, const float* render_hwc, const float* target_chw, float weight,
                                    float* v_render_hwc, float* loss, lfs_stream_t stream) {
    if (H == 0 || W == 0) return LFS_OK;
    if (!render_hwc || !target_chw || !v_render_hwc || !loss) return LFS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    lfs::ProfScope prof("mse_loss", s);
    const uint32_t P = H * W;
    const uint32_t blocks = (P + 255) / 256 < 2048 ? (P + 255) / 256 : 2048;
    hipLaunchKernelGGL(lfs::mse_loss_kernel<false>, dim3(blocks), dim3(256), 0, s, H, W, render_hwc, target_chw, weight / float(3u * P), v_render_hwc, loss);
    return (int)hipGetLastError();
}

extern "C" int lfs_mse_loss_chw_fwd_bwd(uint32_t H, uint32_t W, const float* render_chw, const float* target_chw, float weight,
                                        float* v_render_chw, float* loss, lfs_stream_t stream) {
    if (H == 0 || W == 0) return LFS_OK;
    if (!render_chw || !target_chw || !v_render_chw || !loss) return LFS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    lfs::ProfScope prof("mse_loss", s);
    const uint32_t P = H * W;
    const uint32_t blocks = (P + 255) / 256 < 2048 ? (P + 255) / 256 : 2048;
    hipLaunchKernelGGL(lfs::mse_loss_kernel<true>, dim3(blocks), dim3(256), 0, s, H, W, render_chw, target_chw, weight / float(3u * P), v_render_chw, loss);
    return (int)hipGetLastError();
}

// general form (see lfs_photometric_loss_ex_fwd_bwd)
extern "C" int lfs_mse_loss_ex_fwd_bwd(uint32_t H, uint32_t W, const float* render, uint32_t render_is_chw, uint32_t clamp_render, const float* target_chw, float weight,
                                       float* v_render, float* loss, lfs_stream_t stream) {
    if (H == 0 || W == 0) return LFS_OK;
    if (!render || !target_chw || !v_render || !loss) return LFS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    lfs::ProfScope prof("mse_loss", s);
    const uint32_t P = H * W;
    const uint32_t blocks = (P + 255) / 256 < 2048 ? (P + 255) / 256 : 2048;
    if (render_is_chw) hipLaunchKernelGGL(lfs::mse_loss_kernel<true>, dim3(blocks), dim3(256), 0, s, H, W, render, target_chw, weight / float(3u * P), v_render, loss, clamp_render != 0);
    else hipLaunchKernelGGL(lfs::mse_loss_kernel<false>, dim3(blocks), dim3(256), 0, s, H, W, render, target_chw, weight / float(3u * P), v_render, loss, clamp_render != 0);
    return (int)hipGetLastError();
}
