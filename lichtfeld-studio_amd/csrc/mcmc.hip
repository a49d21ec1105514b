// K10-K12 — MCMC relocation, SGLD noise injection and quat->rotmat (replace
// gsplat::relocation / add_noise / quats_to_rotmats; reference:
// gsplat/RelocationCUDA.cu:12-43 and :88-144, gsplat/QuatToRotmatCUDA.cu:14-39).
// Small streaming kernels, one lane per Gaussian, vector loads where the layout
// allows (quats are 16-byte rows).
#include "lfs_math.cuh"
#include "../../include/lfs_gsplat.h"

namespace lfs {

__global__ void __launch_bounds__(256) quats_to_rotmats_kernel(const uint32_t N, const float* __restrict__ quats, float* __restrict__ rotmats) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float4 q = reinterpret_cast<const float4*>(quats)[i];
    const m3 R = quat_to_rotmat(q.x, q.y, q.z, q.w);
    float* o = rotmats + size_t(i) * 9;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) o[3 * r + c] = R.m[r][c];
}

// "3D Gaussian Splatting as Markov Chain Monte Carlo", Eq. 9
__global__ void __launch_bounds__(256) relocation_kernel(
    const uint32_t N, const float* __restrict__ opacities, const float* __restrict__ scales, const int32_t* __restrict__ ratios,
    const float* __restrict__ binoms, const int32_t n_max, float* __restrict__ new_opacities, float* __restrict__ new_scales) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int32_t n = ratios[i];
    const float o = opacities[i];
    const float no = 1.f - powf(1.f - o, 1.f / float(n));
    new_opacities[i] = no;
    float denom = 0.f;
    for (int32_t a = 1; a <= n; ++a) {
        float sgn = 1.f;     // (-1)^k
        for (int32_t k = 0; k <= a - 1; ++k) {
            const float bin = binoms[(a - 1) * n_max + k];
            // pow(-1,k) / sqrt(k+1) * no^(k+1); powf keeps the reference's rounding of the power term
            const float term = (sgn / sqrtf(float(k + 1))) * powf(no, float(k + 1));
            denom += bin * term;
            sgn = -sgn;
        }
    }
    const float coeff = o / denom;
#pragma unroll
    for (int d = 0; d < 3; ++d) new_scales[3 * i + d] = coeff * scales[3 * i + d];
}

__global__ void __launch_bounds__(256) add_noise_kernel(
    const uint32_t N, const float* __restrict__ raw_opacities, const float* __restrict__ raw_scales, const float* __restrict__ raw_quats,
    const float* __restrict__ noise, float* __restrict__ means, const float current_lr) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float s2[3] = {__expf(2.f * raw_scales[3 * i]), __expf(2.f * raw_scales[3 * i + 1]), __expf(2.f * raw_scales[3 * i + 2])};
    const float4 q = reinterpret_cast<const float4*>(raw_quats)[i];
    const m3 R = quat_to_rotmat(q.x, q.y, q.z, q.w, 1e12f);
    // covariance = R diag(s2) R^T ; transformed noise = covariance * noise
    const f3 nz{noise[3 * i], noise[3 * i + 1], noise[3 * i + 2]};
    m3 cov;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            cov.m[r][c] = R.m[r][0] * s2[0] * R.m[c][0] + R.m[r][1] * s2[1] * R.m[c][1] + R.m[r][2] * s2[2] * R.m[c][2];
    const f3 tn = mul(cov, nz);
    const float opacity = 1.f / (1.f + __expf(-raw_opacities[i]));
    const float op_sigmoid = 1.f / (1.f + __expf(100.f * opacity - 0.5f));
    const float nf = current_lr * op_sigmoid;
    means[3 * i] += nf * tn.x; means[3 * i + 1] += nf * tn.y; means[3 * i + 2] += nf * tn.z;
}

} // namespace lfs

extern "C" int lfs_quats_to_rotmats(uint32_t N, const float* quats, float* rotmats, lfs_stream_t stream) {
    if (N == 0) return LFS_OK;
    if (!quats || !rotmats) return LFS_E_INVALID;
    hipLaunchKernelGGL(lfs::quats_to_rotmats_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, quats, rotmats);
    return (int)hipGetLastError();
}

extern "C" int lfs_relocation(
    uint32_t N, const float* opacities, const float* scales, const int32_t* ratios, const float* binoms, int32_t n_max,
    float* new_opacities, float* new_scales, lfs_stream_t stream) {
    if (N == 0) return LFS_OK;
    if (!opacities || !scales || !ratios || !binoms || !new_opacities || !new_scales || n_max <= 0) return LFS_E_INVALID;
    hipLaunchKernelGGL(lfs::relocation_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       N, opacities, scales, ratios, binoms, n_max, new_opacities, new_scales);
    return (int)hipGetLastError();
}

extern "C" int lfs_add_noise(
    uint32_t N, const float* raw_opacities, const float* raw_scales, const float* raw_quats, const float* noise,
    float* means, float current_lr, lfs_stream_t stream) {
    if (N == 0) return LFS_OK;
    if (!raw_opacities || !raw_scales || !raw_quats || !noise || !means) return LFS_E_INVALID;
    hipLaunchKernelGGL(lfs::add_noise_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       N, raw_opacities, raw_scales, raw_quats, noise, means, current_lr);
    return (int)hipGetLastError();
}
