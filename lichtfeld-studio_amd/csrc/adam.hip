// K13 — fused Adam (replaces fast_gs::optimizer::adam_step; reference:
// fastgs/optimizer/include/adam_kernels.cuh:13-36, src/adam.cu:10-35, caller
// src/training/optimizers/fused_adam.cpp:22-95).
//
// Pure HBM streaming: 16 B read + 12 B write per element, 59 elements per
// Gaussian — the largest single HBM consumer of a training step (SURVEY.md §8a
// a10). The reference launches one scalar-load kernel per parameter tensor on the
// legacy default stream; here every lane moves 16 bytes per access, a grid-stride
// loop keeps ~2k workgroups resident, and lfs_adam_step_multi covers all six
// parameter groups of FusedAdam::step with ONE launch.
// Compiled with -ffp-contract=off: identical rounding to the reference formula.
#include <hip/hip_runtime.h>
#include "../../include/lfs_gsplat.h"
#include "lfs_prof.h"
#include "lfs_adam.cuh"

namespace lfs {

__device__ __forceinline__ void adam_range(float* __restrict__ param, float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                            const float* __restrict__ grad, const int64_t n, const AdamScalars& s,
                                            const int64_t worker, const int64_t n_workers) {
    // 16-byte body (pointers from the caching allocator / flat buckets are 16-byte aligned when
    // their element offset is a multiple of 4; otherwise fall back to scalars)
    const bool aligned = ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(exp_avg) |
                           reinterpret_cast<uintptr_t>(exp_avg_sq) | reinterpret_cast<uintptr_t>(grad)) & 15) == 0;
    if (aligned) {
        const int64_t n4 = n >> 2;
        for (int64_t i = worker; i < n4; i += n_workers) {
            float4 p = reinterpret_cast<float4*>(param)[i];
            float4 m = reinterpret_cast<float4*>(exp_avg)[i];
            float4 v = reinterpret_cast<float4*>(exp_avg_sq)[i];
            const float4 g = reinterpret_cast<const float4*>(grad)[i];
            adam_elem(p.x, m.x, v.x, g.x, s); adam_elem(p.y, m.y, v.y, g.y, s);
            adam_elem(p.z, m.z, v.z, g.z, s); adam_elem(p.w, m.w, v.w, g.w, s);
            reinterpret_cast<float4*>(param)[i] = p;
            reinterpret_cast<float4*>(exp_avg)[i] = m;
            reinterpret_cast<float4*>(exp_avg_sq)[i] = v;
        }
        for (int64_t i = (n4 << 2) + worker; i < n; i += n_workers) adam_elem(param[i], exp_avg[i], exp_avg_sq[i], grad[i], s);
    } else {
        for (int64_t i = worker; i < n; i += n_workers) adam_elem(param[i], exp_avg[i], exp_avg_sq[i], grad[i], s);
    }
}

__global__ void __launch_bounds__(256) adam_kernel(float* param, float* exp_avg, float* exp_avg_sq, const float* grad,
                                                   const int64_t n, const AdamScalars s) {
    adam_range(param, exp_avg, exp_avg_sq, grad, n, s, int64_t(blockIdx.x) * blockDim.x + threadIdx.x, int64_t(gridDim.x) * blockDim.x);
}

struct AdamMulti {
    lfs_adam_tensor t[LFS_ADAM_MAX_TENSORS];
    int32_t block_begin[LFS_ADAM_MAX_TENSORS + 1]; // workgroup ranges per tensor
    int32_t n;
};

__global__ void __launch_bounds__(256) adam_multi_kernel(const AdamMulti a) {
    int ti = 0;
#pragma unroll
    for (int k = 1; k < LFS_ADAM_MAX_TENSORS; ++k) if (k < a.n && int32_t(blockIdx.x) >= a.block_begin[k]) ti = k;
    const lfs_adam_tensor& t = a.t[ti];
    const AdamScalars s{t.lr, t.beta1, t.beta2, t.eps, t.bias_correction1_rcp, t.bias_correction2_sqrt_rcp};
    const int64_t nb = a.block_begin[ti + 1] - a.block_begin[ti];
    adam_range(t.param, t.exp_avg, t.exp_avg_sq, t.grad, t.n_elements, s,
               int64_t(int32_t(blockIdx.x) - a.block_begin[ti]) * blockDim.x + threadIdx.x, nb * blockDim.x);
}

static inline int adam_blocks(int64_t n) {
    int64_t b = (n / 4 + 255) / 256; // one float4 per lane per trip
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return int(b);
}

} // namespace lfs

extern "C" int lfs_adam_step(
    float* param, float* exp_avg, float* exp_avg_sq, const float* param_grad, int64_t n_elements,
    float lr, float beta1, float beta2, float eps, float bias_correction1_rcp, float bias_correction2_sqrt_rcp,
    lfs_stream_t stream) {
    if (n_elements == 0) return LFS_OK;
    if (n_elements < 0 || !param || !exp_avg || !exp_avg_sq || !param_grad) return LFS_E_INVALID;
    const lfs::AdamScalars s{lr, beta1, beta2, eps, bias_correction1_rcp, bias_correction2_sqrt_rcp};
    lfs::ProfScope prof("adam", (hipStream_t)stream);
    hipLaunchKernelGGL(lfs::adam_kernel, dim3(lfs::adam_blocks(n_elements)), dim3(256), 0, (hipStream_t)stream,
                       param, exp_avg, exp_avg_sq, param_grad, n_elements, s);
    return (int)hipGetLastError();
}

extern "C" int lfs_adam_step_multi(const lfs_adam_tensor* tensors, int32_t n_tensors, lfs_stream_t stream) {
    if (n_tensors == 0) return LFS_OK;
    if (!tensors || n_tensors < 0 || n_tensors > LFS_ADAM_MAX_TENSORS) return LFS_E_INVALID;
    lfs::AdamMulti a;
    a.n = 0; a.block_begin[0] = 0;
    int64_t total = 0;
    for (int i = 0; i < n_tensors; ++i) total += tensors[i].n_elements > 0 ? tensors[i].n_elements : 0;
    if (total == 0) return LFS_OK;
    // share ~4096 workgroups between the tensors in proportion to their size
    for (int i = 0; i < n_tensors; ++i) {
        const lfs_adam_tensor& t = tensors[i];
        if (t.n_elements <= 0) continue;
        if (!t.param || !t.exp_avg || !t.exp_avg_sq || !t.grad) return LFS_E_INVALID;
        int64_t b = (t.n_elements * 4096 + total - 1) / total;
        const int64_t bmax = (t.n_elements / 4 + 255) / 256;
        if (b > bmax) b = bmax;
        if (b < 1) b = 1;
        a.t[a.n] = t;
        a.block_begin[a.n + 1] = a.block_begin[a.n] + int32_t(b);
        ++a.n;
    }
    lfs::ProfScope prof("adam_multi", (hipStream_t)stream);
    hipLaunchKernelGGL(lfs::adam_multi_kernel, dim3(a.block_begin[a.n]), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

