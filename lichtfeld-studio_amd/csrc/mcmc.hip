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

// ---------------------------------------------------------------------------------------------------------------------------
// lfs_mcmc_relocate: MCMC::relocate_gs without a host round trip (see include/lfs_gsplat.h)
// ---------------------------------------------------------------------------------------------------------------------------
constexpr uint32_t RL_BLOCK = 1024; // elements per workgroup of the prefix sums (256 threads x 4)
LFS_DI float relocate_weight(const float raw_o, const float4 q, const float min_opacity, bool& dead) {
    const float o = 1.f / (1.f + expf(-raw_o));
    dead = o <= min_opacity || (q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w) < 1e-8f;
    return dead ? 0.f : o;
}
// sums of 1024 weights each, in a fixed order (thread-sequential, then a tree in LDS): deterministic fp64
__global__ void __launch_bounds__(256) relocate_block_sums_kernel(const uint32_t N, const float* __restrict__ raw_o, const float* __restrict__ raw_q,
                                                                  const float min_opacity, double* __restrict__ block_sums, int32_t* __restrict__ counts,
                                                                  int32_t* __restrict__ n_dead) {
    __shared__ double s[256];
    __shared__ int32_t sd[256];
    const uint32_t base = blockIdx.x * RL_BLOCK + threadIdx.x * 4;
    double acc = 0.0; int32_t nd = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t i = base + k;
        if (i < N) {
            bool dead;
            acc += double(relocate_weight(raw_o[i], reinterpret_cast<const float4*>(raw_q)[i], min_opacity, dead));
            nd += dead ? 1 : 0;
            counts[i] = 0;
        }
    }
    s[threadIdx.x] = acc; sd[threadIdx.x] = nd;
    __syncthreads();
    for (uint32_t m = 128; m >= 1; m >>= 1) {
        if (threadIdx.x < m) { s[threadIdx.x] += s[threadIdx.x + m]; sd[threadIdx.x] += sd[threadIdx.x + m]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { block_sums[blockIdx.x] = s[0]; if (n_dead && sd[0]) atomicAdd(n_dead, sd[0]); }
}
// exclusive scan of the block sums by one workgroup (<= 16 M Gaussians: 16 384 sums); block_sums[nb] = total
__global__ void __launch_bounds__(256) relocate_scan_blocks_kernel(const uint32_t nb, double* __restrict__ block_sums) {
    __shared__ double s[256];
    __shared__ double carry;
    if (threadIdx.x == 0) carry = 0.0;
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const double v = i < nb ? block_sums[i] : 0.0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t off = 1; off < 256; off <<= 1) { // Hillis-Steele inclusive scan
            const double add = threadIdx.x >= off ? s[threadIdx.x - off] : 0.0;
            __syncthreads();
            s[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < nb) block_sums[i] = carry + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) carry += s[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[nb] = carry;
}
// inclusive prefix sums of the weights (cdf[i]); thread t of a workgroup owns elements 4t .. 4t+3 of its 1024
__global__ void __launch_bounds__(256) relocate_cdf_kernel(const uint32_t N, const float* __restrict__ raw_o, const float* __restrict__ raw_q,
                                                           const float min_opacity, const double* __restrict__ block_sums, double* __restrict__ cdf) {
    __shared__ double s[256];
    const uint32_t base = blockIdx.x * RL_BLOCK + threadIdx.x * 4;
    double w[4], acc = 0.0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t i = base + k;
        bool dead = true;
        w[k] = i < N ? double(relocate_weight(raw_o[i], reinterpret_cast<const float4*>(raw_q)[i], min_opacity, dead)) : 0.0;
        acc += w[k];
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        const double add = threadIdx.x >= off ? s[threadIdx.x - off] : 0.0;
        __syncthreads();
        s[threadIdx.x] += add;
        __syncthreads();
    }
    double run = block_sums[blockIdx.x] + s[threadIdx.x] - acc;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) { run += w[k]; if (base + k < N) cdf[base + k] = run; }
}
// dead i: source = first j with cdf[j] > u_i * total (never a dead j: their weight is 0, so cdf does not move across them)
__global__ void __launch_bounds__(256) relocate_sample_kernel(const uint32_t N, const float* __restrict__ raw_o, const float* __restrict__ raw_q, const float min_opacity,
                                                              const double* __restrict__ cdf, const double* __restrict__ total_p, const double* __restrict__ uniforms,
                                                              int32_t* __restrict__ source, int32_t* __restrict__ counts) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    bool dead;
    relocate_weight(raw_o[i], reinterpret_cast<const float4*>(raw_q)[i], min_opacity, dead);
    const double total = *total_p;
    int32_t src = -1;
    if (dead && total > 0.0) {
        const double target = uniforms[i] * total;
        uint32_t lo = 0, hi = N - 1;             // invariant: the answer is in [lo, hi]
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (cdf[mid] > target) hi = mid; else lo = mid + 1;
        }
        // rounding at the very end of the CDF could land on a trailing dead element: step back to the last alive one
        while (lo > 0 && !(cdf[lo] > (lo ? cdf[lo - 1] : 0.0))) --lo;
        src = int32_t(lo);
        atomicAdd(counts + lo, 1);
    }
    source[i] = src;
}
// every drawn source j: relocated opacity / scale (the arithmetic of relocation_kernel) -> raw values in the temporaries
__global__ void __launch_bounds__(256) relocate_values_kernel(const uint32_t N, const float* __restrict__ raw_o, const float* __restrict__ raw_s, const int32_t* __restrict__ counts,
                                                              const float* __restrict__ binoms, const int32_t n_max, const float min_opacity,
                                                              float* __restrict__ new_raw_o, float* __restrict__ new_raw_s) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N || counts[j] <= 0) return;
    const int32_t n = min(counts[j] + 1, n_max);
    const float o = 1.f / (1.f + expf(-raw_o[j]));
    const float no = 1.f - powf(1.f - o, 1.f / float(n));
    float denom = 0.f;
    for (int32_t a = 1; a <= n; ++a) {
        float sgn = 1.f;
        for (int32_t k = 0; k <= a - 1; ++k) {
            denom += binoms[(a - 1) * n_max + k] * ((sgn / sqrtf(float(k + 1))) * powf(no, float(k + 1)));
            sgn = -sgn;
        }
    }
    const float coeff = o / denom;
    const float oc = fminf(fmaxf(no, min_opacity), 1.f - 1e-7f);   // mcmc.cpp:155
    new_raw_o[j] = logf(oc / (1.f - oc));                            // torch::logit
#pragma unroll
    for (int d = 0; d < 3; ++d) new_raw_s[3 * j + d] = logf(coeff * expf(raw_s[3 * j + d]));
}
struct RelocRows { lfs_param_rows r[6]; };
__global__ void __launch_bounds__(256) relocate_apply_kernel(const uint32_t N, const RelocRows rows, const int32_t* __restrict__ source, const int32_t* __restrict__ counts,
                                                             const float* __restrict__ new_raw_o, const float* __restrict__ new_raw_s) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int32_t src = source[i];
    if (src >= 0) { // dead: becomes a copy of its (updated) source; its own Adam moments stay (the reference does not touch them)
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const uint32_t w = rows.r[t].width;
            float* p = rows.r[t].param;
            for (uint32_t c = 0; c < w; ++c) {
                float v = p[size_t(src) * w + c];
                if (t == 3) v = new_raw_s[3 * src + c];
                if (t == 5) v = new_raw_o[src];
                p[size_t(i) * w + c] = v;
            }
        }
    } else if (counts[i] > 0) { // drawn source: relocated opacity / scale, Adam moments zeroed
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const uint32_t w = rows.r[t].width;
            if (t == 3) for (uint32_t c = 0; c < 3; ++c) rows.r[t].param[size_t(i) * 3 + c] = new_raw_s[3 * i + c];
            if (t == 5) rows.r[t].param[i] = new_raw_o[i];
            for (uint32_t c = 0; c < w; ++c) {
                if (rows.r[t].exp_avg) rows.r[t].exp_avg[size_t(i) * w + c] = 0.f;
                if (rows.r[t].exp_avg_sq) rows.r[t].exp_avg_sq[size_t(i) * w + c] = 0.f;
            }
        }
    }
}

struct RelocWs { double* block_sums; double* cdf; int32_t* source; int32_t* counts; float* new_raw_o; float* new_raw_s; size_t bytes; };
static RelocWs reloc_ws(void* base, uint32_t N) {
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    RelocWs w; char* p = (char*)base; size_t o = 0;
    const size_t nb = (size_t(N) + RL_BLOCK - 1) / RL_BLOCK;
    w.block_sums = (double*)(p + o); o += al(8 * (nb + 1));
    w.cdf = (double*)(p + o); o += al(8 * size_t(N));
    w.source = (int32_t*)(p + o); o += al(4 * size_t(N));
    w.counts = (int32_t*)(p + o); o += al(4 * size_t(N));
    w.new_raw_o = (float*)(p + o); o += al(4 * size_t(N));
    w.new_raw_s = (float*)(p + o); o += al(12 * size_t(N));
    w.bytes = o;
    return w;
}

} // namespace lfs

extern "C" size_t lfs_mcmc_relocate_workspace_bytes(uint32_t N) { return lfs::reloc_ws(nullptr, N).bytes; }

extern "C" int lfs_mcmc_relocate(
    uint32_t N, const lfs_param_rows* rows, const double* uniforms, const float* binoms, int32_t n_max, float min_opacity, int32_t* n_dead,
    void* workspace, size_t workspace_bytes, lfs_stream_t stream) {
    using namespace lfs;
    if (N == 0) return LFS_OK;
    if (!rows || !uniforms || !binoms || !workspace || n_max <= 0 || N > (1u << 24)) return LFS_E_INVALID;
    const uint32_t widths[6] = {3, 3, rows[2].width, 3, 4, 1};
    for (int t = 0; t < 6; ++t) if (rows[t].width != widths[t] || (rows[t].width && !rows[t].param)) return LFS_E_INVALID;
    const RelocWs w = reloc_ws(workspace, N);
    if (workspace_bytes < w.bytes) return LFS_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const float* raw_s = rows[3].param; const float* raw_q = rows[4].param; const float* raw_o = rows[5].param;
    const uint32_t nb = (N + RL_BLOCK - 1) / RL_BLOCK;
    if (n_dead) { hipError_t e = hipMemsetAsync(n_dead, 0, sizeof(int32_t), s); if (e != hipSuccess) return (int)e; }
    hipLaunchKernelGGL(relocate_block_sums_kernel, dim3(nb), dim3(256), 0, s, N, raw_o, raw_q, min_opacity, w.block_sums, w.counts, n_dead);
    hipLaunchKernelGGL(relocate_scan_blocks_kernel, dim3(1), dim3(256), 0, s, nb, w.block_sums);
    hipLaunchKernelGGL(relocate_cdf_kernel, dim3(nb), dim3(256), 0, s, N, raw_o, raw_q, min_opacity, w.block_sums, w.cdf);
    const dim3 g((N + 255) / 256);
    hipLaunchKernelGGL(relocate_sample_kernel, g, dim3(256), 0, s, N, raw_o, raw_q, min_opacity, w.cdf, w.block_sums + nb, uniforms, w.source, w.counts);
    hipLaunchKernelGGL(relocate_values_kernel, g, dim3(256), 0, s, N, raw_o, raw_s, w.counts, binoms, n_max, min_opacity, w.new_raw_o, w.new_raw_s);
    RelocRows rr;
    for (int t = 0; t < 6; ++t) rr.r[t] = rows[t];
    hipLaunchKernelGGL(relocate_apply_kernel, g, dim3(256), 0, s, N, rr, w.source, w.counts, w.new_raw_o, w.new_raw_s);
    return (int)hipGetLastError();
}

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
