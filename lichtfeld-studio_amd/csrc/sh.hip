// K2 / K9 — spherical harmonics colour and its vjp (replace
// gsplat::spherical_harmonics_fwd / _bwd; reference: gsplat/SphericalHarmonicsCUDA.cu
// :20-110 forward polynomial, :112-371 vjp, :373-481 kernels; host
// SphericalHarmonics.cpp:15-76).
//
// HBM-bound: 12*K B of coefficients per Gaussian dominate. The reference maps one
// thread to one (Gaussian, channel) so coefficient loads are 12-B strided across
// lanes; here LPG lanes share one Gaussian and lane k owns basis k, so a wave's
// coefficient load is one fully coalesced 64 x 12 B block, the K-sum is a DPP
// butterfly, and v_coeffs is stored with the same coalesced shape. Every lane
// evaluates the whole (cheap) basis polynomial and picks its own entry with a
// select tree — VALU is ~1/3 of the memory time.
#include "lfs_math.cuh"
#include "lfs_prof.h"
#include "../../include/lfs_gsplat.h"

namespace lfs {

// Sloan's "Efficient Spherical Harmonic Evaluation" (JCGT 2013) basis, degree <= 4.
// b[25]; when GRAD also the partials w.r.t. the unit direction.
template <bool GRAD>
LFS_DI void sh_basis(const int degree, const float x, const float y, const float z,
                     float* __restrict__ b, float* __restrict__ bx, float* __restrict__ by, float* __restrict__ bz) {
#pragma unroll
    for (int k = 0; k < 25; ++k) { b[k] = 0.f; if (GRAD) { bx[k] = 0.f; by[k] = 0.f; bz[k] = 0.f; } }
    b[0] = 0.2820947917738781f;
    if (degree < 1) return;
    const float c1 = 0.48860251190292f;
    b[1] = -c1 * y; b[2] = c1 * z; b[3] = -c1 * x;
    if (GRAD) { by[1] = -c1; bz[2] = c1; bx[3] = -c1; }
    if (degree < 2) return;
    const float z2 = z * z;
    const float t0B = -1.092548430592079f * z;
    const float fC1 = x * x - y * y, fS1 = 2.f * x * y;
    const float c2 = 0.5462742152960395f;
    b[4] = c2 * fS1; b[5] = t0B * y; b[6] = 0.9461746957575601f * z2 - 0.3153915652525201f;
    b[7] = t0B * x; b[8] = c2 * fC1;
    const float fC1_x = 2.f * x, fC1_y = -2.f * y, fS1_x = 2.f * y, fS1_y = 2.f * x;
    const float b6_z = 2.f * 0.9461746957575601f * z;
    if (GRAD) {
        bx[4] = c2 * fS1_x; by[4] = c2 * fS1_y;
        by[5] = t0B; bz[5] = -1.092548430592079f * y;
        bz[6] = b6_z;
        bx[7] = t0B; bz[7] = -1.092548430592079f * x;
        bx[8] = c2 * fC1_x; by[8] = c2 * fC1_y;
    }
    if (degree < 3) return;
    const float t0C = -2.285228997322329f * z2 + 0.4570457994644658f;
    const float t1B = 1.445305721320277f * z;
    const float fC2 = x * fC1 - y * fS1, fS2 = x * fS1 + y * fC1;
    const float c3 = -0.5900435899266435f;
    b[9] = c3 * fS2; b[10] = t1B * fS1; b[11] = t0C * y;
    b[12] = z * (1.865881662950577f * z2 - 1.119528997770346f);
    b[13] = t0C * x; b[14] = t1B * fC1; b[15] = c3 * fC2;
    const float fC2_x = fC1 + x * fC1_x - y * fS1_x, fC2_y = x * fC1_y - fS1 - y * fS1_y;
    const float fS2_x = fS1 + x * fS1_x + y * fC1_x, fS2_y = x * fS1_y + fC1 + y * fC1_y;
    const float b12_z = 3.f * 1.865881662950577f * z2 - 1.119528997770346f;
    if (GRAD) {
        const float t0C_z = -2.285228997322329f * 2.f * z, t1B_z = 1.445305721320277f;
        bx[9] = c3 * fS2_x; by[9] = c3 * fS2_y;
        bx[10] = t1B * fS1_x; by[10] = t1B * fS1_y; bz[10] = t1B_z * fS1;
        by[11] = t0C; bz[11] = t0C_z * y;
        bz[12] = b12_z;
        bx[13] = t0C; bz[13] = t0C_z * x;
        bx[14] = t1B * fC1_x; by[14] = t1B * fC1_y; bz[14] = t1B_z * fC1;
        bx[15] = c3 * fC2_x; by[15] = c3 * fC2_y;
    }
    if (degree < 4) return;
    const float t0D = z * (-4.683325804901025f * z2 + 2.007139630671868f);
    const float t1C = 3.31161143515146f * z2 - 0.47308734787878f;
    const float t2B = -1.770130769779931f * z;
    const float fC3 = x * fC2 - y * fS2, fS3 = x * fS2 + y * fC2;
    const float c4 = 0.6258357354491763f;
    b[16] = c4 * fS3; b[17] = t2B * fS2; b[18] = t1C * fS1; b[19] = t0D * y;
    b[20] = 1.984313483298443f * z * b[12] - 1.006230589874905f * b[6];
    b[21] = t0D * x; b[22] = t1C * fC1; b[23] = t2B * fC2; b[24] = c4 * fC3;
    if (GRAD) {
        const float t0D_z = 3.f * -4.683325804901025f * z2 + 2.007139630671868f;
        const float t1C_z = 2.f * 3.31161143515146f * z, t2B_z = -1.770130769779931f;
        const float fC3_x = fC2 + x * fC2_x - y * fS2_x, fC3_y = x * fC2_y - fS2 - y * fS2_y;
        const float fS3_x = fS2 + y * fC2_x + x * fS2_x, fS3_y = x * fS2_y + fC2 + y * fC2_y;
        bx[16] = c4 * fS3_x; by[16] = c4 * fS3_y;
        bx[17] = t2B * fS2_x; by[17] = t2B * fS2_y; bz[17] = t2B_z * fS2;
        bx[18] = t1C * fS1_x; by[18] = t1C * fS1_y; bz[18] = t1C_z * fS1;
        by[19] = t0D; bz[19] = t0D_z * y;
        bz[20] = 1.984313483298443f * (b[12] + z * b12_z) - 1.006230589874905f * b6_z;
        bx[21] = t0D; bz[21] = t0D_z * x;
        bx[22] = t1C * fC1_x; by[22] = t1C * fC1_y; bz[22] = t1C_z * fC1;
        bx[23] = t2B * fC2_x; by[23] = t2B * fC2_y; bz[23] = t2B_z * fC2;
        bx[24] = c4 * fC3_x; by[24] = c4 * fC3_y;
    }
}

// pick a[k] for a per-lane k without dynamic register indexing (select tree)
template <int LPG>
LFS_DI float pick(const float* a, const int k) {
    if (LPG == 1) return a[0];
    float t[32];
#pragma unroll
    for (int i = 0; i < LPG; ++i) t[i] = (i < 25) ? a[i] : 0.f;
#pragma unroll
    for (int w = LPG / 2, bit = 0; w >= 1; w >>= 1, ++bit) {
        const bool hi = (k >> bit) & 1;
#pragma unroll
        for (int i = 0; i < w; ++i) t[i] = hi ? t[2 * i + 1] : t[2 * i];
    }
    return t[0];
}

template <int LPG>
LFS_DI float group_sum(float v) {
#pragma unroll
    for (int m = LPG / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

template <int LPG>
__global__ void __launch_bounds__(256) sh_fwd_kernel(
    const uint32_t n, const uint32_t K, const int degree,
    const float* __restrict__ dirs, const float* __restrict__ coeffs, const uint8_t* __restrict__ masks,
    float* __restrict__ colors) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t g = tid / LPG;
    const int k = tid % LPG;
    if (g >= n) return;
    const int Kd = (degree + 1) * (degree + 1);
    const bool on = masks == nullptr || masks[g] != 0;
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    if (on) { // uniform per LPG group
        float x = dirs[3 * g], y = dirs[3 * g + 1], z = dirs[3 * g + 2];
        if (degree >= 1) { const float inorm = 1.f / sqrtf(x * x + y * y + z * z); x *= inorm; y *= inorm; z *= inorm; }
        float b[25];
        sh_basis<false>(degree, x, y, z, b, nullptr, nullptr, nullptr);
        const float bk = pick<LPG>(b, k);
        if (k < Kd) {
            const float* cf = coeffs + (size_t(g) * K + k) * 3;
            r0 = bk * cf[0]; r1 = bk * cf[1]; r2 = bk * cf[2];
        }
    }
    r0 = group_sum<LPG>(r0); r1 = group_sum<LPG>(r1); r2 = group_sum<LPG>(r2);
    if (k == 0) { colors[3 * g] = r0; colors[3 * g + 1] = r1; colors[3 * g + 2] = r2; }
}

template <int LPG>
__global__ void __launch_bounds__(256) sh_bwd_kernel(
    const uint32_t n, const uint32_t K, const int degree,
    const float* __restrict__ dirs, const float* __restrict__ coeffs, const uint8_t* __restrict__ masks,
    const float* __restrict__ v_colors, float* __restrict__ v_coeffs, float* __restrict__ v_dirs) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t g = tid / LPG;
    const int k = tid % LPG;
    if (g >= n) return;
    const int Kd = (degree + 1) * (degree + 1);
    const bool on = masks == nullptr || masks[g] != 0;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;   // v_coeffs[g][k][:]
    float gx = 0.f, gy = 0.f, gz = 0.f;   // lane-k share of dL/d(unit dir)
    float x = 0.f, y = 0.f, z = 0.f, inorm = 1.f;
    if (on) {
        x = dirs[3 * g]; y = dirs[3 * g + 1]; z = dirs[3 * g + 2];
        if (degree >= 1) { inorm = 1.f / sqrtf(x * x + y * y + z * z); x *= inorm; y *= inorm; z *= inorm; }
        const float v0 = v_colors[3 * g], v1 = v_colors[3 * g + 1], v2 = v_colors[3 * g + 2];
        float b[25], bx[25], by[25], bz[25];
        if (v_dirs != nullptr) sh_basis<true>(degree, x, y, z, b, bx, by, bz);
        else sh_basis<false>(degree, x, y, z, b, nullptr, nullptr, nullptr);
        const float bk = pick<LPG>(b, k);
        if (k < Kd) {
            o0 = bk * v0; o1 = bk * v1; o2 = bk * v2;
            if (v_dirs != nullptr && degree >= 1) {
                const float* cf = coeffs + (size_t(g) * K + k) * 3;
                const float s = cf[0] * v0 + cf[1] * v1 + cf[2] * v2;
                gx = pick<LPG>(bx, k) * s; gy = pick<LPG>(by, k) * s; gz = pick<LPG>(bz, k) * s;
            }
        }
    }
    if (uint32_t(k) < K) {
        float* vc = v_coeffs + (size_t(g) * K + k) * 3;
        vc[0] = o0; vc[1] = o1; vc[2] = o2;
    }
    if (v_dirs != nullptr) {
        gx = group_sum<LPG>(gx); gy = group_sum<LPG>(gy); gz = group_sum<LPG>(gz);
        if (k == 0) {
            // tangent-plane projection + chain through the normalisation
            const float d = gx * x + gy * y + gz * z;
            v_dirs[3 * g] = (gx - d * x) * inorm; v_dirs[3 * g + 1] = (gy - d * y) * inorm; v_dirs[3 * g + 2] = (gz - d * z) * inorm;
        }
    }
}

// ---------------------------------------------------------------------------
// Fused L2 variants (extensions, not in Ops.h): what rasterizer.cpp:256-263 builds around the op with
// libtorch - dirs = means - campos, masks = all(radii > 0), coeffs = cat(sh0, shN),
// colors = clamp_min(sh + 0.5, 0) - folded into the kernel, so none of those tensors is materialised
// (the cat alone is a 192 MB copy per step at 1M Gaussians, and again in the backward).
// ---------------------------------------------------------------------------
LFS_DI f3 campos_of(const float* __restrict__ vm) { // -R^T t of a rigid row-major [4,4] world->camera matrix
    return {-(vm[0] * vm[3] + vm[4] * vm[7] + vm[8] * vm[11]),
            -(vm[1] * vm[3] + vm[5] * vm[7] + vm[9] * vm[11]),
            -(vm[2] * vm[3] + vm[6] * vm[7] + vm[10] * vm[11])};
}

template <int LPG>
__global__ void __launch_bounds__(256) sh_model_fwd_kernel(
    const uint32_t n, const uint32_t K, const int degree,
    const float* __restrict__ means, const float* __restrict__ viewmat, const float* __restrict__ sh0, const float* __restrict__ shN,
    const int32_t* __restrict__ radii, float* __restrict__ colors) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t g = tid / LPG;
    const int k = tid % LPG;
    if (g >= n) return;
    const int Kd = (degree + 1) * (degree + 1);
    const bool on = radii[2 * g] > 0 && radii[2 * g + 1] > 0;
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    if (on) {
        const f3 cp = campos_of(viewmat);
        float x = means[3 * g] - cp.x, y = means[3 * g + 1] - cp.y, z = means[3 * g + 2] - cp.z;
        if (degree >= 1) { const float inorm = 1.f / sqrtf(x * x + y * y + z * z); x *= inorm; y *= inorm; z *= inorm; }
        float b[25];
        sh_basis<false>(degree, x, y, z, b, nullptr, nullptr, nullptr);
        const float bk = pick<LPG>(b, k);
        if (k < Kd) {
            const float* cf = (k == 0) ? sh0 + size_t(g) * 3 : shN + (size_t(g) * (K - 1) + (k - 1)) * 3;
            r0 = bk * cf[0]; r1 = bk * cf[1]; r2 = bk * cf[2];
        }
    }
    r0 = group_sum<LPG>(r0); r1 = group_sum<LPG>(r1); r2 = group_sum<LPG>(r2);
    if (k == 0) { colors[3 * g] = fmaxf(r0 + 0.5f, 0.f); colors[3 * g + 1] = fmaxf(r1 + 0.5f, 0.f); colors[3 * g + 2] = fmaxf(r2 + 0.5f, 0.f); }
}

// v_colors = dL/d(clamped colors); the clamp passes where the stored colour is > 0. v_sh0 / v_shN are written
// (ACCUM = false) or added to (ACCUM = true: second and later views of a step); v_means += dL/d(dirs).
template <int LPG, bool ACCUM>
__global__ void __launch_bounds__(256) sh_model_bwd_kernel(
    const uint32_t n, const uint32_t K, const int degree,
    const float* __restrict__ means, const float* __restrict__ viewmat, const float* __restrict__ sh0, const float* __restrict__ shN,
    const int32_t* __restrict__ radii, const float* __restrict__ colors, const float* __restrict__ v_colors,
    float* __restrict__ v_sh0, float* __restrict__ v_shN, float* __restrict__ v_means) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t g = tid / LPG;
    const int k = tid % LPG;
    if (g >= n) return;
    const int Kd = (degree + 1) * (degree + 1);
    const bool on = radii[2 * g] > 0 && radii[2 * g + 1] > 0;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
    float x = 0.f, y = 0.f, z = 0.f, inorm = 1.f;
    if (on) {
        const f3 cp = campos_of(viewmat);
        x = means[3 * g] - cp.x; y = means[3 * g + 1] - cp.y; z = means[3 * g + 2] - cp.z;
        if (degree >= 1) { inorm = 1.f / sqrtf(x * x + y * y + z * z); x *= inorm; y *= inorm; z *= inorm; }
        const float v0 = colors[3 * g] > 0.f ? v_colors[3 * g] : 0.f;
        const float v1 = colors[3 * g + 1] > 0.f ? v_colors[3 * g + 1] : 0.f;
        const float v2 = colors[3 * g + 2] > 0.f ? v_colors[3 * g + 2] : 0.f;
        float b[25], bx[25], by[25], bz[25];
        sh_basis<true>(degree, x, y, z, b, bx, by, bz);
        const float bk = pick<LPG>(b, k);
        if (k < Kd) {
            o0 = bk * v0; o1 = bk * v1; o2 = bk * v2;
            if (degree >= 1) {
                const float* cf = (k == 0) ? sh0 + size_t(g) * 3 : shN + (size_t(g) * (K - 1) + (k - 1)) * 3;
                const float s = cf[0] * v0 + cf[1] * v1 + cf[2] * v2;
                gx = pick<LPG>(bx, k) * s; gy = pick<LPG>(by, k) * s; gz = pick<LPG>(bz, k) * s;
            }
        }
    }
    if (uint32_t(k) < K) {
        float* vc = (k == 0) ? v_sh0 + size_t(g) * 3 : v_shN + (size_t(g) * (K - 1) + (k - 1)) * 3;
        if (ACCUM) { if (on) { vc[0] += o0; vc[1] += o1; vc[2] += o2; } }
        else { vc[0] = o0; vc[1] = o1; vc[2] = o2; }
    }
    gx = group_sum<LPG>(gx); gy = group_sum<LPG>(gy); gz = group_sum<LPG>(gz);
    if (k == 0 && on && degree >= 1) {
        const float d = gx * x + gy * y + gz * z;
        v_means[3 * g] += (gx - d * x) * inorm; v_means[3 * g + 1] += (gy - d * y) * inorm; v_means[3 * g + 2] += (gz - d * z) * inorm;
    }
}

static inline int lanes_for(uint32_t k) { return k <= 1 ? 1 : k <= 4 ? 4 : k <= 16 ? 16 : 32; }

} // namespace lfs

extern "C" int lfs_spherical_harmonics_fwd(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* dirs, const float* coeffs, const uint8_t* masks,
    float* colors, lfs_stream_t stream) {
    const uint32_t Kd = (degrees_to_use + 1) * (degrees_to_use + 1);
    if (degrees_to_use > 4 || Kd > K) return LFS_E_INVALID;
    if (n == 0) return LFS_OK;
    if (!dirs || !coeffs || !colors) return LFS_E_INVALID;
    const int lpg = lfs::lanes_for(Kd);
    const uint64_t threads = uint64_t(n) * lpg;
    dim3 grid((threads + 255) / 256), block(256);
    hipStream_t s = (hipStream_t)stream;
    const int deg = int(degrees_to_use);
    lfs::ProfScope prof("sh_fwd", s);
    switch (lpg) {
    case 1: hipLaunchKernelGGL(lfs::sh_fwd_kernel<1>, grid, block, 0, s, n, K, deg, dirs, coeffs, masks, colors); break;
    case 4: hipLaunchKernelGGL(lfs::sh_fwd_kernel<4>, grid, block, 0, s, n, K, deg, dirs, coeffs, masks, colors); break;
    case 16: hipLaunchKernelGGL(lfs::sh_fwd_kernel<16>, grid, block, 0, s, n, K, deg, dirs, coeffs, masks, colors); break;
    default: hipLaunchKernelGGL(lfs::sh_fwd_kernel<32>, grid, block, 0, s, n, K, deg, dirs, coeffs, masks, colors); break;
    }
    return (int)hipGetLastError();
}

extern "C" int lfs_spherical_harmonics_bwd(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* dirs, const float* coeffs, const uint8_t* masks,
    const float* v_colors, float* v_coeffs, float* v_dirs, lfs_stream_t stream) {
    const uint32_t Kd = (degrees_to_use + 1) * (degrees_to_use + 1);
    if (degrees_to_use > 4 || Kd > K || K > 32) return LFS_E_INVALID;
    if (n == 0) return LFS_OK;
    if (!dirs || !coeffs || !v_colors || !v_coeffs) return LFS_E_INVALID;
    const int lpg = lfs::lanes_for(K); // every one of the K rows of v_coeffs is written
    const uint64_t threads = uint64_t(n) * lpg;
    dim3 grid((threads + 255) / 256), block(256);
    hipStream_t s = (hipStream_t)stream;
    const int deg = int(degrees_to_use);
    lfs::ProfScope prof("sh_bwd", s);
    switch (lpg) {
    case 1: hipLaunchKernelGGL(lfs::sh_bwd_kernel<1>, grid, block, 0, s, n, K, deg, dirs, coeffs, masks, v_colors, v_coeffs, v_dirs); break;
    case 4: hipLaunchKernelGGL(lfs::sh_bwd_kernel<4>, grid, block, 0, s, n, K, deg, dirs, coeffs, masks, v_colors, v_coeffs, v_dirs); break;
    case 16: hipLaunchKernelGGL(lfs::sh_bwd_kernel<16>, grid, block, 0, s, n, K, deg, dirs, coeffs, masks, v_colors, v_coeffs, v_dirs); break;
    default: hipLaunchKernelGGL(lfs::sh_bwd_kernel<32>, grid, block, 0, s, n, K, deg, dirs, coeffs, masks, v_colors, v_coeffs, v_dirs); break;
    }
    return (int)hipGetLastError();
}

extern "C" int lfs_sh_model_fwd(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* means, const float* viewmat, const float* sh0, const float* shN,
    const int32_t* radii, float* colors, lfs_stream_t stream) {
    const uint32_t Kd = (degrees_to_use + 1) * (degrees_to_use + 1);
    if (degrees_to_use > 4 || Kd > K || K > 32) return LFS_E_INVALID;
    if (n == 0) return LFS_OK;
    if (!means || !viewmat || !sh0 || (K > 1 && !shN) || !radii || !colors) return LFS_E_INVALID;
    const int lpg = lfs::lanes_for(Kd);
    dim3 grid((uint64_t(n) * lpg + 255) / 256), block(256);
    hipStream_t s = (hipStream_t)stream;
    const int deg = int(degrees_to_use);
    lfs::ProfScope prof("sh_fwd", s);
    switch (lpg) {
    case 1: hipLaunchKernelGGL(lfs::sh_model_fwd_kernel<1>, grid, block, 0, s, n, K, deg, means, viewmat, sh0, shN, radii, colors); break;
    case 4: hipLaunchKernelGGL(lfs::sh_model_fwd_kernel<4>, grid, block, 0, s, n, K, deg, means, viewmat, sh0, shN, radii, colors); break;
    case 16: hipLaunchKernelGGL(lfs::sh_model_fwd_kernel<16>, grid, block, 0, s, n, K, deg, means, viewmat, sh0, shN, radii, colors); break;
    default: hipLaunchKernelGGL(lfs::sh_model_fwd_kernel<32>, grid, block, 0, s, n, K, deg, means, viewmat, sh0, shN, radii, colors); break;
    }
    return (int)hipGetLastError();
}

extern "C" int lfs_sh_model_bwd(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* means, const float* viewmat, const float* sh0, const float* shN,
    const int32_t* radii, const float* colors, const float* v_colors, int accumulate,
    float* v_sh0, float* v_shN, float* v_means, lfs_stream_t stream) {
    const uint32_t Kd = (degrees_to_use + 1) * (degrees_to_use + 1);
    if (degrees_to_use > 4 || Kd > K || K > 32) return LFS_E_INVALID;
    if (n == 0) return LFS_OK;
    if (!means || !viewmat || !sh0 || (K > 1 && (!shN || !v_shN)) || !radii || !colors || !v_colors || !v_sh0 || !v_means) return LFS_E_INVALID;
    const int lpg = lfs::lanes_for(K);
    dim3 grid((uint64_t(n) * lpg + 255) / 256), block(256);
    hipStream_t s = (hipStream_t)stream;
    const int deg = int(degrees_to_use);
    lfs::ProfScope prof("sh_bwd", s);
#define LFS_SHB(L)                                                                                                                       \
    if (accumulate) hipLaunchKernelGGL((lfs::sh_model_bwd_kernel<L, true>), grid, block, 0, s, n, K, deg, means, viewmat, sh0, shN, radii, \
                                       colors, v_colors, v_sh0, v_shN, v_means);                                                          \
    else hipLaunchKernelGGL((lfs::sh_model_bwd_kernel<L, false>), grid, block, 0, s, n, K, deg, means, viewmat, sh0, shN, radii, colors,    \
                            v_colors, v_sh0, v_shN, v_means)
    switch (lpg) {
    case 1: LFS_SHB(1); break;
    case 4: LFS_SHB(4); break;
    case 16: LFS_SHB(16); break;
    default: LFS_SHB(32); break;
    }
#undef LFS_SHB
    return (int)hipGetLastError();
}
