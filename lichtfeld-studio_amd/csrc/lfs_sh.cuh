// Device-side spherical-harmonics building blocks shared by sh.hip (K2 / K9) and by the fused tail of the training step (raster.hip: gut_tail_kernel): the basis
// polynomial, the lane-group sum and the camera centre. Every function pins un-fused IEEE arithmetic in its own body (#pragma clang fp contract(off)), so the
// including translation unit's -ffp-contract setting cannot change a bit (sh.hip is compiled with contraction off, raster.hip with it on).
#pragma once
#include "lfs_math.cuh"

namespace lfs {

// Sloan's "Efficient Spherical Harmonic Evaluation" (JCGT 2013) basis, degree <= 4.
// b[25]; when GRAD also the partials w.r.t. the unit direction.
template <bool GRAD, int MAXDEG = 4>   // MAXDEG: what the caller's lane layout can hold (16 lanes per Gaussian = degree <= 3): higher blocks are compiled out
LFS_DI void sh_basis(const int degree, const float x, const float y, const float z,
                     float* __restrict__ b, float* __restrict__ bx, float* __restrict__ by, float* __restrict__ bz) {
#pragma clang fp contract(off)
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
    if (MAXDEG < 3 || degree < 3) return;
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
    if (MAXDEG < 4 || degree < 4) return;
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

// Sum over the LPG lanes of a group, result in every lane. Default: xor butterfly through ds_bpermute (LDS pipe). LFS_SH_DPP_SUM (compile-time,
// off until it has run on a GPU - DESIGN.md §6b): the same additions as DPP operands - row_ror:8 is lane^8; after that step the values are
// symmetric under ^8, so row_ror:4 delivers the lane^4 partner (in either rotation direction); quad_perm for lane^2 and lane^1. Bit-identical
// sums, no LDS traffic. LPG is 1, 4, 16 or 32 (lanes_for): only the lane^16 step of LPG = 32 crosses a DPP row.
#ifdef LFS_SH_DPP_SUM
template <int CTRL> LFS_DI float sh_dpp(float v) { return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false)); }
template <int LPG>
LFS_DI float group_sum(float v) {
#pragma clang fp contract(off)
    if (LPG >= 32) v += __shfl_xor(v, 16, 64);
    if (LPG >= 16) { v += sh_dpp<0x128>(v); v += sh_dpp<0x124>(v); }
    if (LPG >= 4) { v += sh_dpp<0x4E>(v); v += sh_dpp<0xB1>(v); }
    return v;
}
#else
template <int LPG>
LFS_DI float group_sum(float v) {
#pragma clang fp contract(off)
#pragma unroll
    for (int m = LPG / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
#endif

LFS_DI f3 campos_of(const float* __restrict__ vm) { // -R^T t of a rigid row-major [4,4] world->camera matrix
#pragma clang fp contract(off)
    return {-(vm[0] * vm[3] + vm[4] * vm[7] + vm[8] * vm[11]),
            -(vm[1] * vm[3] + vm[5] * vm[7] + vm[9] * vm[11]),
            -(vm[2] * vm[3] + vm[6] * vm[7] + vm[10] * vm[11])};
}

} // namespace lfs
