// The per-(camera, Gaussian) staging of the world-space rasterizer, shared by raster_pack_kernel (raster.hip: the Ops.h entry points) and by the
// training step, whose projection kernel writes the finished records itself (projection_ut.hip, PACK): the 64-byte record the
// fwd / bwd kernels walk and the 32-byte culling record (silhouette conic of the alpha >= 1/255 ellipsoid, lfs_cull_conic.cuh).
#pragma once
#include "lfs_camera.cuh"
#include "lfs_raster_common.cuh"
#include "lfs_cull_conic.cuh"

namespace lfs {

// Per-Gaussian culling record (the silhouette conic of the alpha >= 1/255 ellipsoid in normalised camera coordinates)
struct __attribute__((aligned(16))) CullRec { float4 a, b; };

// UNIFORM_ORIGIN (global shutter): the record matrix is S^-1 R^T Rinv (camera-space ray directions -> Gaussian frame) and g = S^-1 R^T (o - mu) - since round 6 both in
// the rotated frame of LFS_REC_ROT (lfs_raster_common.cuh): U S^-1 R^T Rinv, |g|^2, |g|; rolling shutters keep S^-1 R^T and g = mu (per-pixel origins) and are never culled.
template <bool UNIFORM_ORIGIN>
LFS_DI void pack_gaussian(const CamDev& cam, const f3 mu, const float4 q, const float sc[3], const float opac, const float c0, const float c1, const float c2,
                          GaussRec& rec, CullRec& cr) {
    // No contraction anywhere in this function (and in conic_record): the records are built by raster_pack_kernel (raster.hip, compiled with contraction) AND by the
    // training step's projection kernel (projection_ut.hip, compiled without) - the two must produce the same bits (the C++ step is held bit for bit to the
    // op-by-op path, tests/test_gpu_gut_step.py). The helpers of lfs_math.cuh take the including file's default, so their bodies are repeated here.
#pragma clang fp contract(off)
    m3 R;
    {   // quat_to_rotmat (lfs_math.cuh; Utils.cuh:80-102)
        float w = q.x, x = q.y, y = q.z, z = q.w;
        const float inv = 1.f / sqrtf(x * x + y * y + z * z + w * w);
        x *= inv; y *= inv; z *= inv; w *= inv;
        const float x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
        R.m[0][0] = 1.f - 2.f * (y2 + z2); R.m[1][0] = 2.f * (xy + wz); R.m[2][0] = 2.f * (xz - wy);
        R.m[0][1] = 2.f * (xy - wz); R.m[1][1] = 1.f - 2.f * (x2 + z2); R.m[2][1] = 2.f * (yz + wx);
        R.m[0][2] = 2.f * (xz + wy); R.m[1][2] = 2.f * (yz - wx); R.m[2][2] = 1.f - 2.f * (x2 + y2);
    }
    const float is[3] = {1.f / sc[0], 1.f / sc[1], 1.f / sc[2]};
    m3 M;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) M.m[r][c] = is[r] * R.m[c][r];
    f3 g = mu, g_unscaled = mu;
    m3 Mr = M;
    if (UNIFORM_ORIGIN) {
        const float omx = cam.origin.x - mu.x, omy = cam.origin.y - mu.y, omz = cam.origin.z - mu.z;
        g = {M.m[0][0] * omx + M.m[0][1] * omy + M.m[0][2] * omz, M.m[1][0] * omx + M.m[1][1] * omy + M.m[1][2] * omz, M.m[2][0] * omx + M.m[2][1] * omy + M.m[2][2] * omz};
        g_unscaled = g;
        const m3& Ri = cam.Rinv; // camera -> world: the kernels then work on CAMERA-space ray directions
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) Mr.m[r][c] = M.m[r][0] * Ri.m[0][c] + M.m[r][1] * Ri.m[1][c] + M.m[r][2] * Ri.m[2][c];
    }
#if LFS_REC_LOG2
    // The record in the form the per-pixel evaluation wants (lfs_raster_common.cuh, REC_SCALE): matrix (and, global shutter, g) times sqrt(0.5 log2 e), and
    // log2(opacity) - alpha = exp2(log2(opac) - |w'|^2) is then three fused multiply-adds and one v_exp_f32 (opacity 0: -inf -> alpha 0, never composited)
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Mr.m[r][c] *= REC_SCALE;
    if (UNIFORM_ORIGIN) g = {g.x * REC_SCALE, g.y * REC_SCALE, g.z * REC_SCALE};
    const float opac_field = __builtin_amdgcn_logf(opac);   // v_log_f32 = log2
#else
    const float opac_field = opac;
#endif
#if LFS_REC_ROT
    if (UNIFORM_ORIGIN) { // the record in the frame in which g lies on the third axis (lfs_raster_common.cuh, LFS_REC_ROT): rows of U M' (one rounding, from double), G^2, G
        RotFrame F;
        rot_frame(g_unscaled, F);
        m3 Mu;
#pragma unroll
        for (int c = 0; c < 3; ++c) rot_apply(F, Mr.m[0][c], Mr.m[1][c], Mr.m[2][c], Mu.m[0][c], Mu.m[1][c], Mu.m[2][c]);
        Mr = Mu;
        const float G = float(F.len * double(REC_SCALE));
        g = {G * G, 0.f, G};
    }
#endif
#if LFS_REC_PKQ
    if (UNIFORM_ORIGIN) { // rows 0 and 1 of U M' interleaved by column: (q.x, q.y) is then a chain of three packed operations on aligned SGPR pairs (raster.hip, ray_eval)
        rec.r0 = make_float4(Mr.m[0][0], Mr.m[1][0], Mr.m[0][1], Mr.m[1][1]);
        rec.r1 = make_float4(Mr.m[0][2], Mr.m[1][2], g.x, g.y);
    } else
#endif
    {
    rec.r0 = make_float4(Mr.m[0][0], Mr.m[0][1], Mr.m[0][2], g.x);
    rec.r1 = make_float4(Mr.m[1][0], Mr.m[1][1], Mr.m[1][2], g.y);
    }
    rec.r2 = make_float4(Mr.m[2][0], Mr.m[2][1], Mr.m[2][2], g.z);
    rec.r3 = make_float4(opac_field, c0, c1, c2);
    ConicRec k = conic_never();
    if (UNIFORM_ORIGIN) {
        const m3& Ri = cam.Rinv;                      // camera -> world, so world -> camera is its transpose
        const float dx = mu.x - cam.origin.x, dy = mu.y - cam.origin.y, dz = mu.z - cam.origin.z;
        const float pc[3] = {Ri.m[0][0] * dx + Ri.m[1][0] * dy + Ri.m[2][0] * dz, Ri.m[0][1] * dx + Ri.m[1][1] * dy + Ri.m[2][1] * dz,
                             Ri.m[0][2] * dx + Ri.m[1][2] * dy + Ri.m[2][2] * dz};
        float A[3][3];                                // A = Rc R S  (Sigma_cam = A A^T)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                A[r][c] = (Ri.m[0][r] * R.m[0][c] + Ri.m[1][r] * R.m[1][c] + Ri.m[2][r] * R.m[2][c]) * sc[c];
        k = conic_record(pc, A, opac);
    }
    cr.a = make_float4(k.px, k.py, k.a, k.b);
    cr.b = make_float4(k.d, k.e, k.g, k.ia);
}

} // namespace lfs
