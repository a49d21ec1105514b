// K1 — 3DGUT projection (replaces gsplat::projection_ut_3dgs_fused;
// reference: gsplat/ProjectionUT3DGSFused.cu:17-203, Projection.cpp:22-110).
//
// HBM-bound streaming kernel: 44 B read + 8 B radii write per Gaussian, + 24 B
// for survivors. One lane per (camera, Gaussian); blockIdx.y is
// the camera so every camera-derived quantity is wave-uniform. The file is
// compiled with -ffp-contract=off: it has VALU headroom to spare and un-fused
// arithmetic keeps the sigma-point cancellation (weights -99 / +16.67) on the
// same rounding path as the oracle.
#include "lfs_camera.cuh"
#include "lfs_prof.h"
#include "lfs_raster_pack.cuh"
#include "lfs_step_internal.h"

namespace lfs {

// ACT (fused training step): quats / scales / opacities are the RAW parameters; the kernel applies gs::SplatData's activations first (the arithmetic of
// activations_fwd_kernel, l2_fused.hip: x / max(|x|, 1e-12), exp, sigmoid), stores the activated values (act_*) and projects those - bit for bit what the
// two separate kernels compute. conics are not needed by the world-space rasterizer and not written then.
// SIMPLE (round 3): the camera block is KNOWN to be an undistorted pinhole with a global shutter - what the trainer renders (rasterizer.cpp:176-181 with the
// COLMAP PINHOLE / SIMPLE_PINHOLE models and every synthetic scene). The camera-model, distortion and shutter branches of lfs_camera.cuh - all wave-uniform,
// but all compiled into the generic kernel: 125 VGPRs and ~70 000 lines of ISA for the worst case, a fisheye camera under a rolling shutter - become
// compile-time constants and fold away; the arithmetic on the path that remains is untouched (same functions, no contraction: bit-identical outputs,
// tests/test_gpu_projection_sh.py).
// PACK (round 4, the training step): the kernel also writes what raster_pack_kernel (raster.hip) would compute from the activated values in a second pass over the
// Gaussians - the rasterizer's 64-byte record (colour slots zero: the SH colour kernel writes them into the record afterwards) and the 32-byte culling record - for
// every Gaussian that has a footprint. Same function (pack_gaussian, contraction-free), same operands: same bits; saves re-reading 44 B per Gaussian and a launch.
// pack_colors [N,3]: the SH colours (the step evaluates them BEFORE this kernel, for every Gaussian - a record is then one full 64-byte store; a first version had
// the SH kernel write its three floats into the finished records afterwards: partial-sector writes, sh_fwd 0.064 -> 0.083 ms).
template <bool ACT, bool SIMPLE = false, bool PACK = false>
__global__ void __launch_bounds__(256) projection_ut_kernel(
    const uint32_t N,
    const float* __restrict__ means, const float* __restrict__ quats, const float* __restrict__ scales,
    const float* __restrict__ opacities, const lfs_cameras cams,
    const float eps2d, const float near_plane, const float far_plane, const float radius_clip,
    const lfs_ut_params ut,
    int32_t* __restrict__ radii, float* __restrict__ means2d, float* __restrict__ depths,
    float* __restrict__ conics, float* __restrict__ compensations,
    float* __restrict__ act_quats = nullptr, float* __restrict__ act_scales = nullptr, float* __restrict__ act_opacities = nullptr,
    uint32_t* __restrict__ zero_words = nullptr, const uint32_t zero_n = 0, CamDev* __restrict__ cams_out = nullptr,
    GaussRec* __restrict__ recs = nullptr, CullRec* __restrict__ cull = nullptr, const float* __restrict__ pack_colors = nullptr) {
    const uint32_t cid = blockIdx.y;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (zero_words != nullptr && cid == 0) // rider of the training step: clear the intersection stage's per-tile totals (saves its memset launch)
        for (uint32_t i = gid; i < zero_n; i += gridDim.x * blockDim.x) zero_words[i] = 0u;
    if (gid >= N) return;
    const size_t idx = size_t(cid) * N + gid;

    CamDev cam;
    lfs_cameras cs = cams;
    if (SIMPLE) {
        cs.camera_model = LFS_CAMERA_PINHOLE; cs.rs_type = LFS_SHUTTER_GLOBAL; cs.viewmats1 = nullptr;
        cs.radial_coeffs = nullptr; cs.tangential_coeffs = nullptr; cs.thin_prism_coeffs = nullptr; cs.n_radial = 0; cs.n_thin_prism = 0;
    }
    cam_init(cam, cs, cid);
    if (cams_out != nullptr && gid == 0) cams_out[cid] = cam; // rider of the training step: the rasterizer's camera state (what cam_prep_kernel computes)

    int32_t out_rx = 0, out_ry = 0;
    float o_m2x = 0.f, o_m2y = 0.f, o_depth = 0.f, o_c0 = 0.f, o_c1 = 0.f, o_c2 = 0.f, o_comp = 0.f;
    f3 p_mean{0.f, 0.f, 0.f}; float4 p_q = make_float4(1.f, 0.f, 0.f, 0.f); float p_sc[3] = {1.f, 1.f, 1.f}, p_op = 0.f; // PACK: the activated operands

    do {
        const f3 mean = ld3(means, gid);
        f3 scale = ld3(scales, gid);
        float4 qin = reinterpret_cast<const float4*>(quats)[gid];
        float opac_in = opacities != nullptr ? opacities[gid] : 0.f;
        if (ACT) {
            const float den = fmaxf(sqrtf(qin.x * qin.x + qin.y * qin.y + qin.z * qin.z + qin.w * qin.w), 1e-12f);
            qin = make_float4(qin.x / den, qin.y / den, qin.z / den, qin.w / den);
            scale = {expf(scale.x), expf(scale.y), expf(scale.z)};
            opac_in = 1.f / (1.f + expf(-opac_in));
            if (cid == 0) {
                reinterpret_cast<float4*>(act_quats)[gid] = qin;
                st3(act_scales, gid, scale);
                act_opacities[gid] = opac_in;
            }
        }
        if (PACK) { p_mean = mean; p_q = qin; p_sc[0] = scale.x; p_sc[1] = scale.y; p_sc[2] = scale.z; p_op = opac_in; }
        const quat rot = qnormalize(quat{qin.x, qin.y, qin.z, qin.w});

        // depth test at the centre-of-exposure pose
        quat qc; f3 tc;
        cam_pose_at(cam, 0.5f, qc, tc);
        const f3 mean_c = qrotate(qc, mean) + tc;
        if (mean_c.z < near_plane || mean_c.z > far_plane) break;

        // sigma points (D = 3)
        const float D = 3.f;
        const float lambda = ut.alpha * ut.alpha * (D + ut.kappa) - D;
        const m3 R = qmat3(rot);
        const float sc[3] = {scale.x, scale.y, scale.z};
        const float w0m = lambda / (D + lambda);
        const float w0c = lambda / (D + lambda) + (1.f - ut.alpha * ut.alpha + ut.beta);
        const float wi = 1.f / (2.f * (D + lambda));
        const float spread = sqrtf(D + lambda);
        const bool require_all = ut.require_all_sigma_points_valid != 0;

        f2 ip[7];
        f2 m{0.f, 0.f};
        bool valid = require_all, bail = false;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            f3 pt = mean;
            if (i > 0) {
                const int a = (i - 1) % 3;
                const float f = spread * sc[a];
                const f3 delta{f * R.m[0][a], f * R.m[1][a], f * R.m[2][a]};
                pt = (i <= 3) ? mean + delta : mean - delta;
            }
            f2 p;
            const bool pv = cam_world_to_image(cam, pt, ut.in_image_margin_factor, p);
            if (require_all) { valid &= pv; if (!pv) { bail = true; } }
            else valid |= pv;
            ip[i] = p;
            const float w = (i == 0) ? w0m : wi;
            if (!bail) { m.x = m.x + w * p.x; m.y = m.y + w * p.y; }
        }
        if (bail || !valid) break;

        float c00 = 0.f, c01 = 0.f, c10 = 0.f, c11 = 0.f;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const float w = (i == 0) ? w0c : wi;
            const float dx = ip[i].x - m.x, dy = ip[i].y - m.y;
            c00 += w * (dx * dx); c01 += w * (dx * dy); c10 += w * (dy * dx); c11 += w * (dy * dy);
        }

        // 2D blur + compensation
        const float det_orig = c00 * c11 - c01 * c10;
        c00 += eps2d; c11 += eps2d;
        const float det = c00 * c11 - c01 * c10;
        const float compensation = sqrtf(fmaxf(0.f, det_orig / det));
        if (det <= 0.f) break;
        const float ood = 1.f / det;

        float extend = 3.33f;
        if (opacities != nullptr) {
            const float op = opac_in * compensation;
            if (op < (1.f / 255.f)) break;
            extend = fminf(extend, sqrtf(2.f * logf(op / (1.f / 255.f))));
        }
        const float b = 0.5f * (c00 + c11);
        const float tmp = sqrtf(fmaxf(0.01f, b * b - det));
        const float r1 = extend * sqrtf(b + tmp);
        const float rx = ceilf(fminf(extend * sqrtf(c00), r1));
        const float ry = ceilf(fminf(extend * sqrtf(c11), r1));
        if (rx <= radius_clip && ry <= radius_clip) break;
        if (m.x + rx <= 0.f || m.x - rx >= float(cam.width) || m.y + ry <= 0.f || m.y - ry >= float(cam.height)) break;

        out_rx = int32_t(rx); out_ry = int32_t(ry);
        o_m2x = m.x; o_m2y = m.y; o_depth = mean_c.z;
        o_c0 = c11 * ood; o_c1 = -c01 * ood; o_c2 = c00 * ood;
        o_comp = compensation;
    } while (false);

    reinterpret_cast<int2*>(radii)[idx] = make_int2(out_rx, out_ry);
    reinterpret_cast<float2*>(means2d)[idx] = make_float2(o_m2x, o_m2y);
    depths[idx] = o_depth;
    if (conics != nullptr) { conics[3 * idx] = o_c0; conics[3 * idx + 1] = o_c1; conics[3 * idx + 2] = o_c2; }
    if (compensations != nullptr) compensations[idx] = o_comp;
    if (PACK && out_rx > 0 && out_ry > 0) {
        GaussRec rec; CullRec cr;
        const f3 col = pack_colors != nullptr ? ld3(pack_colors, gid) : f3{0.f, 0.f, 0.f}; // (uniform) NULL, the pipelined step: the SH colour kernel writes the rgb slots of the finished record afterwards
        pack_gaussian<true>(cam, p_mean, p_q, p_sc, p_op, col.x, col.y, col.z, rec, cr);
        recs[idx] = rec;
        cull[idx] = cr;
    }
}

} // namespace lfs

static bool simple_camera(const lfs_cameras* c) {
    return c->camera_model == LFS_CAMERA_PINHOLE && c->rs_type == LFS_SHUTTER_GLOBAL && !c->viewmats1 && !c->radial_coeffs && !c->tangential_coeffs && !c->thin_prism_coeffs;
}

extern "C" int lfs_projection_ut_3dgs_fused(
    uint32_t N, const float* means, const float* quats, const float* scales, const float* opacities,
    const lfs_cameras* cams, float eps2d, float near_plane, float far_plane, float radius_clip,
    const lfs_ut_params* ut_params,
    int32_t* radii, float* means2d, float* depths, float* conics, float* compensations,
    lfs_stream_t stream) {
    if (!cams || !cams->viewmats0 || !cams->Ks) return LFS_E_INVALID;
    if (cams->camera_model != LFS_CAMERA_PINHOLE && cams->camera_model != LFS_CAMERA_FISHEYE) return LFS_E_UNSUPPORTED;
    if (N == 0 || cams->C == 0) return LFS_OK; // ProjectionUT3DGSFused.cu:242-245
    if (!means || !quats || !scales || !radii || !means2d || !depths || !conics) return LFS_E_INVALID;
    lfs_ut_params ut = {0.1f, 2.f, 0.f, 0.1f, 1};
    if (ut_params) ut = *ut_params;
    dim3 grid((N + 255) / 256, cams->C);
    lfs::ProfScope prof("projection_ut", (hipStream_t)stream);
    if (simple_camera(cams))
        hipLaunchKernelGGL((lfs::projection_ut_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream,
                           N, means, quats, scales, opacities, *cams, eps2d, near_plane, far_plane, radius_clip, ut,
                           radii, means2d, depths, conics, compensations, nullptr, nullptr, nullptr);
    else
        hipLaunchKernelGGL((lfs::projection_ut_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream,
                           N, means, quats, scales, opacities, *cams, eps2d, near_plane, far_plane, radius_clip, ut,
                           radii, means2d, depths, conics, compensations, nullptr, nullptr, nullptr);
    return (int)hipGetLastError();
}

int lfs::activations_project_ut_impl(
    uint32_t N, const float* means, const float* raw_quats, const float* raw_scales, const float* raw_opacities, const lfs_cameras* cams,
    float eps2d, float near_plane, float far_plane, float radius_clip, const lfs_ut_params* ut_params,
    float* quats, float* scales, float* opacities, int32_t* radii, float* means2d, float* depths, uint32_t* zero_words, uint32_t zero_n, void* cams_out, hipStream_t stream,
    void* recs_out, void* cull_out, const float* pack_colors) {
    if (!cams || !cams->viewmats0 || !cams->Ks) return LFS_E_INVALID;
    if (cams->camera_model != LFS_CAMERA_PINHOLE && cams->camera_model != LFS_CAMERA_FISHEYE) return LFS_E_UNSUPPORTED;
    if (N == 0 || cams->C == 0) return LFS_OK;
    if (!means || !raw_quats || !raw_scales || !raw_opacities || !quats || !scales || !opacities || !radii || !means2d || !depths) return LFS_E_INVALID;
    lfs_ut_params ut = {0.1f, 2.f, 0.f, 0.1f, 1};
    if (ut_params) ut = *ut_params;
    dim3 grid((N + 255) / 256, cams->C);
    lfs::ProfScope prof("activations_projection_ut", (hipStream_t)stream);
    if (recs_out != nullptr) { // the training step: records + culling records from the same pass (one camera, undistorted pinhole, global shutter)
        if (!simple_camera(cams) || cams->C != 1 || !cull_out) return LFS_E_INVALID;
        hipLaunchKernelGGL((lfs::projection_ut_kernel<true, true, true>), grid, dim3(256), 0, (hipStream_t)stream,
                           N, means, raw_quats, raw_scales, raw_opacities, *cams, eps2d, near_plane, far_plane, radius_clip, ut,
                           radii, means2d, depths, nullptr, nullptr, quats, scales, opacities, zero_words, zero_n, static_cast<lfs::CamDev*>(cams_out),
                           static_cast<lfs::GaussRec*>(recs_out), static_cast<lfs::CullRec*>(cull_out), pack_colors);
        return (int)hipGetLastError();
    }
    if (simple_camera(cams))
        hipLaunchKernelGGL((lfs::projection_ut_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream,
                           N, means, raw_quats, raw_scales, raw_opacities, *cams, eps2d, near_plane, far_plane, radius_clip, ut,
                           radii, means2d, depths, nullptr, nullptr, quats, scales, opacities, zero_words, zero_n, static_cast<lfs::CamDev*>(cams_out));
    else
        hipLaunchKernelGGL((lfs::projection_ut_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream,
                           N, means, raw_quats, raw_scales, raw_opacities, *cams, eps2d, near_plane, far_plane, radius_clip, ut,
                           radii, means2d, depths, nullptr, nullptr, quats, scales, opacities, zero_words, zero_n, static_cast<lfs::CamDev*>(cams_out));
    return (int)hipGetLastError();
}

extern "C" int lfs_activations_project_ut(
    uint32_t N, const float* means, const float* raw_quats, const float* raw_scales, const float* raw_opacities, const lfs_cameras* cams,
    float eps2d, float near_plane, float far_plane, float radius_clip, const lfs_ut_params* ut_params,
    float* quats, float* scales, float* opacities, int32_t* radii, float* means2d, float* depths, lfs_stream_t stream) {
    return lfs::activations_project_ut_impl(N, means, raw_quats, raw_scales, raw_opacities, cams, eps2d, near_plane, far_plane, radius_clip, ut_params, quats, scales,
                                            opacities, radii, means2d, depths, nullptr, 0, nullptr, (hipStream_t)stream, nullptr, nullptr, nullptr);
}
