// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).
//
// CPU restatement (templated on float / double) of every operator on the
// north-star hot path: the gsplat:: ops of
// /root/reference/gsplat/Ops.h and the fastgs fused Adam. Each function names
// the reference file:line it follows.
//
// PARITY PINNING STATUS: every operator restated here is held to the reference's own code run on this machine.
//   * spherical harmonics fwd, quat->rotmat and tile intersection: the reference's CPU code tests/torch_impl.cpp (built in place into oracle/_ref) and golden
//     vectors generated from it (tests/golden/, oracle/make_golden.py) - since round 1.
//   * ALL device kernels of the path - projection_ut_3dgs_fused, spherical harmonics fwd / bwd, the tile-intersection kernels with the radix sort and the
//     offsets, rasterize_to_pixels_from_world_3dgs_{fwd,bwd}, relocation, add_noise, quats_to_rotmats, adam_step: the reference's .cu kernels compiled in place as
//     host code under oracle/ref_emul/ (ref_kernels.cpp, `make refk`), golden vectors tests/golden/refk_*.npz (oracle/make_golden_refk*.py), tests
//     tests/test_oracle_refk_golden.py and tests/test_oracle_refk_sh_isect_golden.py - round 2. The reference's own test-suite has no tests or fixtures for them.
//   * their composition into the render + backward of a training step: the reference's rasterize() + autograd Functions + Camera on CPU libtorch
//     (ref_raster_shim.cpp, tests/golden/ref_raster.npz, tests/test_oracle_ref_raster_golden.py).
#pragma once
#include "oracle_cameras.hpp"
#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>

namespace orc {

static const float kAlphaThreshold = 1.f / 255.f; // gsplat/Common.h:53

// ---------------------------------------------------------------------------
// K12 quats_to_rotmats  (QuatToRotmatCUDA.cu:14-39: row-major store of R)
// ---------------------------------------------------------------------------
template <class T> void quats_to_rotmats(int64_t N, const T* quats, T* rotmats) {
    for (int64_t i = 0; i < N; ++i) {
        M3<T> R = quat_to_rotmat(V4<T>{quats[4 * i], quats[4 * i + 1], quats[4 * i + 2], quats[4 * i + 3]});
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) rotmats[9 * i + 3 * r + c] = R.m[r][c];
    }
}

// ---------------------------------------------------------------------------
// K1 projection_ut_3dgs_fused  (ProjectionUT3DGSFused.cu:17-203)
// ---------------------------------------------------------------------------
template <class T> struct ProjArgs {
    uint32_t C, N;
    const T *means, *quats, *scales, *opacities, *viewmats0, *viewmats1, *Ks;
    uint32_t width, height;
    T eps2d, near_plane, far_plane, radius_clip;
    int camera_model;
    UTParams<T> ut;
    int shutter;
    const T* radial; int n_radial;
    const T* tangential;
    const T* thin_prism; int n_thin;
    int32_t* radii; T *means2d, *depths, *conics, *compensations;
};

template <class T> void projection_ut_3dgs_fused(const ProjArgs<T>& a) {
#pragma omp parallel for schedule(static)
    for (int64_t idx = 0; idx < int64_t(a.C) * a.N; ++idx) {
        const uint32_t cid = idx / a.N, gid = idx % a.N;
        auto cull = [&]() { a.radii[2 * idx] = 0; a.radii[2 * idx + 1] = 0; };
        V3<T> mean{a.means[3 * gid], a.means[3 * gid + 1], a.means[3 * gid + 2]};
        V3<T> scale{a.scales[3 * gid], a.scales[3 * gid + 1], a.scales[3 * gid + 2]};
        Q4<T> quat = qnormalize(Q4<T>{a.quats[4 * gid], a.quats[4 * gid + 1], a.quats[4 * gid + 2], a.quats[4 * gid + 3]});
        RSParams<T> rs(a.viewmats0 + 16 * cid, a.viewmats1 ? a.viewmats1 + 16 * cid : nullptr);
        // :74-82 depth test at the centre-of-exposure pose
        ShutterPose<T> pose = interpolate_shutter_pose(T(0.5f), rs);
        V3<T> mean_c = qrotate(pose.q, mean) + pose.t;
        if (mean_c.z < a.near_plane || mean_c.z > a.far_plane) { cull(); continue; }

        const int radial_stride = (a.camera_model == FISHEYE) ? 4 : a.n_radial;
        Camera<T> cam = make_camera<T>(
            a.camera_model, a.width, a.height, a.shutter, a.Ks + 9 * cid,
            a.radial ? a.radial + radial_stride * cid : nullptr, a.n_radial,
            a.tangential ? a.tangential + 2 * cid : nullptr,
            a.thin_prism ? a.thin_prism + a.n_thin * cid : nullptr, a.n_thin);
        ImageGaussian<T> ig = unscented_transform(cam, rs, a.ut, mean, scale, quat);
        if (!ig.valid) { cull(); continue; }

        // Utils.cuh:171-179 add_blur
        M2<T> cov = ig.cov;
        T det_orig = cov.m[0][0] * cov.m[1][1] - cov.m[0][1] * cov.m[1][0];
        cov.m[0][0] += a.eps2d; cov.m[1][1] += a.eps2d;
        T det = cov.m[0][0] * cov.m[1][1] - cov.m[0][1] * cov.m[1][0];
        T compensation = std::sqrt(std::max(T(0), det_orig / det));
        if (det <= T(0)) { cull(); continue; }
        // glm::inverse(mat2)
        T ood = T(1) / det;
        T inv00 = cov.m[1][1] * ood, inv01 = -cov.m[0][1] * ood, inv11 = cov.m[0][0] * ood;

        T extend = T(3.33f);
        if (a.opacities) {
            T op = a.opacities[gid] * compensation;
            if (op < T(kAlphaThreshold)) { cull(); continue; }
            extend = std::min(extend, std::sqrt(T(2) * std::log(op / T(kAlphaThreshold))));
        }
        // :168-175 tight rectangular bound
        T b = T(0.5f) * (cov.m[0][0] + cov.m[1][1]);
        T tmp = std::sqrt(std::max(T(0.01f), b * b - det));
        T v1 = b + tmp;
        T r1 = extend * std::sqrt(v1);
        T rx = std::ceil(std::min(extend * std::sqrt(cov.m[0][0]), r1));
        T ry = std::ceil(std::min(extend * std::sqrt(cov.m[1][1]), r1));
        if (rx <= a.radius_clip && ry <= a.radius_clip) { cull(); continue; }
        if (ig.mean.x + rx <= 0 || ig.mean.x - rx >= T(a.width) || ig.mean.y + ry <= 0 || ig.mean.y - ry >= T(a.height)) { cull(); continue; }
        a.radii[2 * idx] = int32_t(rx); a.radii[2 * idx + 1] = int32_t(ry);
        a.means2d[2 * idx] = ig.mean.x; a.means2d[2 * idx + 1] = ig.mean.y;
        a.depths[idx] = mean_c.z;
        a.conics[3 * idx] = inv00; a.conics[3 * idx + 1] = inv01; a.conics[3 * idx + 2] = inv11;
        if (a.compensations) a.compensations[idx] = compensation;
    }
}

// ---------------------------------------------------------------------------
// K2 / K9 spherical harmonics (SphericalHarmonicsCUDA.cu:20-110, 112-371)
// Sloan's fast evaluation; `want_grad` also fills d(basis)/d(x,y,z).
// ---------------------------------------------------------------------------
template <class T> inline void sh_basis(int degree, T x, T y, T z, T* b, T* bx, T* by, T* bz, bool want_grad) {
    const int K = (degree + 1) * (degree + 1);
    if (want_grad) for (int k = 0; k < K; ++k) bx[k] = by[k] = bz[k] = T(0);
    b[0] = T(0.2820947917738781f);
    if (degree < 1) return;
    const T c1 = T(0.48860251190292f);
    b[1] = -c1 * y; b[2] = c1 * z; b[3] = -c1 * x;
    if (want_grad) { by[1] = -c1; bz[2] = c1; bx[3] = -c1; }
    if (degree < 2) return;
    const T z2 = z * z;
    const T t0B = T(-1.092548430592079f) * z;
    const T fC1 = x * x - y * y, fS1 = T(2) * x * y;
    const T c2 = T(0.5462742152960395f);
    b[4] = c2 * fS1; b[5] = t0B * y; b[6] = T(0.9461746957575601f) * z2 - T(0.3153915652525201f);
    b[7] = t0B * x; b[8] = c2 * fC1;
    const T fC1_x = T(2) * x, fC1_y = T(-2) * y, fS1_x = T(2) * y, fS1_y = T(2) * x;
    const T b6_z = T(2) * T(0.9461746957575601f) * z;
    if (want_grad) {
        bx[4] = c2 * fS1_x; by[4] = c2 * fS1_y;
        by[5] = t0B; bz[5] = T(-1.092548430592079f) * y;
        bz[6] = b6_z;
        bx[7] = t0B; bz[7] = T(-1.092548430592079f) * x;
        bx[8] = c2 * fC1_x; by[8] = c2 * fC1_y;
    }
    if (degree < 3) return;
    const T t0C = T(-2.285228997322329f) * z2 + T(0.4570457994644658f);
    const T t1B = T(1.445305721320277f) * z;
    const T fC2 = x * fC1 - y * fS1, fS2 = x * fS1 + y * fC1;
    const T c3 = T(-0.5900435899266435f);
    b[9] = c3 * fS2; b[10] = t1B * fS1; b[11] = t0C * y;
    b[12] = z * (T(1.865881662950577f) * z2 - T(1.119528997770346f));
    b[13] = t0C * x; b[14] = t1B * fC1; b[15] = c3 * fC2;
    const T fC2_x = fC1 + x * fC1_x - y * fS1_x, fC2_y = x * fC1_y - fS1 - y * fS1_y;
    const T fS2_x = fS1 + x * fS1_x + y * fC1_x, fS2_y = x * fS1_y + fC1 + y * fC1_y;
    const T b12_z = T(3) * T(1.865881662950577f) * z2 - T(1.119528997770346f);
    if (want_grad) {
        const T t0C_z = T(-2.285228997322329f) * T(2) * z, t1B_z = T(1.445305721320277f);
        bx[9] = c3 * fS2_x; by[9] = c3 * fS2_y;
        bx[10] = t1B * fS1_x; by[10] = t1B * fS1_y; bz[10] = t1B_z * fS1;
        by[11] = t0C; bz[11] = t0C_z * y;
        bz[12] = b12_z;
        bx[13] = t0C; bz[13] = t0C_z * x;
        bx[14] = t1B * fC1_x; by[14] = t1B * fC1_y; bz[14] = t1B_z * fC1;
        bx[15] = c3 * fC2_x; by[15] = c3 * fC2_y;
    }
    if (degree < 4) return;
    const T t0D = z * (T(-4.683325804901025f) * z2 + T(2.007139630671868f));
    const T t1C = T(3.31161143515146f) * z2 - T(0.47308734787878f);
    const T t2B = T(-1.770130769779931f) * z;
    const T fC3 = x * fC2 - y * fS2, fS3 = x * fS2 + y * fC2;
    const T c4 = T(0.6258357354491763f);
    b[16] = c4 * fS3; b[17] = t2B * fS2; b[18] = t1C * fS1; b[19] = t0D * y;
    b[20] = T(1.984313483298443f) * z * b[12] - T(1.006230589874905f) * b[6];
    b[21] = t0D * x; b[22] = t1C * fC1; b[23] = t2B * fC2; b[24] = c4 * fC3;
    if (want_grad) {
        const T t0D_z = T(3) * T(-4.683325804901025f) * z2 + T(2.007139630671868f);
        const T t1C_z = T(2) * T(3.31161143515146f) * z, t2B_z = T(-1.770130769779931f);
        const T fC3_x = fC2 + x * fC2_x - y * fS2_x, fC3_y = x * fC2_y - fS2 - y * fS2_y;
        const T fS3_x = fS2 + y * fC2_x + x * fS2_x, fS3_y = x * fS2_y + fC2 + y * fC2_y;
        bx[16] = c4 * fS3_x; by[16] = c4 * fS3_y;
        bx[17] = t2B * fS2_x; by[17] = t2B * fS2_y; bz[17] = t2B_z * fS2;
        bx[18] = t1C * fS1_x; by[18] = t1C * fS1_y; bz[18] = t1C_z * fS1;
        by[19] = t0D; bz[19] = t0D_z * y;
        bz[20] = T(1.984313483298443f) * (b[12] + z * b12_z) - T(1.006230589874905f) * b6_z;
        bx[21] = t0D; bz[21] = t0D_z * x;
        bx[22] = t1C * fC1_x; by[22] = t1C * fC1_y; bz[22] = t1C_z * fC1;
        bx[23] = t2B * fC2_x; by[23] = t2B * fC2_y; bz[23] = t2B_z * fC2;
        bx[24] = c4 * fC3_x; by[24] = c4 * fC3_y;
    }
}

// colors for masked-out elements are left untouched (reference: at::empty_like,
// SphericalHarmonics.cpp:30; the kernel returns early :392-394).
template <class T> void spherical_harmonics_fwd(
    int64_t N, int K, int degree, const T* dirs, const T* coeffs, const uint8_t* masks, T* colors) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        if (masks && !masks[i]) continue;
        T b[25], dummy[1];
        T x = dirs[3 * i], y = dirs[3 * i + 1], z = dirs[3 * i + 2];
        if (degree >= 1) { T inorm = T(1) / std::sqrt(x * x + y * y + z * z); x *= inorm; y *= inorm; z *= inorm; }
        sh_basis<T>(degree, x, y, z, b, dummy, dummy, dummy, false);
        const T* cf = coeffs + int64_t(i) * K * 3;
        for (int c = 0; c < 3; ++c) {
            T r = b[0] * cf[c];
            if (degree >= 1) r += T(0.48860251190292f) * (-y * cf[3 + c] + z * cf[6 + c] - x * cf[9 + c]);
            int k0 = 4;
            for (int d = 2; d <= degree; ++d) {
                int k1 = (d + 1) * (d + 1);
                T s = b[k0] * cf[3 * k0 + c];
                for (int k = k0 + 1; k < k1; ++k) s += b[k] * cf[3 * k + c];
                r += s; k0 = k1;
            }
            colors[3 * i + c] = r;
        }
    }
}

// v_coeffs / v_dirs must be zero-initialised by the caller (reference:
// zeros_like, SphericalHarmonics.cpp:57-60).
template <class T> void spherical_harmonics_bwd(
    int64_t N, int K, int degree, const T* dirs, const T* coeffs, const uint8_t* masks,
    const T* v_colors, T* v_coeffs, T* v_dirs) {
    const int Kd = (degree + 1) * (degree + 1);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        if (masks && !masks[i]) continue;
        T b[25], bx[25], by[25], bz[25];
        T x = dirs[3 * i], y = dirs[3 * i + 1], z = dirs[3 * i + 2];
        T inorm = T(1);
        if (degree >= 1) { inorm = T(1) / std::sqrt(x * x + y * y + z * z); x *= inorm; y *= inorm; z *= inorm; }
        sh_basis<T>(degree, x, y, z, b, bx, by, bz, v_dirs != nullptr);
        const T* cf = coeffs + int64_t(i) * K * 3;
        T* vcf = v_coeffs + int64_t(i) * K * 3;
        T acc[3] = {T(0), T(0), T(0)};
        for (int c = 0; c < 3; ++c) {
            const T vc = v_colors[3 * i + c];
            for (int k = 0; k < Kd; ++k) vcf[3 * k + c] = b[k] * vc;
            if (v_dirs && degree >= 1) {
                T vx = T(0), vy = T(0), vz = T(0);
                int k0 = 1;
                for (int d = 1; d <= degree; ++d) {
                    int k1 = (d + 1) * (d + 1);
                    T sx = T(0), sy = T(0), sz = T(0);
                    for (int k = k0; k < k1; ++k) { sx += bx[k] * cf[3 * k + c]; sy += by[k] * cf[3 * k + c]; sz += bz[k] * cf[3 * k + c]; }
                    vx += vc * sx; vy += vc * sy; vz += vc * sz; k0 = k1;
                }
                // :146-148 project on the tangent plane and undo the normalisation
                T d = vx * x + vy * y + vz * z;
                acc[0] += (vx - d * x) * inorm; acc[1] += (vy - d * y) * inorm; acc[2] += (vz - d * z) * inorm;
            }
        }
        if (v_dirs) { v_dirs[3 * i] += acc[0]; v_dirs[3 * i + 1] += acc[1]; v_dirs[3 * i + 2] += acc[2]; }
    }
}

// ---------------------------------------------------------------------------
// K3-K6 tile intersection (IntersectTile.cu:24-113, 206-252, 290-342;
// Intersect.cpp:15-137). Integer stage: bit-exact contract.
// ---------------------------------------------------------------------------
inline uint32_t sat_u32(float v) { // CUDA float->uint32 cvt saturates
    if (!(v > 0.f)) return 0u;
    if (v >= 4294967296.f) return 0xFFFFFFFFu;
    return uint32_t(v);
}
struct TileRect { uint32_t x0, y0, x1, y1; };
inline bool tile_rect(const float* means2d, const int32_t* radii, int64_t idx, uint32_t ts, uint32_t tw, uint32_t th, TileRect& r) {
    const float rx = float(radii[2 * idx]), ry = float(radii[2 * idx + 1]);
    if (rx <= 0 || ry <= 0) return false;
    const float trx = rx / float(ts), try_ = ry / float(ts);
    const float tx = means2d[2 * idx] / float(ts), ty = means2d[2 * idx + 1] / float(ts);
    r.x0 = std::min(sat_u32(std::floor(tx - trx)), tw);
    r.y0 = std::min(sat_u32(std::floor(ty - try_)), th);
    r.x1 = std::min(sat_u32(std::ceil(tx + trx)), tw);
    r.y1 = std::min(sat_u32(std::ceil(ty + try_)), th);
    return true;
}
inline uint32_t bit_count_floor_log2_plus1(uint32_t v) { return uint32_t(std::floor(std::log2(double(v)))) + 1; } // IntersectTile.cu:150-151

// pass 1: tiles_per_gauss[C*N]; returns n_isects
inline int64_t intersect_tile_count(uint32_t C, uint32_t N, const float* means2d, const int32_t* radii,
                                    uint32_t ts, uint32_t tw, uint32_t th, int32_t* tiles_per_gauss) {
    int64_t total = 0;
    for (int64_t idx = 0; idx < int64_t(C) * N; ++idx) {
        TileRect r;
        int32_t n = 0;
        if (tile_rect(means2d, radii, idx, ts, tw, th, r)) n = int32_t((r.y1 - r.y0) * (r.x1 - r.x0));
        tiles_per_gauss[idx] = n; total += n;
    }
    return total;
}
// pass 2 (+ optional stable sort on the low 32+tile_n_bits+cam_n_bits bits)
inline void intersect_tile_emit(uint32_t C, uint32_t N, const float* means2d, const int32_t* radii, const float* depths,
                                uint32_t ts, uint32_t tw, uint32_t th, bool sort, int64_t n_isects,
                                int64_t* isect_ids, int32_t* flatten_ids) {
    const uint32_t tile_n_bits = bit_count_floor_log2_plus1(tw * th);
    int64_t cur = 0;
    for (int64_t idx = 0; idx < int64_t(C) * N; ++idx) {
        TileRect r;
        if (!tile_rect(means2d, radii, idx, ts, tw, th, r)) continue;
        const int64_t cid = idx / N;
        const int64_t cid_enc = cid << (32 + tile_n_bits);
        uint32_t dbits; std::memcpy(&dbits, depths + idx, 4);
        for (uint32_t i = r.y0; i < r.y1; ++i)
            for (uint32_t j = r.x0; j < r.x1; ++j) {
                int64_t tile_id = int64_t(i) * tw + j;
                isect_ids[cur] = cid_enc | (tile_id << 32) | int64_t(dbits);
                flatten_ids[cur] = int32_t(idx);
                ++cur;
            }
    }
    (void)n_isects;
    if (sort && cur > 0) {
        std::vector<int64_t> perm(cur);
        std::iota(perm.begin(), perm.end(), 0);
        std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return uint64_t(isect_ids[a]) < uint64_t(isect_ids[b]); });
        std::vector<int64_t> k(cur); std::vector<int32_t> v(cur);
        for (int64_t i = 0; i < cur; ++i) { k[i] = isect_ids[perm[i]]; v[i] = flatten_ids[perm[i]]; }
        std::memcpy(isect_ids, k.data(), cur * 8); std::memcpy(flatten_ids, v.data(), cur * 4);
    }
}
// K6: lower-bound offsets per (camera, tile)
inline void intersect_offset(int64_t n_isects, const int64_t* isect_ids, uint32_t C, uint32_t tw, uint32_t th, int32_t* offsets) {
    const uint32_t n_tiles = tw * th;
    const int64_t total = int64_t(C) * n_tiles;
    if (n_isects == 0) { for (int64_t i = 0; i < total; ++i) offsets[i] = 0; return; }
    const uint32_t tile_n_bits = bit_count_floor_log2_plus1(n_tiles);
    auto flat = [&](int64_t key) { int64_t hi = key >> 32; return (hi >> tile_n_bits) * n_tiles + (hi & ((int64_t(1) << tile_n_bits) - 1)); };
    int64_t t = 0;
    for (int64_t i = 0; i < n_isects; ++i) {
        int64_t id = flat(isect_ids[i]);
        for (; t <= id && t < total; ++t) offsets[t] = int32_t(i);
    }
    for (; t < total; ++t) offsets[t] = int32_t(n_isects);
}

// ---------------------------------------------------------------------------
// K7 rasterize_to_pixels_from_world_3dgs_fwd (RasterizeToPixelsFromWorld3DGSFwd.cu:19-279)
// ---------------------------------------------------------------------------
template <class T> struct RasterArgs {
    uint32_t C, N; int64_t n_isects; uint32_t cdim;
    const T *means, *quats, *scales, *colors, *opacities, *backgrounds;
    const uint8_t* masks;
    uint32_t width, height, tile_size, tile_width, tile_height;
    const T *viewmats0, *viewmats1, *Ks;
    int camera_model; int shutter;
    const T* radial; int n_radial; const T* tangential; const T* thin_prism; int n_thin;
    const int32_t *tile_offsets, *flatten_ids;
};

template <class T> inline Camera<T> raster_camera(const RasterArgs<T>& a, uint32_t cid) {
    const int radial_stride = (a.camera_model == FISHEYE) ? 4 : a.n_radial;
    return make_camera<T>(a.camera_model, a.width, a.height, a.shutter, a.Ks + 9 * cid,
                          a.radial ? a.radial + radial_stride * cid : nullptr, a.n_radial,
                          a.tangential ? a.tangential + 2 * cid : nullptr,
                          a.thin_prism ? a.thin_prism + a.n_thin * cid : nullptr, a.n_thin);
}

// Per-Gaussian data shared by fwd and bwd (Fwd.cu:196-220): M = diag(1/s) R^T.
// Geometry is indexed with g % N, colours/opacities with the flattened id g
// (identical to the reference at C == 1).
template <class T> struct GaussGeom { V3<T> xyz; T opac; M3<T> M, R; V4<T> quat; V3<T> scale; };
template <class T> inline GaussGeom<T> load_gauss(const RasterArgs<T>& a, int32_t g) {
    const int64_t gid = int64_t(g) % a.N;
    GaussGeom<T> G;
    G.xyz = {a.means[3 * gid], a.means[3 * gid + 1], a.means[3 * gid + 2]};
    G.opac = a.opacities[g];
    G.quat = {a.quats[4 * gid], a.quats[4 * gid + 1], a.quats[4 * gid + 2], a.quats[4 * gid + 3]};
    G.scale = {a.scales[3 * gid], a.scales[3 * gid + 1], a.scales[3 * gid + 2]};
    G.R = quat_to_rotmat(G.quat);
    const T is[3] = {T(1) / G.scale.x, T(1) / G.scale.y, T(1) / G.scale.z};
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) G.M.m[r][c] = is[r] * G.R.m[c][r];
    return G;
}

template <class T> void rasterize_fwd(const RasterArgs<T>& a, T* render_colors, T* render_alphas, int32_t* last_ids) {
    const uint32_t ts = a.tile_size, CD = a.cdim;
    const int64_t n_tiles = int64_t(a.tile_width) * a.tile_height;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t bt = 0; bt < int64_t(a.C) * n_tiles; ++bt) {
        const uint32_t cid = bt / n_tiles; const int64_t tile_id = bt % n_tiles;
        const uint32_t ty = tile_id / a.tile_width, tx = tile_id % a.tile_width;
        const int32_t* offs = a.tile_offsets + cid * n_tiles;
        T* rc = render_colors + int64_t(cid) * a.height * a.width * CD;
        T* ra = render_alphas + int64_t(cid) * a.height * a.width;
        int32_t* li = last_ids + int64_t(cid) * a.height * a.width;
        const T* bg = a.backgrounds ? a.backgrounds + cid * CD : nullptr;
        Camera<T> cam = raster_camera(a, cid);
        RSParams<T> rs(a.viewmats0 + 16 * cid, a.viewmats1 ? a.viewmats1 + 16 * cid : nullptr);
        const int32_t range_start = offs[tile_id];
        const int32_t range_end = (cid == a.C - 1 && tile_id == n_tiles - 1) ? int32_t(a.n_isects) : offs[tile_id + 1];
        const bool tile_masked = a.masks && !a.masks[cid * n_tiles + tile_id];
        std::vector<GaussGeom<T>> gs; std::vector<int32_t> ids;
        if (!tile_masked) {
            gs.reserve(std::max(0, range_end - range_start));
            for (int32_t k = range_start; k < range_end; ++k) { ids.push_back(a.flatten_ids[k]); gs.push_back(load_gauss(a, a.flatten_ids[k])); }
        }
        std::vector<T> pix(CD);
        for (uint32_t py_ = 0; py_ < ts; ++py_) for (uint32_t px_ = 0; px_ < ts; ++px_) {
            const uint32_t i = ty * ts + py_, j = tx * ts + px_;
            if (!(i < a.height && j < a.width)) continue;
            const int64_t pix_id = int64_t(i) * a.width + j;
            if (tile_masked) { // Fwd.cu:141-150 (alpha / last_ids are zeroed here; the reference leaves them uninitialised)
                for (uint32_t k = 0; k < CD; ++k) rc[pix_id * CD + k] = bg ? bg[k] : T(0);
                ra[pix_id] = T(0); li[pix_id] = 0;
                continue;
            }
            Ray<T> ray = cam.pixel_ray({T(j) + T(0.5f), T(i) + T(0.5f)}, rs);
            bool done = !ray.valid;
            T Tr = T(1); uint32_t cur_idx = 0;
            std::fill(pix.begin(), pix.end(), T(0));
            for (int32_t k = 0; k < int32_t(gs.size()) && !done; ++k) {
                const GaussGeom<T>& G = gs[k];
                V3<T> gro = mul(G.M, ray.o - G.xyz);
                V3<T> grd = safe_normalize(mul(G.M, ray.d));
                V3<T> gc = cross(grd, gro);
                T power = T(-0.5f) * dot(gc, gc);
                T alpha = std::min(T(0.999f), G.opac * std::exp(power));
                if (alpha < T(kAlphaThreshold)) continue;
                T next_T = Tr * (T(1) - alpha);
                if (next_T <= T(1e-4f)) { done = true; break; }
                T vis = alpha * Tr;
                const T* cp = a.colors + int64_t(ids[k]) * CD;
                for (uint32_t c = 0; c < CD; ++c) pix[c] += cp[c] * vis;
                cur_idx = uint32_t(range_start + k);
                Tr = next_T;
            }
            ra[pix_id] = T(1) - Tr;
            for (uint32_t c = 0; c < CD; ++c) rc[pix_id * CD + c] = bg ? pix[c] + Tr * bg[c] : pix[c];
            li[pix_id] = int32_t(cur_idx);
        }
    }
}

// ---------------------------------------------------------------------------
// K8 rasterize_to_pixels_from_world_3dgs_bwd (RasterizeToPixelsFromWorld3DGSBwd.cu:16-373)
// Per-(pixel,Gaussian) terms follow the kernel in T; the sum over pixels (a
// warp reduce + atomicAdd of unspecified order in the reference) is taken in
// double so the oracle is deterministic. Outputs must be zero-initialised.
// ---------------------------------------------------------------------------
template <class T> void rasterize_bwd(const RasterArgs<T>& a, const T* render_alphas, const int32_t* last_ids,
                                      const T* v_render_colors, const T* v_render_alphas,
                                      T* v_means, T* v_quats, T* v_scales, T* v_colors, T* v_opacities) {
    const uint32_t ts = a.tile_size, CD = a.cdim;
    const int64_t n_tiles = int64_t(a.tile_width) * a.tile_height;
    const int64_t CN = int64_t(a.C) * a.N;
    std::vector<double> am(3 * a.N, 0.0), aq(4 * a.N, 0.0), as(3 * a.N, 0.0), ac(CN * CD, 0.0), ao(CN, 0.0);
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t bt = 0; bt < int64_t(a.C) * n_tiles; ++bt) {
        const uint32_t cid = bt / n_tiles; const int64_t tile_id = bt % n_tiles;
        const uint32_t ty = tile_id / a.tile_width, tx = tile_id % a.tile_width;
        const int32_t* offs = a.tile_offsets + cid * n_tiles;
        const T* ralpha = render_alphas + int64_t(cid) * a.height * a.width;
        const int32_t* li = last_ids + int64_t(cid) * a.height * a.width;
        const T* vrc = v_render_colors + int64_t(cid) * a.height * a.width * CD;
        const T* vra = v_render_alphas + int64_t(cid) * a.height * a.width;
        const T* bg = a.backgrounds ? a.backgrounds + cid * CD : nullptr;
        if (a.masks && !a.masks[cid * n_tiles + tile_id]) continue; // masked tiles never composited anything
        Camera<T> cam = raster_camera(a, cid);
        RSParams<T> rs(a.viewmats0 + 16 * cid, a.viewmats1 ? a.viewmats1 + 16 * cid : nullptr);
        const int32_t range_start = offs[tile_id];
        const int32_t range_end = (cid == a.C - 1 && tile_id == n_tiles - 1) ? int32_t(a.n_isects) : offs[tile_id + 1];
        const int32_t n = std::max(0, range_end - range_start);
        if (n == 0) continue;
        std::vector<GaussGeom<T>> gs(n); std::vector<int32_t> ids(n);
        for (int32_t k = 0; k < n; ++k) { ids[k] = a.flatten_ids[range_start + k]; gs[k] = load_gauss(a, ids[k]); }
        // local accumulators for this tile (double)
        std::vector<double> lm(3 * n, 0.0), lq(4 * n, 0.0), ls(3 * n, 0.0), lc(size_t(CD) * n, 0.0), lo(n, 0.0);
        std::vector<T> buffer(CD), vc(CD);
        for (uint32_t py_ = 0; py_ < ts; ++py_) for (uint32_t px_ = 0; px_ < ts; ++px_) {
            const uint32_t i = ty * ts + py_, j = tx * ts + px_;
            if (!(i < a.height && j < a.width)) continue;
            const int64_t pix_id = int64_t(i) * a.width + j;
            Ray<T> ray = cam.pixel_ray({T(j) + T(0.5f), T(i) + T(0.5f)}, rs);
            if (!ray.valid) continue;
            const T T_final = T(1) - ralpha[pix_id];
            T Tr = T_final;
            std::fill(buffer.begin(), buffer.end(), T(0));
            const int32_t bin_final = li[pix_id];
            for (uint32_t c = 0; c < CD; ++c) vc[c] = vrc[pix_id * CD + c];
            const T v_ra = vra[pix_id];
            for (int32_t k = n - 1; k >= 0; --k) { // back to front
                if (range_start + k > bin_final) continue;
                const GaussGeom<T>& G = gs[k];
                V3<T> omm = ray.o - G.xyz;
                V3<T> gro = mul(G.M, omm);
                V3<T> grd = mul(G.M, ray.d);
                V3<T> grd_n = safe_normalize(grd);
                V3<T> gc = cross(grd_n, gro);
                T power = T(-0.5f) * dot(gc, gc);
                T vis = std::exp(power);
                T alpha = std::min(T(0.999f), G.opac * vis);
                if (power > T(0) || alpha < T(kAlphaThreshold)) continue;
                T ra_ = T(1) / (T(1) - alpha);
                Tr *= ra_;
                const T fac = alpha * Tr;
                const T* cp = a.colors + int64_t(ids[k]) * CD;
                T v_alpha = T(0);
                for (uint32_t c = 0; c < CD; ++c) {
                    lc[size_t(k) * CD + c] += double(fac * vc[c]);
                    v_alpha += (cp[c] * Tr - buffer[c] * ra_) * vc[c];
                }
                v_alpha += T_final * ra_ * v_ra;
                if (bg) {
                    T accum = T(0);
                    for (uint32_t c = 0; c < CD; ++c) accum += bg[c] * vc[c];
                    v_alpha += -T_final * ra_ * accum;
                }
                if (G.opac * vis <= T(0.999f)) {
                    const T v_vis = G.opac * v_alpha;
                    const T v_gd = T(-0.5f) * vis * v_vis;
                    V3<T> v_gc = (T(2) * v_gd) * gc;
                    V3<T> v_grd_n = -cross(v_gc, gro);
                    V3<T> v_gro = cross(v_gc, grd_n);
                    V3<T> v_grd = safe_normalize_bw(grd, v_grd_n);
                    // v_Mt = v_grd (x) d + v_gro (x) (o - mu)   [dL/dM, M = S^-1 R^T]
                    M3<T> v_M = add(outer(v_grd, ray.d), outer(v_gro, omm));
                    V3<T> v_omm = mul(transpose(G.M), v_gro);
                    V4<T> vq{T(0), T(0), T(0), T(0)}; V3<T> vs{T(0), T(0), T(0)};
                    // P = R S = M^T  ->  dL/dP = (dL/dM)^T
                    quat_scale_to_preci_half_vjp(G.quat, G.scale, G.R, transpose(v_M), vq, vs);
                    lm[3 * k] += double(-v_omm.x); lm[3 * k + 1] += double(-v_omm.y); lm[3 * k + 2] += double(-v_omm.z);
                    lq[4 * k] += double(vq.x); lq[4 * k + 1] += double(vq.y); lq[4 * k + 2] += double(vq.z); lq[4 * k + 3] += double(vq.w);
                    ls[3 * k] += double(vs.x); ls[3 * k + 1] += double(vs.y); ls[3 * k + 2] += double(vs.z);
                    lo[k] += double(vis * v_alpha);
                }
                for (uint32_t c = 0; c < CD; ++c) buffer[c] += cp[c] * fac;
            }
        }
#pragma omp critical
        {
            for (int32_t k = 0; k < n; ++k) {
                const int64_t g = ids[k], gid = g % a.N;
                for (int d = 0; d < 3; ++d) { am[3 * gid + d] += lm[3 * k + d]; as[3 * gid + d] += ls[3 * k + d]; }
                for (int d = 0; d < 4; ++d) aq[4 * gid + d] += lq[4 * k + d];
                for (uint32_t c = 0; c < CD; ++c) ac[g * CD + c] += lc[size_t(k) * CD + c];
                ao[g] += lo[k];
            }
        }
    }
    for (int64_t i = 0; i < 3 * int64_t(a.N); ++i) { v_means[i] += T(am[i]); v_scales[i] += T(as[i]); }
    for (int64_t i = 0; i < 4 * int64_t(a.N); ++i) v_quats[i] += T(aq[i]);
    for (int64_t i = 0; i < CN * CD; ++i) v_colors[i] += T(ac[i]);
    for (int64_t i = 0; i < CN; ++i) v_opacities[i] += T(ao[i]);
}

// ---------------------------------------------------------------------------
// K10 relocation (RelocationCUDA.cu:12-43)
// ---------------------------------------------------------------------------
template <class T> void relocation(int64_t N, const T* opacities, const T* scales, const int32_t* ratios,
                                   const T* binoms, int n_max, T* new_opacities, T* new_scales) {
    for (int64_t idx = 0; idx < N; ++idx) {
        const int n_idx = ratios[idx];
        T denom = T(0);
        const T no = T(1) - std::pow(T(1) - opacities[idx], T(1) / T(n_idx));
        new_opacities[idx] = no;
        for (int i = 1; i <= n_idx; ++i)
            for (int k = 0; k <= i - 1; ++k) {
                T bin = binoms[(i - 1) * n_max + k];
                T term = (std::pow(T(-1), T(k)) / std::sqrt(T(k + 1))) * std::pow(no, T(k + 1));
                denom += bin * term;
            }
        T coeff = opacities[idx] / denom;
        for (int d = 0; d < 3; ++d) new_scales[3 * idx + d] = coeff * scales[3 * idx + d];
    }
}

// ---------------------------------------------------------------------------
// K11 add_noise (RelocationCUDA.cu:88-144), in place on means
// ---------------------------------------------------------------------------
template <class T> void add_noise(int64_t N, const T* raw_opacities, const T* raw_scales, const T* raw_quats,
                                  const T* noise, T* means, T current_lr) {
    for (int64_t idx = 0; idx < N; ++idx) {
        const T s2[3] = {std::exp(T(2) * raw_scales[3 * idx]), std::exp(T(2) * raw_scales[3 * idx + 1]), std::exp(T(2) * raw_scales[3 * idx + 2])};
        M3<T> R = quat_to_rotmat(V4<T>{raw_quats[4 * idx], raw_quats[4 * idx + 1], raw_quats[4 * idx + 2], raw_quats[4 * idx + 3]}, T(1e12f));
        M3<T> RS;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) RS.m[r][c] = R.m[r][c] * s2[c];
        M3<T> cov = mul(RS, transpose(R));
        V3<T> tn = mul(cov, V3<T>{noise[3 * idx], noise[3 * idx + 1], noise[3 * idx + 2]});
        T opacity = T(1) / (T(1) + std::exp(-raw_opacities[idx]));
        T op_sig = T(1) / (T(1) + std::exp(T(100) * opacity - T(0.5f)));
        T nf = current_lr * op_sig;
        means[3 * idx] += nf * tn.x; means[3 * idx + 1] += nf * tn.y; means[3 * idx + 2] += nf * tn.z;
    }
}

// ---------------------------------------------------------------------------
// K13 Adam (fastgs/optimizer/include/adam_kernels.cuh:13-36)
// ---------------------------------------------------------------------------
template <class T> void adam_step(int64_t n, T* param, T* exp_avg, T* exp_avg_sq, const T* grad,
                                  T lr, T beta1, T beta2, T eps, T bc1_rcp, T bc2_sqrt_rcp) {
    for (int64_t i = 0; i < n; ++i) {
        const T g = grad[i];
        const T m1 = beta1 * exp_avg[i] + (T(1) - beta1) * g;
        const T m2 = beta2 * exp_avg_sq[i] + (T(1) - beta2) * g * g;
        const T denom = std::sqrt(m2) * bc2_sqrt_rcp + eps;
        const T step = lr * bc1_rcp;
        param[i] -= step * m1 / denom;
        exp_avg[i] = m1; exp_avg_sq[i] = m2;
    }
}

} // namespace orc
