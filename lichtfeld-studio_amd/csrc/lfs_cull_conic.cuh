// Exact per-cell culling of the world-space rasterizer (raster.hip): the silhouette conic of a Gaussian's alpha >= 1/255 ellipsoid and its
// maximum over a cell's box of rays. Plain C++ (no HIP types) so that the same code is compiled into the kernels and into the host-side
// brute-force test (tests/cull_conic_host.cpp, run by tests/test_cull_conic.py).
//
// A pixel can only composite a Gaussian when opac * exp(-D^2 / 2) >= 1/255, D = Mahalanobis distance from the centre p to the pixel's ray
// LINE, i.e. D^2 <= r^2 = 2 ln(255 opac). In camera space with Sigma = A A^T the rays d = (u, v, 1) that satisfy it are
//     f(d) = r^2 d^T adj(Sigma) d - |A^T (p x d)|^2 >= 0,
// both terms sums of squares (adj(Sigma) = C C^T with C the cofactor matrix of A), so the coefficients carry no cancellation even for
// needles. Once the ellipsoid lies in front of the camera plane this is an ellipse in (u, v); the record stores f as a quadratic in
// (du, dv) = (u - p.x/p.z, v - p.y/p.z) divided by its negated dv^2 coefficient (> 0):
//     f' = a du^2 + 2 b du dv - dv^2 + 2 d du + 2 e dv + g.
// f' is concave, and the centre ray (du = dv = 0) is inside the ellipse: its maximum over a box is positive if the box contains the centre
// ray, otherwise it is attained on one of the four edges (1-D concave quadratics: clamped vertex). g = +inf encodes "never cull", a conic
// that is negative everywhere encodes "always culled" (opacity below 1/255).
#pragma once
#include <math.h>
#ifndef LFS_CONIC_FN
#if defined(__HIPCC__)
#define LFS_CONIC_FN __host__ __device__ __forceinline__
#else
#define LFS_CONIC_FN inline
#endif
#endif

namespace lfs {

struct ConicRec { float px, py, a, b, d, e, g, ia; }; // ia = -1 / a

LFS_CONIC_FN ConicRec conic_never() { return ConicRec{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, INFINITY, 0.f}; }
LFS_CONIC_FN ConicRec conic_always() { return ConicRec{3.0e38f, 3.0e38f, -1.f, 0.f, 0.f, 0.f, -1.f, 1.f}; }

// pc: Gaussian centre in camera space; A = Rc R diag(scale) (row-major, Sigma_cam = A A^T); opac: the activated opacity
LFS_CONIC_FN ConicRec conic_record(const float pc[3], const float A[3][3], const float opac) {
#if defined(__clang__)
#pragma clang fp contract(off) // the record is built in two translation units with different contraction defaults (raster.hip, projection_ut.hip): same bits in both
#endif
    if (opac < (1.f / 255.f)) return conic_always();
    const float Szz = A[2][0] * A[2][0] + A[2][1] * A[2][1] + A[2][2] * A[2][2];
    const float r2 = fmaxf(0.f, 2.f * logf(255.f * opac)) * 1.02f + 0.02f; // safety margin on the radius
    // the ellipsoid has to stay clear of the camera plane (z extent sqrt(r2 Szz) < p.z / 1.22): bounded silhouette, camera outside
    if (!(pc[2] > 0.f && pc[2] * pc[2] > 1.5f * r2 * Szz)) return conic_never();
    const float inv = 1.f / pc[2];
    const float px = pc[0] * inv, py = pc[1] * inv;
    float An[3][3], bx[3], by[3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) An[r][c] = A[r][c] * inv;
    for (int c = 0; c < 3; ++c) { bx[c] = An[0][c] - px * An[2][c]; by[c] = An[1][c] - py * An[2][c]; }
    // |A^T (p x d)|^2 = |du by - dv bx|^2
    const float Buu = by[0] * by[0] + by[1] * by[1] + by[2] * by[2];
    const float Buv = -(bx[0] * by[0] + bx[1] * by[1] + bx[2] * by[2]);
    const float Bvv = bx[0] * bx[0] + bx[1] * bx[1] + bx[2] * bx[2];
    float C[3][3]; // cofactors: column j = cross product of the other two columns of An
    for (int j = 0; j < 3; ++j) {
        const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        C[0][j] = An[1][j1] * An[2][j2] - An[2][j1] * An[1][j2];
        C[1][j] = An[2][j1] * An[0][j2] - An[0][j1] * An[2][j2];
        C[2][j] = An[0][j1] * An[1][j2] - An[1][j1] * An[0][j2];
    }
    float Ctp[3]; // C^T (px, py, 1)
    for (int j = 0; j < 3; ++j) Ctp[j] = C[0][j] * px + C[1][j] * py + C[2][j];
    const float Gpx = C[0][0] * Ctp[0] + C[0][1] * Ctp[1] + C[0][2] * Ctp[2];
    const float Gpy = C[1][0] * Ctp[0] + C[1][1] * Ctp[1] + C[1][2] * Ctp[2];
    const float c0 = Ctp[0] * Ctp[0] + Ctp[1] * Ctp[1] + Ctp[2] * Ctp[2];
    const float Gxx = C[0][0] * C[0][0] + C[0][1] * C[0][1] + C[0][2] * C[0][2];
    const float Gxy = C[0][0] * C[1][0] + C[0][1] * C[1][1] + C[0][2] * C[1][2];
    const float Gyy = C[1][0] * C[1][0] + C[1][1] * C[1][1] + C[1][2] * C[1][2];
    const float quu = r2 * Gxx - Buu, quv = r2 * Gxy - Buv, qvv = r2 * Gyy - Bvv;
    if (!(quu < 0.f && qvv < 0.f)) return conic_never();
    const float n = -1.f / qvv;
    ConicRec k{px, py, quu * n, quv * n, r2 * Gpx * n, r2 * Gpy * n, r2 * c0 * n, 0.f};
    k.ia = -1.f / k.a;
    const float chk = k.px + k.py + k.a + k.b + k.d + k.e + k.g + k.ia; // any inf / NaN poisons the sum
    if (!(chk - chk == 0.f)) return conic_never();
    return k;
}

// true when no ray (u, v, 1) with u in [u0, u1], v in [v0, v1] can composite the Gaussian (up to the rounding of the evaluation, which
// counts as "can": tol = 8e-6 of the magnitude of the terms)
LFS_CONIC_FN bool conic_culled(const ConicRec& k, const float u0, const float u1, const float v0, const float v1) {
    const float U0 = u0 - k.px, U1 = u1 - k.px, V0 = v0 - k.py, V1 = v1 - k.py;
    const bool centre_in = U0 <= 0.f && U1 >= 0.f && V0 <= 0.f && V1 >= 0.f;
    // du = U, dv in [V0, V1]:   -dv^2 + 2 (b U + e) dv + (a U^2 + 2 d U + g)
    const float l0 = k.b * U0 + k.e, l1 = k.b * U1 + k.e;
    const float w0 = fminf(fmaxf(l0, V0), V1), w1 = fminf(fmaxf(l1, V0), V1);
    const float eu0 = (2.f * l0 - w0) * w0 + ((k.a * U0 + 2.f * k.d) * U0 + k.g);
    const float eu1 = (2.f * l1 - w1) * w1 + ((k.a * U1 + 2.f * k.d) * U1 + k.g);
    // dv = V, du in [U0, U1]:   a du^2 + 2 (b V + d) du + (-V^2 + 2 e V + g)
    const float m0 = k.b * V0 + k.d, m1 = k.b * V1 + k.d;
    const float x0 = fminf(fmaxf(m0 * k.ia, U0), U1), x1 = fminf(fmaxf(m1 * k.ia, U0), U1);
    const float ev0 = (k.a * x0 + 2.f * m0) * x0 + ((2.f * k.e - V0) * V0 + k.g);
    const float ev1 = (k.a * x1 + 2.f * m1) * x1 + ((2.f * k.e - V1) * V1 + k.g);
    const float best = fmaxf(fmaxf(eu0, eu1), fmaxf(ev0, ev1));
    const float R2 = fmaxf(U0 * U0, U1 * U1) + fmaxf(V0 * V0, V1 * V1);
    const float tol = 8e-6f * (fabsf(k.a) + 2.f * fabsf(k.b) + 1.f) * R2;
    return !centre_in && best <= -tol; // (a NaN - inf - inf for rays within 1e-19 rad of the camera plane - compares false: not culled)
}

} // namespace lfs
