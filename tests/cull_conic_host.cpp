// Host-side brute-force check of lichtfeld-studio_amd/csrc/lfs_cull_conic.cuh (the code the rasterizer's cull kernel runs): for random and
// adversarial Gaussians (needles down to 1e-4, screen-filling, sub-pixel, far, right in front of the camera plane, low opacity) and 8x8 /
// 16x8 pixel cells around and away from their projection, a culled cell must contain NO pixel-centre ray that reaches alpha >= 1/255
// (double-precision reference). Also reports how many of the cullable cells were culled (effectiveness; the plane test this replaced
// reached ~95 % on the same distribution). Built and run by tests/test_cull_conic.py:  g++ -O2 -std=c++17 cull_conic_host.cpp
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include "../lichtfeld-studio_amd/csrc/lfs_cull_conic.cuh"

static void rotmat(const double q[4], double R[3][3]) {
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
    const double r[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)},
                            {2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)},
                            {2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i][j] = r[i][j];
}

// double-precision truth: min over the ray of the Mahalanobis distance^2 <= 2 ln(255 opac) for some pixel centre of the cell?
static bool visible(const double p[3], const double A[3][3], double opac, double x0, double y0, int nx, int ny, double fx) {
    if (opac < 1.0 / 255.0) return false;
    const double r2 = 2.0 * std::log(255.0 * opac);
    // D^2 = |A^-1 p|^2 - (A^-1 d . A^-1 p)^2 / |A^-1 d|^2 ; A^-1 = diag(1/s) Rfull^T: solve through the adjugate
    double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) + A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    double Ai[3][3];
    Ai[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) / det; Ai[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det; Ai[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det;
    Ai[1][0] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) / det; Ai[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det; Ai[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det;
    Ai[2][0] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) / det; Ai[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det; Ai[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det;
    double gp[3];
    for (int i = 0; i < 3; ++i) gp[i] = Ai[i][0] * p[0] + Ai[i][1] * p[1] + Ai[i][2] * p[2];
    const double pp = gp[0] * gp[0] + gp[1] * gp[1] + gp[2] * gp[2];
    for (int iy = 0; iy < ny; ++iy)
        for (int ix = 0; ix < nx; ++ix) {
            const double d[3] = {(x0 + ix + 0.5) / fx, (y0 + iy + 0.5) / fx, 1.0};
            double gd[3];
            for (int i = 0; i < 3; ++i) gd[i] = Ai[i][0] * d[0] + Ai[i][1] * d[1] + Ai[i][2] * d[2];
            const double dd = gd[0] * gd[0] + gd[1] * gd[1] + gd[2] * gd[2], dp = gd[0] * gp[0] + gd[1] * gp[1] + gd[2] * gp[2];
            if (pp - dp * dp / dd <= r2) return true;
        }
    return false;
}

int main(int argc, char** argv) {
    const long N = argc > 1 ? atol(argv[1]) : 200000;
    std::mt19937_64 rng(argc > 2 ? (uint64_t)atoll(argv[2]) : 1);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::normal_distribution<double> G(0.0, 1.0);
    auto logu = [&](double lo, double hi) { return std::exp(std::log(lo) + U(rng) * (std::log(hi) - std::log(lo))); };
    const double fx = 1200.0;
    const double opacs[] = {0.003, 0.0045, 0.01, 0.05, 0.3, 0.9, 0.999};
    long cells = 0, vis = 0, cullable = 0, culled = 0, false_culls = 0, never = 0;
    for (long it = 0; it < N; ++it) {
        const int kind = int(it % 8); // 0,6 normal  1,7 needles  2 huge  3 tiny  4 far  5 near
        double z = 0.3 + U(rng) * 19.7;
        if (kind == 4) z = 50 + U(rng) * 450;
        if (kind == 5) z = 0.05 + U(rng) * 0.55;
        const double p[3] = {(2 * U(rng) - 1) * z * 0.9, (1.2 * U(rng) - 0.6) * z * 0.9, z};
        double s[3];
        for (int k = 0; k < 3; ++k) {
            if (kind == 1 || kind == 7) s[k] = logu(1e-4, 0.5);
            else if (kind == 2) s[k] = logu(0.2, 5.0);
            else if (kind == 3) s[k] = logu(1e-5, 1e-3);
            else s[k] = std::exp(std::log(0.02) + 0.6 * G(rng));
        }
        const double q[4] = {G(rng), G(rng), G(rng), G(rng)};
        double R[3][3], A[3][3];
        rotmat(q, R);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = R[i][j] * s[j];
        const double opac = opacs[rng() % 7];
        float pf[3] = {(float)p[0], (float)p[1], (float)p[2]}, Af[3][3];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Af[i][j] = (float)A[i][j];
        // the double-precision reference sees exactly the float inputs the record was built from
        const double pd[3] = {pf[0], pf[1], pf[2]};
        double Ad[3][3];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Ad[i][j] = Af[i][j];
        const lfs::ConicRec rec = lfs::conic_record(pf, Af, (float)opac);
        if (std::isinf(rec.g)) ++never;
        const double cu = p[0] / p[2] * fx, cv = p[1] / p[2] * fx;
        for (int c = 0; c < 14; ++c) {
            const int wide = c & 1; // 8x8 and 16x8 cells
            long ox = c < 11 ? (long)(rng() % 11) - 5 : (long)(rng() % 200) - 100, oy = c < 11 ? (long)(rng() % 11) - 5 : (long)(rng() % 120) - 60;
            const double x0 = std::floor(cu / 8) * 8 + ox * 8, y0 = std::floor(cv / 8) * 8 + oy * 8;
            const int nx = wide ? 16 : 8, ny = 8;
            // the kernel's box: extreme pixel-centre rays +- a quarter pixel
            const float u0 = (float)((x0 + 0.5) / fx - 0.25 / fx), u1 = (float)((x0 + nx - 0.5) / fx + 0.25 / fx);
            const float v0 = (float)((y0 + 0.5) / fx - 0.25 / fx), v1 = (float)((y0 + ny - 0.5) / fx + 0.25 / fx);
            const bool cul = lfs::conic_culled(rec, u0, u1, v0, v1);
            const bool v = visible(pd, Ad, (double)(float)opac, x0, y0, nx, ny, fx);
            ++cells; vis += v; cullable += !v; culled += cul;
            if (cul && v) {
                ++false_culls;
                if (false_culls <= 5) printf("FALSE CULL kind %d p %.6g %.6g %.6g s %.3g %.3g %.3g opac %.4g cell %ld %ld wide %d\n", kind, p[0], p[1], p[2], s[0], s[1], s[2], opac, ox, oy, wide);
            }
        }
    }
    printf("{\"cells\": %ld, \"visible\": %ld, \"cullable\": %ld, \"culled\": %ld, \"false_culls\": %ld, \"never_cull_records\": %ld}\n", cells, vis, cullable, culled, false_culls, never);
    return false_culls ? 1 : 0;
}
