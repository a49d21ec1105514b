// ORACLE — TEST INFRASTRUCTURE ONLY. CPU restatement of the reference's default ("fastgs") EWA training rasterizer
// (SURVEY.md §8f row 1): fastgs/rasterization/include/kernels_forward.cuh:19-205 (preprocess), :207-330 (instances),
// :353-459 (blend); kernel_utils.cuh:15-148 (SH colour, exact tile test); kernels_backward.cuh:19-233 (preprocess
// backward), :236-448 (blend backward); constants rasterization_config.h:14-33.
// PINNED to the reference's own fastgs code: forward.cu / backward.cu and the kernels of its headers are run on the CPU (oracle/ref_kernels_fastgs.cpp,
// `make -C oracle refk_fastgs`), their outputs are committed as tests/golden/refk_fastgs.npz, and tests/test_oracle_refk_fastgs_golden.py holds this restatement
// to them (counts identical, image 3e-7, gradients 1e-4 relative L2). The EWA covariance / SH colour are also cross-checked against the reference's
// tests/torch_impl.cpp in tests/test_oracle_fastgs.py.
// Ordering inside a tile: ascending (depth bits, primitive index) — the reference sorts by depth only and breaks ties by the
// arrival order of an atomicAdd (kernels_forward.cuh:200-203), i.e. not deterministically.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace orc {
namespace fg {

constexpr int TILE = 16;
template <class T> struct Consts {
    static constexpr T dilation = T(0.3f);
    static constexpr T min_alpha_rcp = T(255.0f);
    static constexpr T min_alpha = T(1.0f) / T(255.0f);
    static constexpr T max_alpha = T(0.999f);
    static constexpr T t_threshold = T(1e-4f);
};

template <class T> struct Args {
    int64_t N; const T* means; const T* scales_raw; const T* rot_raw; const T* opac_raw; const T* sh0; const T* sh_rest;
    const T* w2c; const T* cam_pos; int active_sh_bases; int total_rest; int W, H; T fx, fy, cx, cy, near_, far_;
};

template <class T> struct Prim {
    bool visible = false; T mx = 0, my = 0, ca = 0, cb = 0, cc = 0, opacity = 0, col[3] = {0, 0, 0}; float depth = 0.f;
    uint32_t x0 = 0, x1 = 0, y0 = 0, y1 = 0; uint32_t n_touched = 0;
};

// kernel_utils.cuh:15-36 / :38-106
template <class T> void sh_color(const Args<T>& a, int64_t i, T* out) {
    const T* c0 = a.sh0 + 3 * i;
    const T* cr = a.sh_rest + size_t(i) * a.total_rest * 3;
    T r[3] = {T(0.5f) + T(0.28209479177387814) * c0[0], T(0.5f) + T(0.28209479177387814) * c0[1], T(0.5f) + T(0.28209479177387814) * c0[2]};
    if (a.active_sh_bases > 1) {
        T dx = a.means[3 * i] - a.cam_pos[0], dy = a.means[3 * i + 1] - a.cam_pos[1], dz = a.means[3 * i + 2] - a.cam_pos[2];
        const T inv = T(1) / std::sqrt(dx * dx + dy * dy + dz * dz);
        const T x = dx * inv, y = dy * inv, z = dz * inv;
        auto add = [&](T w, int k) { for (int c = 0; c < 3; ++c) r[c] += w * cr[3 * k + c]; };
        add(T(-0.48860251190291987) * y, 0); add(T(0.48860251190291987) * z, 1); add(T(-0.48860251190291987) * x, 2);
        if (a.active_sh_bases > 4) {
            const T xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
            add(T(1.0925484305920792) * xy, 3); add(T(-1.0925484305920792) * yz, 4); add(T(0.94617469575755997) * zz - T(0.31539156525251999), 5);
            add(T(-1.0925484305920792) * xz, 6); add(T(0.54627421529603959) * xx - T(0.54627421529603959) * yy, 7);
            if (a.active_sh_bases > 9) {
                add(T(0.59004358992664352) * y * (T(-3) * xx + yy), 8); add(T(2.8906114426405538) * xy * z, 9);
                add(T(0.45704579946446572) * y * (T(1) - T(5) * zz), 10); add(T(0.3731763325901154) * z * (T(5) * zz - T(3)), 11);
                add(T(0.45704579946446572) * x * (T(1) - T(5) * zz), 12); add(T(1.4453057213202769) * z * (xx - yy), 13);
                add(T(0.59004358992664352) * x * (-xx + T(3) * yy), 14);
            }
        }
    }
    out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
}

// kernel_utils.cuh:108-148 (mean already shifted by -0.5; rect = pixel INDEX range of the cell)
template <class T> bool will_contribute(T mx, T my, T ca, T cb, T cc, T rx0, T ry0, T w, T h, T power_threshold) {
    const T rx1 = rx0 + w - 1, ry1 = ry0 + h - 1;
    const T x_min_diff = rx0 - mx, y_min_diff = ry0 - my;
    const T x_left = x_min_diff > 0 ? T(1) : T(0), y_above = y_min_diff > 0 ? T(1) : T(0);
    const T not_in_x = x_left + (mx > rx1 ? T(1) : T(0)), not_in_y = y_above + (my > ry1 ? T(1) : T(0));
    if (not_in_x + not_in_y == T(0)) return true;
    const T ccx = x_left > 0 ? rx0 : rx1, ccy = y_above > 0 ? ry0 : ry1; // closest corner
    const T dfx = mx - ccx, dfy = my - ccy;
    const T dx = std::copysign(w - 1, x_min_diff), dy = std::copysign(h - 1, y_min_diff);
    auto sat = [](T v) { return v != v ? T(0) : std::min(std::max(v, T(0)), T(1)); };
    const T tx = not_in_y * sat((dx * ca * dfx + dx * cb * dfy) / (dx * ca * dx));
    const T ty = not_in_x * sat((dy * cb * dfx + dy * cc * dfy) / (dy * cc * dy));
    const T px = ccx + tx * dx, py = ccy + ty * dy;
    const T ddx = mx - px, ddy = my - py;
    const T power = T(0.5f) * (ca * ddx * ddx + cc * ddy * ddy) + cb * ddx * ddy;
    return power <= power_threshold;
}

struct Cov3 { double dummy; };

// kernels_forward.cuh:19-205
template <class T> void preprocess(const Args<T>& a, std::vector<Prim<T>>& P) {
    using K = Consts<T>;
    const uint32_t gw = (a.W + TILE - 1) / TILE, gh = (a.H + TILE - 1) / TILE;
    P.assign(a.N, Prim<T>());
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < a.N; ++i) {
        Prim<T>& p = P[i];
        const T* m = a.means + 3 * i;
        const T* r1 = a.w2c; const T* r2 = a.w2c + 4; const T* r3 = a.w2c + 8;
        const T depth = r3[0] * m[0] + r3[1] * m[1] + r3[2] * m[2] + r3[3];
        if (depth < a.near_ || depth > a.far_) continue;
        const T opacity = T(1) / (T(1) + std::exp(-a.opac_raw[i]));
        if (opacity < K::min_alpha) continue;
        const T* rs = a.scales_raw + 3 * i;
        const T var[3] = {std::exp(T(2) * rs[0]), std::exp(T(2) * rs[1]), std::exp(T(2) * rs[2])};
        const T qr = a.rot_raw[4 * i], qx = a.rot_raw[4 * i + 1], qy = a.rot_raw[4 * i + 2], qz = a.rot_raw[4 * i + 3];
        const T qn = qr * qr + qx * qx + qy * qy + qz * qz;
        if (qn < T(1e-8f)) continue;
        const T qxx = T(2) * qx * qx / qn, qyy = T(2) * qy * qy / qn, qzz = T(2) * qz * qz / qn;
        const T qxy = T(2) * qx * qy / qn, qxz = T(2) * qx * qz / qn, qyz = T(2) * qy * qz / qn;
        const T qrx = T(2) * qr * qx / qn, qry = T(2) * qr * qy / qn, qrz = T(2) * qr * qz / qn;
        const T R[3][3] = {{T(1) - (qyy + qzz), qxy - qrz, qry + qxz}, {qrz + qxy, T(1) - (qxx + qzz), qyz - qrx}, {qxz - qry, qrx + qyz, T(1) - (qxx + qyy)}};
        T cov[3][3];
        for (int u = 0; u < 3; ++u) for (int v = 0; v < 3; ++v) cov[u][v] = R[u][0] * var[0] * R[v][0] + R[u][1] * var[1] * R[v][1] + R[u][2] * var[2] * R[v][2];
        const T x = (r1[0] * m[0] + r1[1] * m[1] + r1[2] * m[2] + r1[3]) / depth;
        const T y = (r2[0] * m[0] + r2[1] * m[1] + r2[2] * m[2] + r2[3]) / depth;
        const T w = T(a.W), h = T(a.H);
        const T tx = std::min(std::max(x, (T(-0.15f) * w - a.cx) / a.fx), (T(1.15f) * w - a.cx) / a.fx);
        const T ty = std::min(std::max(y, (T(-0.15f) * h - a.cy) / a.fy), (T(1.15f) * h - a.cy) / a.fy);
        const T j11 = a.fx / depth, j13 = -j11 * tx, j22 = a.fy / depth, j23 = -j22 * ty;
        const T jw1[3] = {j11 * r1[0] + j13 * r3[0], j11 * r1[1] + j13 * r3[1], j11 * r1[2] + j13 * r3[2]};
        const T jw2[3] = {j22 * r2[0] + j23 * r3[0], j22 * r2[1] + j23 * r3[1], j22 * r2[2] + j23 * r3[2]};
        T jc1[3], jc2[3];
        for (int v = 0; v < 3; ++v) { jc1[v] = jw1[0] * cov[0][v] + jw1[1] * cov[1][v] + jw1[2] * cov[2][v]; jc2[v] = jw2[0] * cov[0][v] + jw2[1] * cov[1][v] + jw2[2] * cov[2][v]; }
        const T c2a = jc1[0] * jw1[0] + jc1[1] * jw1[1] + jc1[2] * jw1[2] + K::dilation;
        const T c2b = jc1[0] * jw2[0] + jc1[1] * jw2[1] + jc1[2] * jw2[2];
        const T c2c = jc2[0] * jw2[0] + jc2[1] * jw2[1] + jc2[2] * jw2[2] + K::dilation;
        const T det = c2a * c2c - c2b * c2b;
        if (det < T(1e-8f)) continue;
        p.ca = c2c / det; p.cb = -c2b / det; p.cc = c2a / det;
        p.mx = x * a.fx + a.cx; p.my = y * a.fy + a.cy;
        const T power_threshold = std::log(opacity * K::min_alpha_rcp);
        const T f = std::sqrt(T(2) * power_threshold);
        const T ex = std::max(f * std::sqrt(c2a) - T(0.5f), T(0)), ey = std::max(f * std::sqrt(c2c) - T(0.5f), T(0));
        auto rd = [](T v) { return int64_t(std::floor(v)); };
        auto ru = [](T v) { return int64_t(std::ceil(v)); };
        p.x0 = uint32_t(std::min<int64_t>(gw, std::max<int64_t>(0, rd((p.mx - ex) / T(TILE)))));
        p.x1 = uint32_t(std::min<int64_t>(gw, std::max<int64_t>(0, ru((p.mx + ex) / T(TILE)))));
        p.y0 = uint32_t(std::min<int64_t>(gh, std::max<int64_t>(0, rd((p.my - ey) / T(TILE)))));
        p.y1 = uint32_t(std::min<int64_t>(gh, std::max<int64_t>(0, ru((p.my + ey) / T(TILE)))));
        if ((p.x1 - p.x0) * (p.y1 - p.y0) == 0) continue;
        uint32_t cnt = 0;
        for (uint32_t tyi = p.y0; tyi < p.y1; ++tyi) for (uint32_t txi = p.x0; txi < p.x1; ++txi)
            if (will_contribute<T>(p.mx - T(0.5f), p.my - T(0.5f), p.ca, p.cb, p.cc, T(txi * TILE), T(tyi * TILE), T(TILE), T(TILE), power_threshold)) ++cnt;
        if (cnt == 0) continue;
        p.n_touched = cnt; p.opacity = opacity; p.depth = float(depth); p.visible = true;
        sh_color<T>(a, i, p.col);
    }
}

struct Lists { std::vector<int32_t> offsets; std::vector<int32_t> ids; }; // offsets [tiles+1]

template <class T> void build_lists(const Args<T>& a, const std::vector<Prim<T>>& P, Lists& L) {
    const uint32_t gw = (a.W + TILE - 1) / TILE, gh = (a.H + TILE - 1) / TILE;
    std::vector<std::vector<uint64_t>> per(gw * gh);
    for (int64_t i = 0; i < a.N; ++i) {
        const Prim<T>& p = P[i];
        if (!p.visible) continue;
        const T thr = std::log(p.opacity * Consts<T>::min_alpha_rcp);
        uint32_t bits; std::memcpy(&bits, &p.depth, 4);
        for (uint32_t ty = p.y0; ty < p.y1; ++ty) for (uint32_t tx = p.x0; tx < p.x1; ++tx)
            if (will_contribute<T>(p.mx - T(0.5f), p.my - T(0.5f), p.ca, p.cb, p.cc, T(tx * TILE), T(ty * TILE), T(TILE), T(TILE), thr))
                per[ty * gw + tx].push_back((uint64_t(bits) << 32) | uint32_t(i));
    }
    L.offsets.assign(gw * gh + 1, 0); L.ids.clear();
    for (size_t t = 0; t < per.size(); ++t) {
        std::sort(per[t].begin(), per[t].end());
        L.offsets[t] = int32_t(L.ids.size());
        for (uint64_t k : per[t]) L.ids.push_back(int32_t(uint32_t(k)));
    }
    L.offsets[per.size()] = int32_t(L.ids.size());
}

// kernels_forward.cuh:353-459. n_contrib[pixel] = 1 + list position (inside the tile) of the last compositing entry.
template <class T> void blend(const Args<T>& a, const std::vector<Prim<T>>& P, const Lists& L, T* image, T* alpha_map, int32_t* n_contrib) {
    using K = Consts<T>;
    const uint32_t gw = (a.W + TILE - 1) / TILE;
    const int64_t np = int64_t(a.W) * a.H;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t pix = 0; pix < np; ++pix) {
        const int py = int(pix / a.W), px = int(pix % a.W);
        const uint32_t t = (py / TILE) * gw + (px / TILE);
        const T fx_ = T(px) + T(0.5f), fy_ = T(py) + T(0.5f);
        T col[3] = {0, 0, 0}, tr = 1; int32_t nc = 0, possible = 0;
        for (int32_t e = L.offsets[t]; e < L.offsets[t + 1]; ++e) {
            ++possible;
            const Prim<T>& p = P[L.ids[e]];
            const T dx = p.mx - fx_, dy = p.my - fy_;
            const T s = T(0.5f) * (p.ca * dx * dx + p.cc * dy * dy) + p.cb * dx * dy;
            if (s < 0) continue;
            const T al = std::min(p.opacity * std::exp(-s), K::max_alpha);
            if (al < K::min_alpha) continue;
            const T nt = tr * (T(1) - al);
            if (nt < K::t_threshold) break;
            for (int c = 0; c < 3; ++c) col[c] += tr * al * std::max(p.col[c], T(0));
            tr = nt; nc = possible;
        }
        image[pix] = col[0]; image[np + pix] = col[1]; image[2 * np + pix] = col[2];
        alpha_map[pix] = T(1) - tr; n_contrib[pix] = nc;
    }
}

// kernels_backward.cuh:236-448 (per pixel, back to front; the reference walks front to back from per-bucket checkpoints — same
// sums) followed by :19-233 per primitive.
template <class T> void backward(const Args<T>& a, const std::vector<Prim<T>>& P, const Lists& L, const T* image, const T* alpha_map, const int32_t* n_contrib,
                                 const T* g_image, const T* g_alpha, T* g_means, T* g_scales_raw, T* g_rot_raw, T* g_opac_raw, T* g_sh0, T* g_sh_rest,
                                 T* densification_info) {
    using K = Consts<T>;
    const uint32_t gw = (a.W + TILE - 1) / TILE;
    const int64_t np = int64_t(a.W) * a.H, N = a.N;
    std::vector<double> gm2(2 * N, 0.0), gcon(3 * N, 0.0), gop(N, 0.0), gcol(3 * N, 0.0);
    (void)image;
    for (int64_t pix = 0; pix < np; ++pix) {
        const int py = int(pix / a.W), px = int(pix % a.W);
        const uint32_t t = (py / TILE) * gw + (px / TILE);
        const T fx_ = T(px) + T(0.5f), fy_ = T(py) + T(0.5f);
        const T gc[3] = {g_image[pix], g_image[np + pix], g_image[2 * np + pix]};
        const T t_final = T(1) - alpha_map[pix];
        const T g_alpha_common = g_alpha[pix] * t_final;
        T tr = t_final, after[3] = {0, 0, 0};
        for (int32_t e = L.offsets[t] + n_contrib[pix] - 1; e >= L.offsets[t]; --e) {
            const int32_t id = L.ids[e];
            const Prim<T>& p = P[id];
            const T dx = p.mx - fx_, dy = p.my - fy_;
            const T s = T(0.5f) * (p.ca * dx * dx + p.cc * dy * dy) + p.cb * dx * dy;
            if (s < 0) continue;
            const T al = std::min(p.opacity * std::exp(-s), K::max_alpha);
            if (al < K::min_alpha) continue;
            const T oma = T(1) - al, oma_rcp = T(1) / oma;
            tr = tr * oma_rcp; // transmittance in front of this entry
            const T wgt = tr * al;
            T dl_dalpha = g_alpha_common * oma_rcp;
            for (int c = 0; c < 3; ++c) {
                const T cc = std::max(p.col[c], T(0));
                gcol[3 * id + c] += double(wgt * gc[c] * (p.col[c] >= 0 ? T(1) : T(0)));
                dl_dalpha += (tr * cc - after[c] * oma_rcp) * gc[c];
                after[c] += wgt * cc;
            }
            gop[id] += double(al * dl_dalpha);
            const T helper = -al * dl_dalpha;
            gcon[3 * id] += double(T(0.5f) * helper * dx * dx); gcon[3 * id + 1] += double(T(0.5f) * helper * dx * dy); gcon[3 * id + 2] += double(T(0.5f) * helper * dy * dy);
            gm2[2 * id] += double(helper * (p.ca * dx + p.cb * dy)); gm2[2 * id + 1] += double(helper * (p.cb * dx + p.cc * dy));
        }
    }
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < N; ++i) {
        for (int c = 0; c < 3; ++c) { g_means[3 * i + c] = 0; g_scales_raw[3 * i + c] = 0; g_sh0[3 * i + c] = 0; }
        for (int c = 0; c < 4; ++c) g_rot_raw[4 * i + c] = 0;
        for (int c = 0; c < 3 * a.total_rest; ++c) g_sh_rest[size_t(i) * a.total_rest * 3 + c] = 0;
        g_opac_raw[i] = 0;
        const Prim<T>& p = P[i];
        if (!p.visible) continue;
        g_opac_raw[i] = T(gop[i]) * (T(1) - p.opacity);
        const T* m = a.means + 3 * i;
        // ---- SH backward (kernel_utils.cuh:38-106)
        const T gcl[3] = {T(gcol[3 * i]), T(gcol[3 * i + 1]), T(gcol[3 * i + 2])};
        for (int c = 0; c < 3; ++c) g_sh0[3 * i + c] = T(0.28209479177387814) * gcl[c];
        T dcol_dpos[3] = {0, 0, 0};
        if (a.active_sh_bases > 1) {
            const T* cr = a.sh_rest + size_t(i) * a.total_rest * 3;
            T* gr = g_sh_rest + size_t(i) * a.total_rest * 3;
            const T xr = m[0] - a.cam_pos[0], yr = m[1] - a.cam_pos[1], zr = m[2] - a.cam_pos[2];
            const T inv = T(1) / std::sqrt(xr * xr + yr * yr + zr * zr);
            const T x = xr * inv, y = yr * inv, z = zr * inv;
            T gdx[3], gdy[3], gdz[3];
            auto setg = [&](int k, T w) { for (int c = 0; c < 3; ++c) gr[3 * k + c] = w * gcl[c]; };
            setg(0, T(-0.48860251190291987) * y); setg(1, T(0.48860251190291987) * z); setg(2, T(-0.48860251190291987) * x);
            for (int c = 0; c < 3; ++c) { gdx[c] = T(-0.48860251190291987) * cr[6 + c]; gdy[c] = T(-0.48860251190291987) * cr[c]; gdz[c] = T(0.48860251190291987) * cr[3 + c]; }
            if (a.active_sh_bases > 4) {
                const T xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
                setg(3, T(1.0925484305920792) * xy); setg(4, T(-1.0925484305920792) * yz); setg(5, T(0.94617469575755997) * zz - T(0.31539156525251999));
                setg(6, T(-1.0925484305920792) * xz); setg(7, T(0.54627421529603959) * xx - T(0.54627421529603959) * yy);
                for (int c = 0; c < 3; ++c) {
                    gdx[c] += T(1.0925484305920792) * y * cr[9 + c] + T(-1.0925484305920792) * z * cr[18 + c] + T(1.0925484305920792) * x * cr[21 + c];
                    gdy[c] += T(1.0925484305920792) * x * cr[9 + c] + T(-1.0925484305920792) * z * cr[12 + c] + T(-1.0925484305920792) * y * cr[21 + c];
                    gdz[c] += T(-1.0925484305920792) * y * cr[12 + c] + T(1.8923493915151202) * z * cr[15 + c] + T(-1.0925484305920792) * x * cr[18 + c];
                }
                if (a.active_sh_bases > 9) {
                    setg(8, T(0.59004358992664352) * y * (T(-3) * xx + yy)); setg(9, T(2.8906114426405538) * xy * z);
                    setg(10, T(0.45704579946446572) * y * (T(1) - T(5) * zz)); setg(11, T(0.3731763325901154) * z * (T(5) * zz - T(3)));
                    setg(12, T(0.45704579946446572) * x * (T(1) - T(5) * zz)); setg(13, T(1.4453057213202769) * z * (xx - yy));
                    setg(14, T(0.59004358992664352) * x * (-xx + T(3) * yy));
                    for (int c = 0; c < 3; ++c) {
                        gdx[c] += T(-3.5402615395598609) * xy * cr[24 + c] + T(2.8906114426405538) * yz * cr[27 + c] + (T(0.45704579946446572) - T(2.2852289973223288) * zz) * cr[36 + c] +
                                  T(2.8906114426405538) * xz * cr[39 + c] + (T(-1.7701307697799304) * xx + T(1.7701307697799304) * yy) * cr[42 + c];
                        gdy[c] += (T(-1.7701307697799304) * xx + T(1.7701307697799304) * yy) * cr[24 + c] + T(2.8906114426405538) * xz * cr[27 + c] +
                                  (T(0.45704579946446572) - T(2.2852289973223288) * zz) * cr[30 + c] + T(-2.8906114426405538) * yz * cr[39 + c] + T(3.5402615395598609) * xy * cr[42 + c];
                        gdz[c] += T(2.8906114426405538) * xy * cr[27 + c] + T(-4.5704579946446566) * yz * cr[30 + c] + (T(5.597644988851731) * zz - T(1.1195289977703462)) * cr[33 + c] +
                                  T(-4.5704579946446566) * xz * cr[36 + c] + (T(1.4453057213202769) * xx - T(1.4453057213202769) * yy) * cr[39 + c];
                    }
                }
            }
            const T gd[3] = {gdx[0] * gcl[0] + gdx[1] * gcl[1] + gdx[2] * gcl[2], gdy[0] * gcl[0] + gdy[1] * gcl[1] + gdy[2] * gcl[2], gdz[0] * gcl[0] + gdz[1] * gcl[1] + gdz[2] * gcl[2]};
            const T xx = xr * xr, yy = yr * yr, zz = zr * zr, xy = xr * yr, xz = xr * zr, yz = yr * zr;
            const T n2 = xx + yy + zz, rs = T(1) / std::sqrt(n2 * n2 * n2);
            dcol_dpos[0] = ((yy + zz) * gd[0] - xy * gd[1] - xz * gd[2]) * rs;
            dcol_dpos[1] = (-xy * gd[0] + (xx + zz) * gd[1] - yz * gd[2]) * rs;
            dcol_dpos[2] = (-xz * gd[0] - yz * gd[1] + (xx + yy) * gd[2]) * rs;
        }
        // ---- EWA backward (kernels_backward.cuh:56-232)
        const T* r1 = a.w2c; const T* r2 = a.w2c + 4; const T* r3 = a.w2c + 8;
        const T depth = r3[0] * m[0] + r3[1] * m[1] + r3[2] * m[2] + r3[3];
        const T x = (r1[0] * m[0] + r1[1] * m[1] + r1[2] * m[2] + r1[3]) / depth;
        const T y = (r2[0] * m[0] + r2[1] * m[1] + r2[2] * m[2] + r2[3]) / depth;
        const T* rsr = a.scales_raw + 3 * i;
        const T var[3] = {std::exp(T(2) * rsr[0]), std::exp(T(2) * rsr[1]), std::exp(T(2) * rsr[2])};
        const T qr = a.rot_raw[4 * i], qx = a.rot_raw[4 * i + 1], qy = a.rot_raw[4 * i + 2], qz = a.rot_raw[4 * i + 3];
        const T qn = qr * qr + qx * qx + qy * qy + qz * qz;
        const T qxx = T(2) * qx * qx / qn, qyy = T(2) * qy * qy / qn, qzz = T(2) * qz * qz / qn;
        const T qxy = T(2) * qx * qy / qn, qxz = T(2) * qx * qz / qn, qyz = T(2) * qy * qz / qn;
        const T qrx = T(2) * qr * qx / qn, qry = T(2) * qr * qy / qn, qrz = T(2) * qr * qz / qn;
        const T R[3][3] = {{T(1) - (qyy + qzz), qxy - qrz, qry + qxz}, {qrz + qxy, T(1) - (qxx + qzz), qyz - qrx}, {qxz - qry, qrx + qyz, T(1) - (qxx + qyy)}};
        T RS[3][3], cov[3][3];
        for (int u = 0; u < 3; ++u) for (int v = 0; v < 3; ++v) RS[u][v] = R[u][v] * var[v];
        for (int u = 0; u < 3; ++u) for (int v = 0; v < 3; ++v) cov[u][v] = RS[u][0] * R[v][0] + RS[u][1] * R[v][1] + RS[u][2] * R[v][2];
        const T w = T(a.W), h = T(a.H);
        const T tx = std::min(std::max(x, (T(-0.15f) * w - a.cx) / a.fx), (T(1.15f) * w - a.cx) / a.fx);
        const T ty = std::min(std::max(y, (T(-0.15f) * h - a.cy) / a.fy), (T(1.15f) * h - a.cy) / a.fy);
        const T j11 = a.fx / depth, j13 = -j11 * tx, j22 = a.fy / depth, j23 = -j22 * ty;
        const T jw1[3] = {j11 * r1[0] + j13 * r3[0], j11 * r1[1] + j13 * r3[1], j11 * r1[2] + j13 * r3[2]};
        const T jw2[3] = {j22 * r2[0] + j23 * r3[0], j22 * r2[1] + j23 * r3[1], j22 * r2[2] + j23 * r3[2]};
        T jc1[3], jc2[3];
        for (int v = 0; v < 3; ++v) { jc1[v] = jw1[0] * cov[0][v] + jw1[1] * cov[1][v] + jw1[2] * cov[2][v]; jc2[v] = jw2[0] * cov[0][v] + jw2[1] * cov[1][v] + jw2[2] * cov[2][v]; }
        const T A = jc1[0] * jw1[0] + jc1[1] * jw1[1] + jc1[2] * jw1[2] + K::dilation, B = jc1[0] * jw2[0] + jc1[1] * jw2[1] + jc1[2] * jw2[2],
                Cc = jc2[0] * jw2[0] + jc2[1] * jw2[1] + jc2[2] * jw2[2] + K::dilation;
        const T det = A * Cc - B * B, dr = T(1) / det, dr2 = dr * dr;
        const T dcon[3] = {T(gcon[3 * i]), T(gcon[3 * i + 1]), T(gcon[3 * i + 2])};
        const T dcv[3] = {dr2 * (T(2) * B * Cc * dcon[1] - Cc * Cc * dcon[0] - B * B * dcon[2]),
                          dr2 * (B * Cc * dcon[0] - (A * Cc + B * B) * dcon[1] + A * B * dcon[2]),
                          dr2 * (T(2) * A * B * dcon[1] - B * B * dcon[0] - A * A * dcon[2])};
        T dcov3[3][3];
        for (int u = 0; u < 3; ++u) for (int v = 0; v < 3; ++v)
            dcov3[u][v] = (jw1[u] * jw1[v]) * dcv[0] + (u == v ? T(2) * jw1[u] * jw2[u] : jw1[u] * jw2[v] + jw1[v] * jw2[u]) * dcv[1] + (jw2[u] * jw2[v]) * dcv[2];
        T djw1[3], djw2[3];
        for (int v = 0; v < 3; ++v) { djw1[v] = T(2) * (jc1[v] * dcv[0] + jc2[v] * dcv[1]); djw2[v] = T(2) * (jc1[v] * dcv[1] + jc2[v] * dcv[2]); }
        const T dj11 = r1[0] * djw1[0] + r1[1] * djw1[1] + r1[2] * djw1[2], dj22 = r2[0] * djw2[0] + r2[1] * djw2[1] + r2[2] * djw2[2];
        const T dj13 = r3[0] * djw1[0] + r3[1] * djw1[1] + r3[2] * djw1[2], dj23 = r3[0] * djw2[0] + r3[1] * djw2[1] + r3[2] * djw2[2];
        const T h1 = dj11 - T(2) * tx * dj13, h2 = dj22 - T(2) * ty * dj23;
        const T dm2[2] = {T(gm2[2 * i]), T(gm2[2 * i + 1])};
        const T dcam[3] = {j11 * (dm2[0] - dj13 / depth), j22 * (dm2[1] - dj23 / depth), -j11 * (x * dm2[0] + h1 / depth) - j22 * (y * dm2[1] + h2 / depth)};
        for (int c = 0; c < 3; ++c) g_means[3 * i + c] = r1[c] * dcam[0] + r2[c] * dcam[1] + r3[c] * dcam[2] + dcol_dpos[c];
        // upper-triangular convention of the reference: off-diagonal entries of dL_dcov3d are used with a factor 2
        for (int v = 0; v < 3; ++v) {
            const T dvar = R[0][v] * R[0][v] * dcov3[0][0] + R[1][v] * R[1][v] * dcov3[1][1] + R[2][v] * R[2][v] * dcov3[2][2] +
                           T(2) * (R[0][v] * R[1][v] * dcov3[0][1] + R[0][v] * R[2][v] * dcov3[0][2] + R[1][v] * R[2][v] * dcov3[1][2]);
            g_scales_raw[3 * i + v] = T(2) * var[v] * dvar;
        }
        T dR[3][3];
        for (int u = 0; u < 3; ++u) for (int v = 0; v < 3; ++v) dR[u][v] = T(2) * (RS[0][v] * dcov3[u][0] + RS[1][v] * dcov3[u][1] + RS[2][v] * dcov3[u][2]);
        const T dqxx = -dR[1][1] - dR[2][2], dqyy = -dR[0][0] - dR[2][2], dqzz = -dR[0][0] - dR[1][1];
        const T dqxy = dR[0][1] + dR[1][0], dqxz = dR[0][2] + dR[2][0], dqyz = dR[1][2] + dR[2][1];
        const T dqrx = dR[2][1] - dR[1][2], dqry = dR[0][2] - dR[2][0], dqrz = dR[1][0] - dR[0][1];
        const T hn = qxx * dqxx + qyy * dqyy + qzz * dqzz + qxy * dqxy + qxz * dqxz + qyz * dqyz + qrx * dqrx + qry * dqry + qrz * dqrz;
        g_rot_raw[4 * i] = T(2) * (qx * dqrx + qy * dqry + qz * dqrz - qr * hn) / qn;
        g_rot_raw[4 * i + 1] = T(2) * (T(2) * qx * dqxx + qy * dqxy + qz * dqxz + qr * dqrx - qx * hn) / qn;
        g_rot_raw[4 * i + 2] = T(2) * (T(2) * qy * dqyy + qx * dqxy + qz * dqyz + qr * dqry - qy * hn) / qn;
        g_rot_raw[4 * i + 3] = T(2) * (T(2) * qz * dqzz + qx * dqxz + qy * dqyz + qr * dqrz - qz * hn) / qn;
        if (densification_info) {
            densification_info[i] += T(1);
            densification_info[N + i] += std::sqrt((dm2[0] * T(0.5f) * w) * (dm2[0] * T(0.5f) * w) + (dm2[1] * T(0.5f) * h) * (dm2[1] * T(0.5f) * h));
        }
    }
}

} // namespace fg
} // namespace orc
