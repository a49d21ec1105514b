// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).
// CPU restatement of the camera models, shutter-pose interpolation and the
// unscented transform of /root/reference/gsplat/Cameras.cuh. One run-time
// dispatched `Camera<T>` replaces the reference's CRTP class family.
#pragma once
#include "oracle_math.hpp"
#include <algorithm>
#include <array>

namespace orc {

// gsplat/Common.h:46-50
enum CameraModel : int { PINHOLE = 0, ORTHO = 1, FISHEYE = 2 };
// gsplat/Cameras.h:16-22
enum Shutter : int { ROLLING_TOP_TO_BOTTOM = 0, ROLLING_LEFT_TO_RIGHT = 1, ROLLING_BOTTOM_TO_TOP = 2, ROLLING_RIGHT_TO_LEFT = 3, GLOBAL = 4 };

// gsplat/Cameras.h:27-61
template <class T> struct UTParams {
    T alpha = T(0.1f), beta = T(2), kappa = T(0), in_image_margin_factor = T(0.1f);
    bool require_all_sigma_points_valid = true;
};

// Cameras.cuh:33-71. viewmat is row-major [4,4] world->camera.
template <class T> struct RSParams {
    V3<T> t_start, t_end;
    Q4<T> q_start, q_end;
    RSParams(const T* se3_start, const T* se3_end) {
        auto load = [](const T* m, Q4<T>& q, V3<T>& t) {
            M3<T> R;
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R.m[r][c] = m[r * 4 + c];
            q = qcast(R);
            t = {m[3], m[7], m[11]};
        };
        load(se3_start, q_start, t_start);
        if (se3_end) load(se3_end, q_end, t_end); else { q_end = q_start; t_end = t_start; }
    }
};

template <class T> struct ShutterPose {
    V3<T> t; Q4<T> q;
};
// Cameras.cuh:268-280
template <class T> inline ShutterPose<T> interpolate_shutter_pose(T rel, const RSParams<T>& rs) {
    V3<T> t = (T(1) - rel) * rs.t_start + rel * rs.t_end;
    return {t, qslerp(rs.q_start, rs.q_end, rel)};
}

// Cameras.cuh:77-89
template <class T> inline T stable_norm2(T x, T y) {
    T ax = std::fabs(x), ay = std::fabs(y);
    T mn = std::fmin(ax, ay), mx = std::fmax(ax, ay);
    if (mx <= T(0)) return T(0);
    T r = mn / mx;
    return mx * std::sqrt(T(1) + r * r);
}
// Cameras.cuh:91-107 (Horner, highest coefficient first)
template <class T, size_t N> inline T poly(const std::array<T, N>& c, T x) {
    T y = T(0);
    for (size_t i = N; i-- > 0;) y = x * y + c[i];
    return y;
}
template <class T, size_t N> inline T poly_odd(const std::array<T, N>& c, T x) { return x * poly(c, x * x); }
template <class T, size_t N> inline T poly_even(const std::array<T, N>& c, T x) { return poly(c, x * x); }

// Cameras.cuh:759-815  solve 1 + a x + b x^2 + c x^3 = 0 (smallest positive root)
template <class T> inline T fisheye_max_angle(T a, T b, T c) {
    const T INF = std::numeric_limits<T>::max();
    const T PI_ = T(3.14159265358979323846);
    if (c == T(0)) {
        if (b == T(0)) return a >= T(0) ? INF : T(-1) / a;
        T delta = a * a - T(4) * b;
        if (delta >= T(0)) {
            delta = std::sqrt(delta) - a;
            if (delta > T(0)) return T(2) / delta;
        }
    } else {
        T boc = b / c, boc2 = boc * boc;
        T t1 = (T(9) * a * boc - T(2) * b * boc2 - T(27)) / c;
        T t2 = T(3) * a / c - boc2;
        T delta = t1 * t1 + T(4) * t2 * t2 * t2;
        if (delta >= T(0)) {
            T d2 = std::sqrt(delta);
            T cr = std::cbrt((d2 + t1) / T(2));
            if (cr != T(0)) {
                T s = (cr - (t2 / cr) - boc) / T(3);
                if (s > T(0)) return s;
            }
        } else {
            T theta = std::atan2(std::sqrt(-delta), t1) / T(3);
            const T two_third_pi = T(2) * PI_ / T(3);
            T t3 = T(2) * std::sqrt(-t2);
            T soln = INF;
            for (int i : {-1, 0, 1}) {
                T s = (t3 * std::cos(theta + T(i) * two_third_pi) - boc) / T(3);
                if (s > T(0)) soln = std::min(soln, s);
            }
            return soln;
        }
    }
    return INF;
}

template <class T> struct ImagePoint { V2<T> p; bool valid; };
template <class T> struct Ray { V3<T> o, d; bool valid; };

template <class T> struct Camera {
    int model = PINHOLE;     // PINHOLE / FISHEYE
    bool distorted = false;  // PINHOLE with any coefficient tensor -> OpenCV model
    uint32_t width = 0, height = 0;
    int shutter = GLOBAL;
    T fx = 1, fy = 1, cx = 0, cy = 0;
    std::array<T, 6> radial{};     // OpenCV k1..k6 ; fisheye uses [0..3]
    std::array<T, 2> tangential{};
    std::array<T, 4> thin_prism{};
    // fisheye derived state, Cameras.cuh:830-885
    std::array<T, 5> fwd_odd{}, dfwd_even{};
    std::array<T, 2> approx_back{};
    T max_angle = 0;
    T min_2d_norm = T(1e-6f);

    void init_fisheye() {
        T k1 = radial[0], k2 = radial[1], k3 = radial[2], k4 = radial[3];
        fwd_odd = {T(1), k1, k2, k3, k4};
        dfwd_even = {T(1), T(3) * k1, T(5) * k2, T(7) * k3, T(9) * k4};
        T mdx = std::max(T(width) - cx, cx), mdy = std::max(T(height) - cy, cy);
        T max_r = std::sqrt(mdx * mdx + mdy * mdy);
        if (k4 == T(0)) {
            max_angle = std::sqrt(fisheye_max_angle(T(3) * k1, T(5) * k2, T(7) * k3));
        } else {
            // Newton on the derivative polynomial, start 1.57 (Cameras.cuh:857-871, 169-206)
            std::array<T, 4> dd_odd = {T(6) * k1, T(20) * k2, T(42) * k3, T(72) * k4};
            T x = T(1.57f);
            bool conv = false;
            for (int j = 0; j < 20; ++j) {
                T dfdx = poly_odd(dd_odd, x);
                T res = poly_even(dfwd_even, x) - T(0);
                T dx = res / dfdx;
                x -= dx;
                if (std::fabs(dx) < T(1e-6f)) { conv = true; break; }
            }
            max_angle = (!conv || x <= T(0)) ? std::numeric_limits<T>::max() : x;
        }
        max_angle = std::min(max_angle, std::max(max_r / fx, max_r / fy));
        T mnd = std::max(T(width) / T(2) / fx, T(height) / T(2) / fy);
        approx_back = {T(0), max_angle / mnd};
    }

    // Cameras.cuh:228-240
    bool in_bounds(V2<T> p, T margin) const {
        T mx = T(width) * margin, my = T(height) * margin;
        bool v = true;
        v &= (-mx) <= p.x && p.x < (T(width) + mx);
        v &= (-my) <= p.y && p.y < (T(height) + my);
        return v;
    }

    // Cameras.cuh:504-533
    void distortion(V2<T> uv, T& icD, V2<T>& delta, T& r2) const {
        T u2 = uv.x * uv.x, v2 = uv.y * uv.y;
        r2 = u2 + v2;
        T a1 = T(2) * uv.x * uv.y, a2 = r2 + T(2) * u2, a3 = r2 + T(2) * v2;
        T num = T(1) + r2 * (radial[0] + r2 * (radial[1] + r2 * radial[2]));
        T den = T(1) + r2 * (radial[3] + r2 * (radial[4] + r2 * radial[5]));
        icD = num / den;
        delta.x = tangential[0] * a1 + tangential[1] * a2 + r2 * (thin_prism[0] + r2 * thin_prism[1]);
        delta.y = tangential[0] * a3 + tangential[1] * a1 + r2 * (thin_prism[2] + r2 * thin_prism[3]);
    }

    // camera_ray_to_image_point: pinhole :431-455, OpenCV :535-597, fisheye :894-959
    ImagePoint<T> project(V3<T> c, T margin) const {
        if (c.z <= T(0)) return {{T(0), T(0)}, false};
        if (model == FISHEYE) {
            T n = stable_norm2(c.x, c.y);
            if (n <= T(0)) n = std::numeric_limits<float>::epsilon();
            T theta_full = std::atan2(n, c.z);
            T theta = theta_full < max_angle ? theta_full : max_angle;
            T delta = poly_odd(fwd_odd, theta) / n;
            if (delta <= T(0)) return {{T(0), T(0)}, false};
            V2<T> p{fx * delta * c.x + cx, fy * delta * c.y + cy};
            bool v = in_bounds(p, margin);
            v &= theta <= max_angle;
            return {p, v};
        }
        V2<T> uv{c.x / c.z, c.y / c.z};
        if (!distorted) {
            V2<T> p{uv.x * fx + cx, uv.y * fy + cy};
            return {p, in_bounds(p, margin)};
        }
        T icD, r2; V2<T> d;
        distortion(uv, icD, d, r2);
        bool v = icD > T(0.8f);
        V2<T> nd{icD * uv.x + d.x, icD * uv.y + d.y};
        V2<T> p{nd.x * fx + cx, nd.y * fy + cy};
        v &= in_bounds(p, margin);
        return {p, v};
    }

    // Cameras.cuh:635-696 + 698-740 (Newton undistortion, 5 iterations max)
    V2<T> undistort_newton(V2<T> ip, bool& converged) const {
        const T k1 = radial[0], k2 = radial[1], k3 = radial[2], k4 = radial[3], k5 = radial[4], k6 = radial[5];
        const T p1 = tangential[0], p2 = tangential[1];
        const T s1 = thin_prism[0], s2 = thin_prism[1], s3 = thin_prism[2], s4 = thin_prism[3];
        T xd = (ip.x - cx) / fx, yd = (ip.y - cy) / fy;
        T x = xd, y = yd;
        const T eps = T(1e-6f);
        converged = false;
        for (int it = 0; it < 5; ++it) {
            T r = x * x + y * y, r2 = r * r;
            T alpha = T(1) + r * (k1 + r * (k2 + r * k3));
            T beta = T(1) + r * (k4 + r * (k5 + r * k6));
            T d = alpha / beta;
            if (d <= T(0)) break;
            T fx_ = d * x + T(2) * p1 * x * y + p2 * (r + T(2) * x * x) + s1 * r + s2 * r2 - xd;
            T fy_ = d * y + T(2) * p2 * x * y + p1 * (r + T(2) * y * y) + s3 * r + s4 * r2 - yd;
            T alpha_r = k1 + r * (T(2) * k2 + r * (T(3) * k3));
            T beta_r = k4 + r * (T(2) * k5 + r * (T(3) * k6));
            T d_r = (alpha_r * beta - alpha * beta_r) / (beta * beta);
            T d_x = T(2) * x * d_r, d_y = T(2) * y * d_r;
            T fx_x = d + d_x * x + T(2) * p1 * y + T(6) * p2 * x; fx_x += T(2) * x * (s1 + T(2) * s2 * r);
            T fx_y = d_y * x + T(2) * p1 * x + T(2) * p2 * y;     fx_y += T(2) * y * (s1 + T(2) * s2 * r);
            T fy_x = d_x * y + T(2) * p2 * y + T(2) * p1 * x;     fy_x += T(2) * x * (s3 + T(2) * s4 * r);
            T fy_y = d + d_y * y + T(2) * p2 * x + T(6) * p1 * y; fy_y += T(2) * y * (s3 + T(2) * s4 * r);
            T det = fx_y * fy_x - fx_x * fy_y;
            if (std::fabs(det) < eps) break;
            T dx = (fx_ * fy_y - fy_ * fx_y) / det;
            T dy = (fy_ * fx_x - fx_ * fy_x) / det;
            x += dx; y += dy;
            if (std::fabs(dx) < eps && std::fabs(dy) < eps) { converged = true; break; }
        }
        return {x, y};
    }

    // image_point_to_camera_ray: pinhole :457-470, OpenCV :742-754, fisheye :961-1000
    bool unproject(V2<T> ip, V3<T>& dir) const {
        if (model == FISHEYE) {
            V2<T> uv{(ip.x - cx) / fx, (ip.y - cy) / fy};
            T delta = std::sqrt(uv.x * uv.x + uv.y * uv.y);
            // eval_poly_inverse_horner_newton<20> (Cameras.cuh:169-206)
            T th = poly(approx_back, delta);
            bool conv = false;
            for (int j = 0; j < 20; ++j) {
                T dfdx = poly_even(dfwd_even, th);
                T res = poly_odd(fwd_odd, th) - delta;
                T dx = res / dfdx;
                th -= dx;
                if (std::fabs(dx) < T(1e-6f)) { conv = true; break; }
            }
            if (th < T(0) || th >= max_angle || !conv) { dir = {T(0), T(0), T(1)}; return false; }
            if (delta >= min_2d_norm) {
                T sf = std::sin(th) / delta;
                dir = {sf * uv.x, sf * uv.y, std::cos(th)};
            } else {
                dir = {T(0), T(0), T(1)};
            }
            return true;
        }
        bool valid = true;
        V2<T> uv;
        if (!distorted) uv = {(ip.x - cx) / fx, (ip.y - cy) / fy};
        else uv = undistort_newton(ip, valid);
        V3<T> r{uv.x, uv.y, T(1)};
        T len = std::sqrt(dot(r, r));
        dir = {r.x / len, r.y / len, r.z / len};
        return valid;
    }

    // Cameras.cuh:293-320
    T relative_frame_time(V2<T> ip) const {
        T t = T(0);
        switch (shutter) {
        case ROLLING_TOP_TO_BOTTOM: t = std::floor(ip.y) / T(height - 1); break;
        case ROLLING_LEFT_TO_RIGHT: t = std::floor(ip.x) / T(width - 1); break;
        case ROLLING_BOTTOM_TO_TOP: t = (T(height) - std::ceil(ip.y)) / T(height - 1); break;
        case ROLLING_RIGHT_TO_LEFT: t = (T(width) - std::ceil(ip.x)) / T(width - 1); break;
        default: break;
        }
        return t;
    }

    // Cameras.cuh:322-339 + 257-265
    Ray<T> pixel_ray(V2<T> ip, const RSParams<T>& rs) const {
        V3<T> cd;
        if (!unproject(ip, cd)) return {{T(0), T(0), T(0)}, {T(0), T(0), T(0)}, false};
        ShutterPose<T> pose = interpolate_shutter_pose(relative_frame_time(ip), rs);
        M3<T> Rinv = qmat3(qinverse(pose.q));
        return {-mul(Rinv, pose.t), mul(Rinv, cd), true};
    }

    // Cameras.cuh:346-413 (N_ROLLING_SHUTTER_ITERATIONS = 10)
    ImagePoint<T> world_to_image(V3<T> wp, const RSParams<T>& rs, T margin) const {
        ImagePoint<T> s = project(qrotate(rs.q_start, wp) + rs.t_start, margin);
        if (shutter == GLOBAL) return s;
        ImagePoint<T> e = project(qrotate(rs.q_end, wp) + rs.t_end, margin);
        V2<T> prev;
        if (s.valid) prev = s.p; else if (e.valid) prev = e.p; else return {e.p, false};
        for (int j = 0; j < 10; ++j) {
            T rel = relative_frame_time(prev);
            V3<T> t = (T(1) - rel) * rs.t_start + rel * rs.t_end;
            Q4<T> q = qslerp(rs.q_start, rs.q_end, rel);
            prev = project(qrotate(q, wp) + t, margin).p;
        }
        return {prev, true};
    }
};

// Build the Camera the way ProjectionUT3DGSFused.cu:84-131 / Fwd.cu:88-134 do.
// n_radial says how many radial coefficients the caller's tensor really holds
// (the reference reads 6 unconditionally for PINHOLE).
template <class T> inline Camera<T> make_camera(
    int model, uint32_t W, uint32_t H, int shutter, const T* K,
    const T* radial, int n_radial, const T* tangential, const T* thin_prism, int n_thin) {
    Camera<T> cam;
    cam.model = model; cam.width = W; cam.height = H; cam.shutter = shutter;
    cam.fx = K[0]; cam.fy = K[4]; cam.cx = K[2]; cam.cy = K[5];
    if (model == FISHEYE) {
        if (radial) for (int i = 0; i < 4 && i < n_radial; ++i) cam.radial[i] = radial[i];
        cam.init_fisheye();
    } else {
        cam.distorted = radial || tangential || thin_prism;
        if (radial) for (int i = 0; i < 6 && i < n_radial; ++i) cam.radial[i] = radial[i];
        if (tangential) { cam.tangential[0] = tangential[0]; cam.tangential[1] = tangential[1]; }
        if (thin_prism) for (int i = 0; i < 4 && i < n_thin; ++i) cam.thin_prism[i] = thin_prism[i];
    }
    return cam;
}

// Cameras.cuh:1034-1150: sigma points + unscented transform.
template <class T> struct ImageGaussian { V2<T> mean; M2<T> cov; bool valid; };

template <class T> inline ImageGaussian<T> unscented_transform(
    const Camera<T>& cam, const RSParams<T>& rs, const UTParams<T>& ut,
    V3<T> mean, V3<T> scale, Q4<T> rot) {
    const T D = T(3);
    const T lambda = ut.alpha * ut.alpha * (D + ut.kappa) - D;
    M3<T> R = qmat3(rot);
    V3<T> pts[7];
    T wm[7], wc[7];
    pts[0] = mean;
    const T sc[3] = {scale.x, scale.y, scale.z};
    for (int i = 0; i < 3; ++i) {
        T f = std::sqrt(D + lambda) * sc[i];
        V3<T> delta{f * R.m[0][i], f * R.m[1][i], f * R.m[2][i]}; // column i of R
        pts[i + 1] = mean + delta;
        pts[i + 4] = mean - delta;
    }
    wm[0] = lambda / (D + lambda);
    wc[0] = lambda / (D + lambda) + (T(1) - ut.alpha * ut.alpha + ut.beta);
    for (int i = 0; i < 6; ++i) { wm[i + 1] = T(1) / (T(2) * (D + lambda)); wc[i + 1] = wm[i + 1]; }

    bool valid = ut.require_all_sigma_points_valid;
    V2<T> ip[7];
    V2<T> m{T(0), T(0)};
    M2<T> cov{{{T(0), T(0)}, {T(0), T(0)}}};
    for (int i = 0; i < 7; ++i) {
        ImagePoint<T> r = cam.world_to_image(pts[i], rs, ut.in_image_margin_factor);
        if (ut.require_all_sigma_points_valid) {
            valid &= r.valid;
            if (!r.valid) return {m, cov, false};
        } else {
            valid |= r.valid;
        }
        ip[i] = r.p;
        m = m + wm[i] * ip[i];
    }
    if (!valid) return {m, cov, false};
    for (int i = 0; i < 7; ++i) {
        V2<T> d = ip[i] - m;
        cov.m[0][0] += wc[i] * (d.x * d.x);
        cov.m[0][1] += wc[i] * (d.x * d.y);
        cov.m[1][0] += wc[i] * (d.y * d.x);
        cov.m[1][1] += wc[i] * (d.y * d.y);
    }
    return {m, cov, valid};
}

} // namespace orc
