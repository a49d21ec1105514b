// Device camera models for the gfx950 kernels: perfect pinhole, OpenCV pinhole
// (radial/tangential/thin-prism, Newton undistortion), OpenCV fisheye, global and
// rolling shutter, and the 7-sigma-point unscented transform.
// Behavioural spec: /root/reference/gsplat/Cameras.cuh (:228-240 bounds margin,
// :268-280 shutter pose, :293-320 frame time, :346-413 RS projection, :416-471
// pinhole, :473-755 OpenCV, :759-1001 fisheye, :1034-1150 UT). One run-time
// dispatched struct instead of the reference's CRTP family: the model branch is
// wave-uniform, so it costs a scalar compare.
#pragma once
#include "lfs_math.cuh"
#include "../../include/lfs_gsplat.h"

namespace lfs {

struct CamDev {
    int32_t model, distorted, shutter;
    uint32_t width, height;
    float fx, fy, cx, cy;
    float radial[6];
    float tangential[2];
    float thin[4];
    quat q_start, q_end;
    f3 t_start, t_end;
    // fisheye derived state
    float fwd_odd[5], dfwd_even[5], approx_back[2];
    float max_angle;
    // global-shutter constants (valid when shutter == GLOBAL): world ray = (origin, Rinv * d_cam)
    m3 Rinv;
    f3 origin;
};

LFS_DI float poly5(const float* c, float x) { // c0 + c1 x + ... + c4 x^4, Horner from the top
    float y = 0.f;
#pragma unroll
    for (int i = 4; i >= 0; --i) y = x * y + c[i];
    return y;
}

LFS_DI float stable_norm2(float x, float y) {
    float ax = fabsf(x), ay = fabsf(y);
    float mn = fminf(ax, ay), mx = fmaxf(ax, ay);
    if (mx <= 0.f) return 0.f;
    float r = mn / mx;
    return mx * sqrtf(1.f + r * r);
}

// smallest positive root of 1 + a x + b x^2 + c x^3 (fisheye FOV limit)
LFS_DI float fisheye_max_angle(float a, float b, float c) {
    const float INF = 3.402823466e+38f;
    const float PI_ = 3.14159265358979323846f;
    if (c == 0.f) {
        if (b == 0.f) return a >= 0.f ? INF : -1.f / a;
        float delta = a * a - 4.f * b;
        if (delta >= 0.f) {
            delta = sqrtf(delta) - a;
            if (delta > 0.f) return 2.f / delta;
        }
    } else {
        float boc = b / c, boc2 = boc * boc;
        float t1 = (9.f * a * boc - 2.f * b * boc2 - 27.f) / c;
        float t2 = 3.f * a / c - boc2;
        float delta = t1 * t1 + 4.f * t2 * t2 * t2;
        if (delta >= 0.f) {
            float d2 = sqrtf(delta);
            float cr = cbrtf((d2 + t1) / 2.f);
            if (cr != 0.f) {
                float s = (cr - (t2 / cr) - boc) / 3.f;
                if (s > 0.f) return s;
            }
        } else {
            float theta = atan2f(sqrtf(-delta), t1) / 3.f;
            const float two_third_pi = 2.f * PI_ / 3.f;
            float t3 = 2.f * sqrtf(-t2);
            float soln = INF;
#pragma unroll
            for (int i = -1; i <= 1; ++i) {
                float s = (t3 * cosf(theta + float(i) * two_third_pi) - boc) / 3.f;
                if (s > 0.f) soln = fminf(soln, s);
            }
            return soln;
        }
    }
    return INF;
}

LFS_DI void cam_load_pose(const float* __restrict__ vm, quat& q, f3& t) {
    q = qcast_viewmat(vm);
    t = {vm[3], vm[7], vm[11]};
}

// Build the per-camera state from the API-level camera block.
LFS_DI void cam_init(CamDev& cam, const lfs_cameras& in, uint32_t cid) {
    cam.model = in.camera_model;
    cam.shutter = in.rs_type;
    cam.width = in.image_width; cam.height = in.image_height;
    const float* K = in.Ks + 9 * cid;
    cam.fx = K[0]; cam.fy = K[4]; cam.cx = K[2]; cam.cy = K[5];
#pragma unroll
    for (int i = 0; i < 6; ++i) cam.radial[i] = 0.f;
    cam.tangential[0] = cam.tangential[1] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) cam.thin[i] = 0.f;
    const int nr_max = (in.camera_model == LFS_CAMERA_FISHEYE) ? 4 : 6;
    const int rstride = (in.camera_model == LFS_CAMERA_FISHEYE) ? 4 : in.n_radial;
    // (fully unrolled with predicates: a run-time trip count would force the arrays of CamDev into scratch memory)
    if (in.radial_coeffs) {
#pragma unroll
        for (int i = 0; i < 6; ++i) if (i < nr_max && i < in.n_radial) cam.radial[i] = in.radial_coeffs[rstride * cid + i];
    }
    if (in.tangential_coeffs) { cam.tangential[0] = in.tangential_coeffs[2 * cid]; cam.tangential[1] = in.tangential_coeffs[2 * cid + 1]; }
    if (in.thin_prism_coeffs) {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i < in.n_thin_prism) cam.thin[i] = in.thin_prism_coeffs[in.n_thin_prism * cid + i];
    }
    cam.distorted = (in.camera_model != LFS_CAMERA_FISHEYE) &&
                    (in.radial_coeffs != nullptr || in.tangential_coeffs != nullptr || in.thin_prism_coeffs != nullptr);
    cam_load_pose(in.viewmats0 + 16 * cid, cam.q_start, cam.t_start);
    if (in.viewmats1) cam_load_pose(in.viewmats1 + 16 * cid, cam.q_end, cam.t_end);
    else { cam.q_end = cam.q_start; cam.t_end = cam.t_start; }
    cam.max_angle = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) { cam.fwd_odd[i] = 0.f; cam.dfwd_even[i] = 0.f; }
    cam.approx_back[0] = cam.approx_back[1] = 0.f;
    if (in.camera_model == LFS_CAMERA_FISHEYE) {
        const float k1 = cam.radial[0], k2 = cam.radial[1], k3 = cam.radial[2], k4 = cam.radial[3];
        cam.fwd_odd[0] = 1.f; cam.fwd_odd[1] = k1; cam.fwd_odd[2] = k2; cam.fwd_odd[3] = k3; cam.fwd_odd[4] = k4;
        cam.dfwd_even[0] = 1.f; cam.dfwd_even[1] = 3.f * k1; cam.dfwd_even[2] = 5.f * k2; cam.dfwd_even[3] = 7.f * k3; cam.dfwd_even[4] = 9.f * k4;
        float mdx = fmaxf(float(cam.width) - cam.cx, cam.cx), mdy = fmaxf(float(cam.height) - cam.cy, cam.cy);
        float max_r = sqrtf(mdx * mdx + mdy * mdy);
        float ma;
        if (k4 == 0.f) {
            ma = sqrtf(fisheye_max_angle(3.f * k1, 5.f * k2, 7.f * k3));
        } else {
            const float dd[5] = {6.f * k1, 20.f * k2, 42.f * k3, 72.f * k4, 0.f};
            float x = 1.57f; bool conv = false;
            for (int j = 0; j < 20; ++j) {
                float x2 = x * x;
                float dfdx = x * poly5(dd, x2);
                float res = poly5(cam.dfwd_even, x2);
                float dx = res / dfdx;
                x -= dx;
                if (fabsf(dx) < 1e-6f) { conv = true; break; }
            }
            ma = (!conv || x <= 0.f) ? 3.402823466e+38f : x;
        }
        cam.max_angle = fminf(ma, fmaxf(max_r / cam.fx, max_r / cam.fy));
        float mnd = fmaxf(float(cam.width) / 2.f / cam.fx, float(cam.height) / 2.f / cam.fy);
        cam.approx_back[0] = 0.f; cam.approx_back[1] = cam.max_angle / mnd;
    }
    // global-shutter ray constants: pose at relative frame time 0 == start pose
    cam.Rinv = qmat3(qinverse(cam.q_start));
    cam.origin = -mul(cam.Rinv, cam.t_start);
}

LFS_DI bool cam_in_bounds(const CamDev& cam, f2 p, float margin) {
    float mx = float(cam.width) * margin, my = float(cam.height) * margin;
    bool v = (-mx) <= p.x && p.x < (float(cam.width) + mx);
    v &= (-my) <= p.y && p.y < (float(cam.height) + my);
    return v;
}

LFS_DI void cam_distortion(const CamDev& cam, f2 uv, float& icD, f2& delta) {
    float u2 = uv.x * uv.x, v2 = uv.y * uv.y;
    float r2 = u2 + v2;
    float a1 = 2.f * uv.x * uv.y, a2 = r2 + 2.f * u2, a3 = r2 + 2.f * v2;
    float num = 1.f + r2 * (cam.radial[0] + r2 * (cam.radial[1] + r2 * cam.radial[2]));
    float den = 1.f + r2 * (cam.radial[3] + r2 * (cam.radial[4] + r2 * cam.radial[5]));
    icD = num / den;
    delta.x = cam.tangential[0] * a1 + cam.tangential[1] * a2 + r2 * (cam.thin[0] + r2 * cam.thin[1]);
    delta.y = cam.tangential[0] * a3 + cam.tangential[1] * a1 + r2 * (cam.thin[2] + r2 * cam.thin[3]);
}

// camera-space point -> image point (+ validity incl. the in-image margin)
LFS_DI bool cam_project(const CamDev& cam, f3 c, float margin, f2& p) {
    p = {0.f, 0.f};
    if (c.z <= 0.f) return false;
    if (cam.model == LFS_CAMERA_FISHEYE) {
        float n = stable_norm2(c.x, c.y);
        if (n <= 0.f) n = 1.1920928955078125e-07f;
        float theta_full = atan2f(n, c.z);
        float theta = theta_full < cam.max_angle ? theta_full : cam.max_angle;
        float delta = theta * poly5(cam.fwd_odd, theta * theta) / n;
        if (delta <= 0.f) return false;
        p = {cam.fx * delta * c.x + cam.cx, cam.fy * delta * c.y + cam.cy};
        bool v = cam_in_bounds(cam, p, margin);
        v &= theta <= cam.max_angle;
        return v;
    }
    f2 uv{c.x / c.z, c.y / c.z};
    if (!cam.distorted) {
        p = {uv.x * cam.fx + cam.cx, uv.y * cam.fy + cam.cy};
        return cam_in_bounds(cam, p, margin);
    }
    float icD; f2 d;
    cam_distortion(cam, uv, icD, d);
    bool v = icD > 0.8f;
    f2 nd{icD * uv.x + d.x, icD * uv.y + d.y};
    p = {nd.x * cam.fx + cam.cx, nd.y * cam.fy + cam.cy};
    v &= cam_in_bounds(cam, p, margin);
    return v;
}

LFS_DI f2 cam_undistort_newton(const CamDev& cam, f2 ip, bool& converged) {
    const float k1 = cam.radial[0], k2 = cam.radial[1], k3 = cam.radial[2], k4 = cam.radial[3], k5 = cam.radial[4], k6 = cam.radial[5];
    const float p1 = cam.tangential[0], p2 = cam.tangential[1];
    const float s1 = cam.thin[0], s2 = cam.thin[1], s3 = cam.thin[2], s4 = cam.thin[3];
    const float xd = (ip.x - cam.cx) / cam.fx, yd = (ip.y - cam.cy) / cam.fy;
    float x = xd, y = yd;
    const float eps = 1e-6f;
    converged = false;
    for (int it = 0; it < 5; ++it) {
        float r = x * x + y * y, r2 = r * r;
        float alpha = 1.f + r * (k1 + r * (k2 + r * k3));
        float beta = 1.f + r * (k4 + r * (k5 + r * k6));
        float d = alpha / beta;
        if (d <= 0.f) break;
        float fx_ = d * x + 2.f * p1 * x * y + p2 * (r + 2.f * x * x) + s1 * r + s2 * r2 - xd;
        float fy_ = d * y + 2.f * p2 * x * y + p1 * (r + 2.f * y * y) + s3 * r + s4 * r2 - yd;
        float alpha_r = k1 + r * (2.f * k2 + r * (3.f * k3));
        float beta_r = k4 + r * (2.f * k5 + r * (3.f * k6));
        float d_r = (alpha_r * beta - alpha * beta_r) / (beta * beta);
        float d_x = 2.f * x * d_r, d_y = 2.f * y * d_r;
        float fx_x = d + d_x * x + 2.f * p1 * y + 6.f * p2 * x; fx_x += 2.f * x * (s1 + 2.f * s2 * r);
        float fx_y = d_y * x + 2.f * p1 * x + 2.f * p2 * y;     fx_y += 2.f * y * (s1 + 2.f * s2 * r);
        float fy_x = d_x * y + 2.f * p2 * y + 2.f * p1 * x;     fy_x += 2.f * x * (s3 + 2.f * s4 * r);
        float fy_y = d + d_y * y + 2.f * p2 * x + 6.f * p1 * y; fy_y += 2.f * y * (s3 + 2.f * s4 * r);
        float det = fx_y * fy_x - fx_x * fy_y;
        if (fabsf(det) < eps) break;
        float dx = (fx_ * fy_y - fy_ * fx_y) / det;
        float dy = (fy_ * fx_x - fx_ * fy_x) / det;
        x += dx; y += dy;
        if (fabsf(dx) < eps && fabsf(dy) < eps) { converged = true; break; }
    }
    return {x, y};
}

// image point -> normalised camera-space direction
LFS_DI bool cam_unproject(const CamDev& cam, f2 ip, f3& dir) {
    if (cam.model == LFS_CAMERA_FISHEYE) {
        f2 uv{(ip.x - cam.cx) / cam.fx, (ip.y - cam.cy) / cam.fy};
        float delta = sqrtf(uv.x * uv.x + uv.y * uv.y);
        float th = delta * cam.approx_back[1] + cam.approx_back[0];
        bool conv = false;
        for (int j = 0; j < 20; ++j) {
            float t2 = th * th;
            float dfdx = poly5(cam.dfwd_even, t2);
            float res = th * poly5(cam.fwd_odd, t2) - delta;
            float dx = res / dfdx;
            th -= dx;
            if (fabsf(dx) < 1e-6f) { conv = true; break; }
        }
        if (th < 0.f || th >= cam.max_angle || !conv) { dir = {0.f, 0.f, 1.f}; return false; }
        if (delta >= 1e-6f) {
            float sf = sinf(th) / delta;
            dir = {sf * uv.x, sf * uv.y, cosf(th)};
        } else {
            dir = {0.f, 0.f, 1.f};
        }
        return true;
    }
    bool valid = true;
    f2 uv;
    if (!cam.distorted) uv = {(ip.x - cam.cx) / cam.fx, (ip.y - cam.cy) / cam.fy};
    else uv = cam_undistort_newton(cam, ip, valid);
    float len = sqrtf(uv.x * uv.x + uv.y * uv.y + 1.f);
    dir = {uv.x / len, uv.y / len, 1.f / len};
    return valid;
}

LFS_DI float cam_relative_frame_time(const CamDev& cam, f2 ip) {
    switch (cam.shutter) {
    case LFS_SHUTTER_ROLLING_TOP_TO_BOTTOM: return floorf(ip.y) / float(cam.height - 1);
    case LFS_SHUTTER_ROLLING_LEFT_TO_RIGHT: return floorf(ip.x) / float(cam.width - 1);
    case LFS_SHUTTER_ROLLING_BOTTOM_TO_TOP: return (float(cam.height) - ceilf(ip.y)) / float(cam.height - 1);
    case LFS_SHUTTER_ROLLING_RIGHT_TO_LEFT: return (float(cam.width) - ceilf(ip.x)) / float(cam.width - 1);
    default: return 0.f;
    }
}

LFS_DI void cam_pose_at(const CamDev& cam, float rel, quat& q, f3& t) {
    t = (1.f - rel) * cam.t_start + rel * cam.t_end;
    q = qslerp(cam.q_start, cam.q_end, rel);
}

// pixel centre -> world-space ray
LFS_DI bool cam_pixel_ray(const CamDev& cam, f2 ip, f3& o, f3& d) {
    f3 cd;
    if (!cam_unproject(cam, ip, cd)) { o = {0.f, 0.f, 0.f}; d = {0.f, 0.f, 0.f}; return false; }
    if (cam.shutter == LFS_SHUTTER_GLOBAL) {
        o = cam.origin; d = mul(cam.Rinv, cd);
        return true;
    }
    quat q; f3 t;
    cam_pose_at(cam, cam_relative_frame_time(cam, ip), q, t);
    m3 Rinv = qmat3(qinverse(q));
    o = -mul(Rinv, t); d = mul(Rinv, cd);
    return true;
}

// world point -> image point under the (rolling) shutter model
LFS_DI bool cam_world_to_image(const CamDev& cam, f3 wp, float margin, f2& p) {
    bool vs = cam_project(cam, qrotate(cam.q_start, wp) + cam.t_start, margin, p);
    if (cam.shutter == LFS_SHUTTER_GLOBAL) return vs;
    f2 pe;
    bool ve = cam_project(cam, qrotate(cam.q_end, wp) + cam.t_end, margin, pe);
    f2 prev;
    if (vs) prev = p; else if (ve) prev = pe; else { p = pe; return false; }
    for (int j = 0; j < 10; ++j) {
        quat q; f3 t;
        cam_pose_at(cam, cam_relative_frame_time(cam, prev), q, t);
        cam_project(cam, qrotate(q, wp) + t, margin, prev);
    }
    p = prev;
    return true;
}

} // namespace lfs
