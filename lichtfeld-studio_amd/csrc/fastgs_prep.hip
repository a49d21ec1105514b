// The reference's default training rasterizer ("fastgs", EWA splatting) — per-primitive stages:
// fused activation + covariance + EWA projection + SH colour + exact tile counting (replaces preprocess_cu,
// fastgs/rasterization/include/kernels_forward.cuh:19-205), instance emission (create_instances_cu :224-330, here a
// scatter straight into per-tile buckets, as intersect.hip does) and the per-primitive backward (preprocess_backward_cu,
// kernels_backward.cuh:19-233, incl. convert_sh_to_color_backward kernel_utils.cuh:38-106 and densification_info).
// Streaming, HBM-bound kernels: compiled with -ffp-contract=off so that the tile-membership test rounds like the oracle.
#include "lfs_fastgs.cuh"
#include "lfs_prof.h"
#include "lfs_tilelists.cuh"
#include "../../include/lfs_gsplat.h"

namespace lfs {
namespace fgs {

// kernel_utils.cuh:108-148 — does the maximum of the Gaussian over the pixel-index rectangle [rx0, rx0+w-1] x [ry0, ry0+h-1]
// reach the power threshold? (mean already shifted by -0.5)
LFS_DI bool will_contribute(float mx, float my, float ca, float cb, float cc, float rx0, float ry0, float w, float h, float power_threshold) {
    const float rx1 = rx0 + w - 1.f, ry1 = ry0 + h - 1.f;
    const float x_min_diff = rx0 - mx, y_min_diff = ry0 - my;
    const float x_left = x_min_diff > 0.f ? 1.f : 0.f, y_above = y_min_diff > 0.f ? 1.f : 0.f;
    const float not_in_x = x_left + (mx > rx1 ? 1.f : 0.f), not_in_y = y_above + (my > ry1 ? 1.f : 0.f);
    if (not_in_x + not_in_y == 0.f) return true;
    const float ccx = x_left > 0.f ? rx0 : rx1, ccy = y_above > 0.f ? ry0 : ry1;
    const float dfx = mx - ccx, dfy = my - ccy;
    const float dx = copysignf(w - 1.f, x_min_diff), dy = copysignf(h - 1.f, y_min_diff);
    const float tx = not_in_y * __saturatef((dx * ca * dfx + dx * cb * dfy) / (dx * ca * dx));
    const float ty = not_in_x * __saturatef((dy * cb * dfx + dy * cc * dfy) / (dy * cc * dy));
    const float px = ccx + tx * dx, py = ccy + ty * dy;
    const float ddx = mx - px, ddy = my - py;
    return 0.5f * (ca * ddx * ddx + cc * ddy * ddy) + cb * ddx * ddy <= power_threshold;
}

struct Cov { // what forward and backward both need of a primitive's geometry
    float depth, x, y, var[3], R[3][3], RS[3][3], cov[3][3], qr, qx, qy, qz, qn;
    float qxx, qyy, qzz, qxy, qxz, qyz, qrx, qry, qrz;
    float tx, ty, j11, j13, j22, j23, jw1[3], jw2[3], jc1[3], jc2[3], a, b, c;
};
LFS_DI void ewa(const Frame& f, const float* __restrict__ m, const float* __restrict__ rs, const float4 q, Cov& o) {
    const float* r1 = f.w2c; const float* r2 = f.w2c + 4; const float* r3 = f.w2c + 8;
    o.depth = r3[0] * m[0] + r3[1] * m[1] + r3[2] * m[2] + r3[3];
    o.x = (r1[0] * m[0] + r1[1] * m[1] + r1[2] * m[2] + r1[3]) / o.depth;
    o.y = (r2[0] * m[0] + r2[1] * m[1] + r2[2] * m[2] + r2[3]) / o.depth;
    o.var[0] = expf(2.f * rs[0]); o.var[1] = expf(2.f * rs[1]); o.var[2] = expf(2.f * rs[2]);
    o.qr = q.x; o.qx = q.y; o.qy = q.z; o.qz = q.w;
    o.qn = o.qr * o.qr + o.qx * o.qx + o.qy * o.qy + o.qz * o.qz;
    o.qxx = 2.f * o.qx * o.qx / o.qn; o.qyy = 2.f * o.qy * o.qy / o.qn; o.qzz = 2.f * o.qz * o.qz / o.qn;
    o.qxy = 2.f * o.qx * o.qy / o.qn; o.qxz = 2.f * o.qx * o.qz / o.qn; o.qyz = 2.f * o.qy * o.qz / o.qn;
    o.qrx = 2.f * o.qr * o.qx / o.qn; o.qry = 2.f * o.qr * o.qy / o.qn; o.qrz = 2.f * o.qr * o.qz / o.qn;
    o.R[0][0] = 1.f - (o.qyy + o.qzz); o.R[0][1] = o.qxy - o.qrz; o.R[0][2] = o.qry + o.qxz;
    o.R[1][0] = o.qrz + o.qxy; o.R[1][1] = 1.f - (o.qxx + o.qzz); o.R[1][2] = o.qyz - o.qrx;
    o.R[2][0] = o.qxz - o.qry; o.R[2][1] = o.qrx + o.qyz; o.R[2][2] = 1.f - (o.qxx + o.qyy);
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int v = 0; v < 3; ++v) o.RS[u][v] = o.R[u][v] * o.var[v];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int v = 0; v < 3; ++v) o.cov[u][v] = o.RS[u][0] * o.R[v][0] + o.RS[u][1] * o.R[v][1] + o.RS[u][2] * o.R[v][2];
    const float w = float(f.width), h = float(f.height);
    o.tx = fminf(fmaxf(o.x, (-0.15f * w - f.cx) / f.fx), (1.15f * w - f.cx) / f.fx);
    o.ty = fminf(fmaxf(o.y, (-0.15f * h - f.cy) / f.fy), (1.15f * h - f.cy) / f.fy);
    o.j11 = f.fx / o.depth; o.j13 = -o.j11 * o.tx; o.j22 = f.fy / o.depth; o.j23 = -o.j22 * o.ty;
#pragma unroll
    for (int v = 0; v < 3; ++v) { o.jw1[v] = o.j11 * r1[v] + o.j13 * r3[v]; o.jw2[v] = o.j22 * r2[v] + o.j23 * r3[v]; }
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        o.jc1[v] = o.jw1[0] * o.cov[0][v] + o.jw1[1] * o.cov[1][v] + o.jw1[2] * o.cov[2][v];
        o.jc2[v] = o.jw2[0] * o.cov[0][v] + o.jw2[1] * o.cov[1][v] + o.jw2[2] * o.cov[2][v];
    }
    o.a = o.jc1[0] * o.jw1[0] + o.jc1[1] * o.jw1[1] + o.jc1[2] * o.jw1[2] + DILATION;
    o.b = o.jc1[0] * o.jw2[0] + o.jc1[1] * o.jw2[1] + o.jc1[2] * o.jw2[2];
    o.c = o.jc2[0] * o.jw2[0] + o.jc2[1] * o.jw2[1] + o.jc2[2] * o.jw2[2] + DILATION;
}

// ---------------------------------------------------------------------------
// forward preprocess: one thread per primitive. Per-tile counts go through an LDS histogram (one coalesced global atomic
// per touched (workgroup, tile), see intersect.hip) when the tile grid fits.
// ---------------------------------------------------------------------------
template <bool LDS_HIST>
__global__ void __launch_bounds__(1024) fg_preprocess_kernel(
    const uint32_t N, const uint32_t per_block, const float* __restrict__ means, const float* __restrict__ scales_raw, const float* __restrict__ rot_raw,
    const float* __restrict__ opac_raw, const Frame f,
    GaussRec* __restrict__ rec, float2* __restrict__ mean2d_o, float4* __restrict__ conic_opacity_o, ushort4* __restrict__ bounds_o,
    uint32_t* __restrict__ n_touched_o, uint32_t* __restrict__ depth_bits_o, uint32_t* __restrict__ totals) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    const uint32_t T = f.gw * f.gh;
    if (LDS_HIST) {
        for (uint32_t t = threadIdx.x; t < T; t += blockDim.x) hist[t] = 0u;
        __syncthreads();
    }
    const uint32_t begin = blockIdx.x * per_block, end = min(begin + per_block, N);
    for (uint32_t i = begin + threadIdx.x; i < end; i += blockDim.x) {
        uint32_t n_touched = 0;
        do {
            const float* m = means + 3 * size_t(i);
            const float* r3 = f.w2c + 8;
            const float depth = r3[0] * m[0] + r3[1] * m[1] + r3[2] * m[2] + r3[3];
            if (depth < f.near_ || depth > f.far_) break;
            const float opacity = 1.0f / (1.0f + expf(-opac_raw[i]));
            if (opacity < MIN_ALPHA) break;
            const float4 q = reinterpret_cast<const float4*>(rot_raw)[i];
            if (q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w < 1e-8f) break;
            Cov cv;
            ewa(f, m, scales_raw + 3 * size_t(i), q, cv);
            const float det = cv.a * cv.c - cv.b * cv.b;
            if (det < 1e-8f) break;
            const float ca = cv.c / det, cb = -cv.b / det, cc = cv.a / det;
            const float mx = cv.x * f.fx + f.cx, my = cv.y * f.fy + f.cy;
            const float power_threshold = logf(opacity * MIN_ALPHA_RCP);
            const float fac = sqrtf(2.0f * power_threshold);
            const float ex = fmaxf(fac * sqrtf(cv.a) - 0.5f, 0.f), ey = fmaxf(fac * sqrtf(cv.c) - 0.5f, 0.f);
            const uint32_t x0 = min(f.gw, uint32_t(max(0, __float2int_rd((mx - ex) / float(TILE)))));
            const uint32_t x1 = min(f.gw, uint32_t(max(0, __float2int_ru((mx + ex) / float(TILE)))));
            const uint32_t y0 = min(f.gh, uint32_t(max(0, __float2int_rd((my - ey) / float(TILE)))));
            const uint32_t y1 = min(f.gh, uint32_t(max(0, __float2int_ru((my + ey) / float(TILE)))));
            if ((x1 - x0) * (y1 - y0) == 0) break;
            for (uint32_t ty = y0; ty < y1; ++ty)
                for (uint32_t tx = x0; tx < x1; ++tx)
                    if (will_contribute(mx - 0.5f, my - 0.5f, ca, cb, cc, float(tx * TILE), float(ty * TILE), float(TILE), float(TILE), power_threshold)) {
                        ++n_touched;
                        if (LDS_HIST) atomicAdd(&hist[ty * f.gw + tx], 1u); else atomicAdd(&totals[ty * f.gw + tx], 1u);
                    }
            if (n_touched == 0) break;
            // (r2 = max(SH colour + 0.5, 0) is filled in by the SH kernel of sh.hip right after this one)
            float4* r = reinterpret_cast<float4*>(rec + i);
            r[0] = make_float4(mx, my, 0.5f * LOG2E * ca, LOG2E * cb);
            r[1] = make_float4(0.5f * LOG2E * cc, LOG2E * power_threshold, opacity, 0.f);
            mean2d_o[i] = make_float2(mx, my);
            conic_opacity_o[i] = make_float4(ca, cb, cc, opacity);
            bounds_o[i] = make_ushort4(uint16_t(x0), uint16_t(x1), uint16_t(y0), uint16_t(y1));
            depth_bits_o[i] = __float_as_uint(depth);
        } while (false);
        n_touched_o[i] = n_touched;
    }
    if (LDS_HIST) {
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < T; t += blockDim.x) {
            const uint32_t c = hist[t];
            if (c) atomicAdd(&totals[t], c);
        }
    }
}

// ---------------------------------------------------------------------------
// scatter: (depth bits << 32 | primitive) of every instance into its tile bucket (unordered inside the bucket; the
// per-tile sort orders by (depth, primitive))
// ---------------------------------------------------------------------------
template <bool LDS_HIST>
__global__ void __launch_bounds__(1024) fg_scatter_kernel(
    const uint32_t N, const uint32_t per_block, const Frame f, const float2* __restrict__ mean2d, const float4* __restrict__ conic_opacity,
    const ushort4* __restrict__ bounds, const uint32_t* __restrict__ n_touched, const uint32_t* __restrict__ depth_bits,
    const int32_t* __restrict__ offsets, uint32_t* __restrict__ cursor, int64_t* __restrict__ keys) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t T = f.gw * f.gh;
    uint32_t* cnt = lds; uint32_t* base_s = lds + T;
    const uint32_t begin = blockIdx.x * per_block, end = min(begin + per_block, N);
    auto visit = [&](uint32_t i, auto&& fn) {
        if (n_touched[i] == 0) return;
        const float2 m = mean2d[i]; const float4 co = conic_opacity[i]; const ushort4 b = bounds[i];
        const float thr = logf(co.w * MIN_ALPHA_RCP);
        for (uint32_t ty = b.z; ty < b.w; ++ty)
            for (uint32_t tx = b.x; tx < b.y; ++tx)
                if (will_contribute(m.x - 0.5f, m.y - 0.5f, co.x, co.y, co.z, float(tx * TILE), float(ty * TILE), float(TILE), float(TILE), thr)) fn(ty * f.gw + tx);
    };
    if (LDS_HIST) {
        for (uint32_t t = threadIdx.x; t < T; t += blockDim.x) cnt[t] = 0u;
        __syncthreads();
        for (uint32_t i = begin + threadIdx.x; i < end; i += blockDim.x) visit(i, [&](uint32_t t) { atomicAdd(&cnt[t], 1u); });
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < T; t += blockDim.x) {
            const uint32_t c = cnt[t];
            base_s[t] = c ? atomicAdd(&cursor[t], c) : 0u;
            cnt[t] = 0u;
        }
        __syncthreads();
    }
    for (uint32_t i = begin + threadIdx.x; i < end; i += blockDim.x) {
        const uint64_t key = (uint64_t(depth_bits[i]) << 32) | uint64_t(i);
        visit(i, [&](uint32_t t) {
            const uint32_t slot = LDS_HIST ? base_s[t] + atomicAdd(&cnt[t], 1u) : atomicAdd(&cursor[t], 1u);
            keys[size_t(offsets[t]) + slot] = int64_t(key);
        });
    }
}

// ---------------------------------------------------------------------------
// backward preprocess: blend-backward accumulator -> gradients of the raw parameters (+ densification_info)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fg_preprocess_bwd_kernel(
    const uint32_t N, const float* __restrict__ means, const float* __restrict__ scales_raw, const float* __restrict__ rot_raw,
    const Frame f, const GaussRec* __restrict__ rec, const float4* __restrict__ conic_opacity,
    const uint32_t* __restrict__ n_touched, const float* __restrict__ acc, float* __restrict__ g_means, float* __restrict__ g_scales_raw, float* __restrict__ g_rot_raw,
    float* __restrict__ g_opac_raw, float* __restrict__ densification_info) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (n_touched[i] == 0) { // the reference leaves these rows at the zeros they were allocated with
#pragma unroll
        for (int c = 0; c < 3; ++c) { g_means[3 * size_t(i) + c] = 0.f; g_scales_raw[3 * size_t(i) + c] = 0.f; }
        reinterpret_cast<float4*>(g_rot_raw)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        g_opac_raw[i] = 0.f;
        return;
    }
    const float4* a4 = reinterpret_cast<const float4*>(acc + size_t(i) * ACC_STRIDE);
    const float4 a0 = a4[0], a1 = a4[1], a2 = a4[2];
    const GaussRec r = rec[i];
    // accumulator row (fastgs_blend.hip): {S1 = sum h dx, S2 = sum h dy, sum h dx dx, sum h dx dy | sum h dy dy, dc.r, dc.g, dc.b | sum alpha dL/dalpha}
    // with h = -alpha dL/dalpha: dL/dmean2d = conic (S1, S2), dL/dconic = 0.5 (sum h dx dx, sum h dx dy, sum h dy dy)
    const float4 co = conic_opacity[i];
    const float dm2[2] = {co.x * a0.x + co.y * a0.y, co.y * a0.x + co.z * a0.y};
    const float dcon[3] = {0.5f * a0.z, 0.5f * a0.w, 0.5f * a1.x};
    const float opacity = r.r1.z;
    g_opac_raw[i] = a2.x * (1.0f - opacity);
    const float* m = means + 3 * size_t(i);
    const float dpos[3] = {0.f, 0.f, 0.f}; // (the colour -> position term is added by the SH backward kernel of sh.hip, which runs after this one)
    // ---- EWA backward
    const float4 q = reinterpret_cast<const float4*>(rot_raw)[i];
    Cov cv;
    ewa(f, m, scales_raw + 3 * size_t(i), q, cv);
    const float* r1 = f.w2c; const float* r2 = f.w2c + 4; const float* r3 = f.w2c + 8;
    const float A = cv.a, B = cv.b, C = cv.c;
    const float det = A * C - B * B, dr = 1.0f / det, dr2 = dr * dr;
    const float dcv[3] = {dr2 * (2.0f * B * C * dcon[1] - C * C * dcon[0] - B * B * dcon[2]),
                          dr2 * (B * C * dcon[0] - (A * C + B * B) * dcon[1] + A * B * dcon[2]),
                          dr2 * (2.0f * A * B * dcon[1] - B * B * dcon[0] - A * A * dcon[2])};
    float G[3][3];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int v = 0; v < 3; ++v)
            G[u][v] = (cv.jw1[u] * cv.jw1[v]) * dcv[0] + (cv.jw1[u] * cv.jw2[v] + cv.jw1[v] * cv.jw2[u]) * dcv[1] + (cv.jw2[u] * cv.jw2[v]) * dcv[2];
    float djw1[3], djw2[3];
#pragma unroll
    for (int v = 0; v < 3; ++v) { djw1[v] = 2.0f * (cv.jc1[v] * dcv[0] + cv.jc2[v] * dcv[1]); djw2[v] = 2.0f * (cv.jc1[v] * dcv[1] + cv.jc2[v] * dcv[2]); }
    const float dj11 = r1[0] * djw1[0] + r1[1] * djw1[1] + r1[2] * djw1[2], dj22 = r2[0] * djw2[0] + r2[1] * djw2[1] + r2[2] * djw2[2];
    const float dj13 = r3[0] * djw1[0] + r3[1] * djw1[1] + r3[2] * djw1[2], dj23 = r3[0] * djw2[0] + r3[1] * djw2[1] + r3[2] * djw2[2];
    const float h1 = dj11 - 2.0f * cv.tx * dj13, h2 = dj22 - 2.0f * cv.ty * dj23;
    const float dcam[3] = {cv.j11 * (dm2[0] - dj13 / cv.depth), cv.j22 * (dm2[1] - dj23 / cv.depth),
                           -cv.j11 * (cv.x * dm2[0] + h1 / cv.depth) - cv.j22 * (cv.y * dm2[1] + h2 / cv.depth)};
#pragma unroll
    for (int c = 0; c < 3; ++c) g_means[3 * size_t(i) + c] = r1[c] * dcam[0] + r2[c] * dcam[1] + r3[c] * dcam[2] + dpos[c];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        const float dvar = cv.R[0][v] * cv.R[0][v] * G[0][0] + cv.R[1][v] * cv.R[1][v] * G[1][1] + cv.R[2][v] * cv.R[2][v] * G[2][2] +
                           2.0f * (cv.R[0][v] * cv.R[1][v] * G[0][1] + cv.R[0][v] * cv.R[2][v] * G[0][2] + cv.R[1][v] * cv.R[2][v] * G[1][2]);
        g_scales_raw[3 * size_t(i) + v] = 2.0f * cv.var[v] * dvar;
    }
    float dR[3][3];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int v = 0; v < 3; ++v) dR[u][v] = 2.0f * (cv.RS[0][v] * G[u][0] + cv.RS[1][v] * G[u][1] + cv.RS[2][v] * G[u][2]);
    const float dqxx = -dR[1][1] - dR[2][2], dqyy = -dR[0][0] - dR[2][2], dqzz = -dR[0][0] - dR[1][1];
    const float dqxy = dR[0][1] + dR[1][0], dqxz = dR[0][2] + dR[2][0], dqyz = dR[1][2] + dR[2][1];
    const float dqrx = dR[2][1] - dR[1][2], dqry = dR[0][2] - dR[2][0], dqrz = dR[1][0] - dR[0][1];
    const float hn = cv.qxx * dqxx + cv.qyy * dqyy + cv.qzz * dqzz + cv.qxy * dqxy + cv.qxz * dqxz + cv.qyz * dqyz + cv.qrx * dqrx + cv.qry * dqry + cv.qrz * dqrz;
    reinterpret_cast<float4*>(g_rot_raw)[i] = make_float4(
        2.0f * (cv.qx * dqrx + cv.qy * dqry + cv.qz * dqrz - cv.qr * hn) / cv.qn,
        2.0f * (2.0f * cv.qx * dqxx + cv.qy * dqxy + cv.qz * dqxz + cv.qr * dqrx - cv.qx * hn) / cv.qn,
        2.0f * (2.0f * cv.qy * dqyy + cv.qx * dqxy + cv.qz * dqyz + cv.qr * dqry - cv.qy * hn) / cv.qn,
        2.0f * (2.0f * cv.qz * dqzz + cv.qx * dqxz + cv.qy * dqyz + cv.qr * dqrz - cv.qz * hn) / cv.qn);
    if (densification_info != nullptr) { // kernels_backward.cuh:229-232
        const float gx = dm2[0] * 0.5f * float(f.width), gy = dm2[1] * 0.5f * float(f.height);
        densification_info[i] += 1.0f;
        densification_info[size_t(N) + i] += sqrtf(gx * gx + gy * gy);
    }
}

static inline uint32_t per_block_for(uint32_t N) {
    size_t pb = (size_t(N) + 511) / 512;
    if (pb < 1024) pb = 1024;
    return uint32_t((pb + 1023) / 1024 * 1024);
}

// host-side launchers used by fastgs_blend.hip as well
int launch_scatter(uint32_t N, const Frame& f, const PrimWs& w, int64_t* keys, hipStream_t s) {
    const uint32_t T = f.gw * f.gh, pb = per_block_for(N), blocks = (N + pb - 1) / pb;
    lfs::ProfScope prof("fastgs_scatter", s);
    if (size_t(T) * 8 <= 64 * 1024)
        hipLaunchKernelGGL(fg_scatter_kernel<true>, dim3(blocks), dim3(1024), size_t(T) * 8, s, N, pb, f, w.mean2d, w.conic_opacity, w.bounds, w.n_touched, w.depth_bits, w.offsets, w.cursor, keys);
    else
        hipLaunchKernelGGL(fg_scatter_kernel<false>, dim3(blocks), dim3(1024), 0, s, N, pb, f, w.mean2d, w.conic_opacity, w.bounds, w.n_touched, w.depth_bits, w.offsets, w.cursor, keys);
    return (int)hipGetLastError();
}
int launch_preprocess_bwd(uint32_t N, const float* means, const float* scales_raw, const float* rot_raw, const float* sh0, const float* sh_rest, const Frame& f, const PrimWs& w,
                          float* g_means, float* g_scales_raw, float* g_rot_raw, float* g_opac_raw, float* g_sh0, float* g_sh_rest, float* densification_info, hipStream_t s,
                          const ShAdamArgs* adam) {
    {
        lfs::ProfScope prof("fastgs_preprocess_bwd", s);
        hipLaunchKernelGGL(fg_preprocess_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, means, scales_raw, rot_raw, f, w.rec, w.conic_opacity, w.n_touched, w.acc,
                           g_means, g_scales_raw, g_rot_raw, g_opac_raw, densification_info);
    }
    // SH backward (convert_sh_to_color_backward, kernel_utils.cuh:38-106) with the coalesced three-phase kernel of sh.hip: reads dL/d(clamped colour) from the
    // accumulator rows (floats 6..8 of 16), the clamp mask from the record's colour (floats 8..10 of 16), writes g_sh0 / g_sh_rest fully, adds dL/dposition to g_means
    const uint32_t degree = f.active_sh_bases >= 16 ? 3 : f.active_sh_bases >= 9 ? 2 : f.active_sh_bases >= 4 ? 1 : 0;
    return sh_records_bwd(N, 1 + f.total_rest, degree, means, f.cam_pos, sh0, sh_rest, w.n_touched, reinterpret_cast<const float*>(w.rec) + 8, 16, w.acc + 5, 16,
                          g_sh0, g_sh_rest, g_means, s, adam);
}

} // namespace fgs
} // namespace lfs

using namespace lfs;

namespace lfs { namespace fgs {
// one pinned int64 + event per process: the reference's trainer (and this one) has a single rendering thread
struct Readback { int64_t* host = nullptr; hipEvent_t event = nullptr; bool armed = false; };
static Readback& readback() {
    static Readback rb = [] {
        Readback r;
        if (hipHostMalloc((void**)&r.host, sizeof(int64_t), hipHostMallocDefault) != hipSuccess) r.host = nullptr;
        if (hipEventCreateWithFlags(&r.event, hipEventDisableTiming) != hipSuccess) r.event = nullptr;
        return r;
    }();
    return rb;
}
} } // namespace lfs::fgs

// n_instances of the last lfs_fastgs_preprocess call on this process: waits for the read-back event only (not for the SH kernel queued after it)
extern "C" int lfs_fastgs_wait_n_instances(int64_t* n_instances) {
    if (!n_instances) return LFS_E_INVALID;
    lfs::fgs::Readback& rb = lfs::fgs::readback();
    if (!rb.armed) return LFS_E_INVALID;
    const hipError_t e = hipEventSynchronize(rb.event);
    if (e != hipSuccess) return (int)e;
    *n_instances = *rb.host;
    rb.armed = false;
    return LFS_OK;
}

extern "C" size_t lfs_fastgs_primitive_workspace_bytes(uint32_t N, uint32_t width, uint32_t height) { return fgs::prim_ws(nullptr, N, width, height).bytes; }
extern "C" size_t lfs_fastgs_instance_workspace_bytes(uint32_t width, uint32_t height, int64_t n_instances) {
    return n_instances < 0 ? 0 : fgs::inst_ws(nullptr, width, height, uint64_t(n_instances)).bytes;
}

extern "C" int lfs_fastgs_preprocess(
    uint32_t N, const float* means, const float* scales_raw, const float* rotations_raw, const float* opacities_raw,
    const float* sh_coefficients_0, const float* sh_coefficients_rest, uint32_t total_bases_sh_rest, const float* w2c, const float* cam_position,
    uint32_t active_sh_bases, uint32_t width, uint32_t height, float fx, float fy, float cx, float cy, float near_plane, float far_plane,
    int64_t* n_instances, void* primitive_workspace, size_t primitive_workspace_bytes, lfs_stream_t stream) {
    if (!n_instances || !primitive_workspace || !w2c || !cam_position || width == 0 || height == 0) return LFS_E_INVALID;
    if (active_sh_bases == 0 || active_sh_bases > 16 || (active_sh_bases > 1 && total_bases_sh_rest + 1 < active_sh_bases)) return LFS_E_INVALID;
    fgs::PrimWs w = fgs::prim_ws(primitive_workspace, N, width, height);
    if (primitive_workspace_bytes < w.bytes) return LFS_E_WORKSPACE;
    if (N > 0 && (!means || !scales_raw || !rotations_raw || !opacities_raw || !sh_coefficients_0 || (total_bases_sh_rest > 0 && !sh_coefficients_rest))) return LFS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    fgs::Frame f{w2c, cam_position, active_sh_bases, total_bases_sh_rest, width, height, (width + fgs::TILE - 1) / fgs::TILE, (height + fgs::TILE - 1) / fgs::TILE,
                 fx, fy, cx, cy, near_plane, far_plane};
    const uint32_t T = f.gw * f.gh;
    if (T > 65535u) return LFS_E_UNSUPPORTED; // bounds are ushort4 (as in the reference: ushort tile keys)
    hipError_t e = hipMemsetAsync(w.totals, 0, (char*)w.offsets - (char*)w.totals, s); // totals + cursor
    if (e != hipSuccess) return (int)e;
    {
        lfs::ProfScope prof("fastgs_preprocess", s);
        if (N > 0) {
            const uint32_t pb = fgs::per_block_for(N), blocks = (N + pb - 1) / pb;
            if (size_t(T) * 8 <= 64 * 1024)
                hipLaunchKernelGGL(fgs::fg_preprocess_kernel<true>, dim3(blocks), dim3(1024), size_t(T) * 4, s, N, pb, means, scales_raw, rotations_raw, opacities_raw,
                                   f, w.rec, w.mean2d, w.conic_opacity, w.bounds, w.n_touched, w.depth_bits, w.totals);
            else
                hipLaunchKernelGGL(fgs::fg_preprocess_kernel<false>, dim3(blocks), dim3(1024), 0, s, N, pb, means, scales_raw, rotations_raw, opacities_raw,
                                   f, w.rec, w.mean2d, w.conic_opacity, w.bounds, w.n_touched, w.depth_bits, w.totals);
        }
        hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, s, T, w.totals, w.offsets, n_instances);
    }
    // The host needs n_instances to size the instance workspace (the reference syncs here, forward.cu:114-117). Queue the read-back BEFORE the SH
    // launch and mark it with an event: lfs_fastgs_wait_n_instances() then returns as soon as the count has landed while the GPU is already
    // evaluating SH colours, instead of idling through the host round trip (~40 us).
    {
        fgs::Readback& rb = fgs::readback();
        if (rb.host && rb.event) {
            if (hipMemcpyAsync(rb.host, n_instances, sizeof(int64_t), hipMemcpyDeviceToHost, s) == hipSuccess) { (void)hipEventRecord(rb.event, s); rb.armed = true; }
        }
    }
    if (N > 0) { // SH colour (convert_sh_to_color, kernel_utils.cuh:15-36) of the visible primitives, written into the records
        const uint32_t degree = active_sh_bases >= 16 ? 3 : active_sh_bases >= 9 ? 2 : active_sh_bases >= 4 ? 1 : 0;
        const int rc = sh_records_fwd(N, 1 + total_bases_sh_rest, degree, means, cam_position, sh_coefficients_0, sh_coefficients_rest, w.n_touched,
                                      reinterpret_cast<float*>(w.rec) + 8, 16, s);
        if (rc) return rc;
    }
    return (int)hipGetLastError();
}
