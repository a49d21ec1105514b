// Bilateral-grid appearance model: slice forward / backward and the total-variation regulariser (SURVEY.md §8f row 2,
// BASELINE config 5). Reference behaviour: src/training/kernels/bilateral_grid_forward.cu:13-94 (slice),
// bilateral_grid_backward.cu:14-153 (slice backward), bilateral_grid_tv.cu:12-135 (TV), include/kernels/bilateral_grid.cuh:12-33.
//
// CDNA4 design. A pixel's (x, y) grid coordinates are a pure function of its position ("uniform coordinates"), only the
// guidance axis z depends on its colour. A 64x4 pixel tile therefore touches a tiny x/y window of the grid (2-3 x 2 cells at
// 1080p with a 16x16 grid): the workgroup stages that window (all 12 affine channels, all L levels) in LDS once, every
// trilinear tap of the forward and of the backward's dL/drgb is an LDS read, and the backward's dL/dgrid - 96 scatter-adds per
// pixel in the reference, straight to global memory (bilateral_grid_backward.cu:97-101): ~2e8 atomics per 1080p image onto
// 24 576 addresses - is computed as the dense contraction over pixels it is, on the f32 matrix cores (see
// slice_bwd_window_kernel), then leaves the workgroup with ONE global atomic per touched cell. Windows that do not fit
// (tiny images against big grids) take the generic global path; both paths evaluate the same per-pixel expressions.
//
// Layout extension: colours are addressed as rgb[pixel * ps + channel * cs] so the HWC image of the 3DGUT rasterizer
// (ps = 3, cs = 1) and the CHW image of the fastgs rasterizer (ps = 1, cs = h * w) are consumed without a permute, and
// clamp_input folds BilateralGrid::apply's clamp(rgb, 0, 1) (components/bilateral_grid.cpp:115) into both passes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lfs_gsplat.h"
#include "lfs_prof.h"

namespace lfs {
namespace bg {

constexpr int TILE_W = 64, TILE_H = 4, THREADS = TILE_W * TILE_H;
constexpr int LDS_FLOATS = 6144;               // forward: 24 KB value window at most

struct Dims { int L, H, W, h, w; uint32_t ps, cs; int clamp_input; };

struct Tap { // the trilinear stencil of one pixel (bilateral_grid_forward.cu:33-49)
    int x0, x1, y0, y1, z0, z1; float fx, fy, fz, z;
};

__device__ __forceinline__ float grid_coord(int i, int n, int G) { return (float)i / (float)(n - 1) * (float)(G - 1); }

__device__ __forceinline__ Tap make_tap(const Dims& d, int hi, int wi, float r, float g, float b) {
    Tap t;
    const float x = grid_coord(wi, d.w, d.W), y = grid_coord(hi, d.h, d.H);
    const float gz = 0.299f * r + 0.587f * g + 0.114f * b;
    t.z = gz * (float)(d.L - 1);
    t.x0 = (int)floorf(x); t.y0 = (int)floorf(y);
    int z0 = (int)floorf(t.z);
    t.x1 = min(t.x0 + 1, d.W - 1); t.y1 = min(t.y0 + 1, d.H - 1);
    t.z1 = min(max(z0 + 1, 0), d.L - 1);
    // the reference clamps z0 only from below (:46); colours above 1 would read out of bounds there - clamp both sides
    t.z0 = min(max(z0, 0), d.L - 1);
    t.fx = x - (float)t.x0; t.fy = y - (float)t.y0; t.fz = t.z - (float)t.z0;
    return t;
}

__device__ __forceinline__ void corner_weights(const Tap& t, float w[8]) {
    const float ax = 1.f - t.fx, ay = 1.f - t.fy, az = 1.f - t.fz;
    w[0] = ax * ay * az; w[1] = t.fx * ay * az; w[2] = ax * t.fy * az; w[3] = t.fx * t.fy * az;
    w[4] = ax * ay * t.fz; w[5] = t.fx * ay * t.fz; w[6] = ax * t.fy * t.fz; w[7] = t.fx * t.fy * t.fz;
}

// x/y window of the grid a pixel tile touches; uniform per workgroup. grid_coord is monotone in i, so the first and the
// last pixel of the tile bound it.
struct Window { int xa, nx, ya, ny, cells; };
__device__ __forceinline__ Window tile_window(const Dims& d, int px0, int py0, int rows) {
    Window wd;
    const int px1 = min(px0 + TILE_W, d.w) - 1, py1 = min(py0 + rows, d.h) - 1;
    wd.xa = (int)floorf(grid_coord(px0, d.w, d.W)); wd.ya = (int)floorf(grid_coord(py0, d.h, d.H));
    const int xb = min((int)floorf(grid_coord(px1, d.w, d.W)) + 1, d.W - 1), yb = min((int)floorf(grid_coord(py1, d.h, d.H)) + 1, d.H - 1);
    wd.nx = xb - wd.xa + 1; wd.ny = yb - wd.ya + 1;
    wd.cells = wd.nx * wd.ny * d.L;
    return wd;
}

// LDS window layout: [ci][z][y - ya][x - xa]
__device__ __forceinline__ void load_window(const Dims& d, const Window& wd, const float* __restrict__ grid, float* s) {
    const int n = 12 * wd.cells, plane = d.L * d.H * d.W;
    for (int i = threadIdx.x; i < n; i += THREADS) {
        int r = i;
        const int x = r % wd.nx; r /= wd.nx;
        const int y = r % wd.ny; r /= wd.ny;
        const int z = r % d.L, ci = r / d.L;
        s[i] = grid[ci * plane + (z * d.H + y + wd.ya) * d.W + x + wd.xa];
    }
}

template <bool LDS>
struct GridView { // corner offsets inside one channel plane + the channel stride, for the LDS window or the global grid
    int off[8], stride;
    __device__ __forceinline__ GridView(const Dims& d, const Window& wd, const Tap& t) {
        int x0 = t.x0, x1 = t.x1, y0 = t.y0, y1 = t.y1, rw, rh;
        if (LDS) { x0 -= wd.xa; x1 -= wd.xa; y0 -= wd.ya; y1 -= wd.ya; rw = wd.nx; rh = wd.ny; stride = wd.cells; }
        else { rw = d.W; rh = d.H; stride = d.L * d.H * d.W; }
        const int a = t.z0 * rh, b = t.z1 * rh;
        off[0] = (a + y0) * rw + x0; off[1] = (a + y0) * rw + x1; off[2] = (a + y1) * rw + x0; off[3] = (a + y1) * rw + x1;
        off[4] = (b + y0) * rw + x0; off[5] = (b + y0) * rw + x1; off[6] = (b + y1) * rw + x0; off[7] = (b + y1) * rw + x1;
    }
};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

template <bool LDS>
__global__ void __launch_bounds__(THREADS) slice_fwd_kernel(Dims d, const float* __restrict__ grid, const float* __restrict__ rgb,
                                                            float* __restrict__ out) {
    extern __shared__ float s_val[];
    const int px0 = blockIdx.x * TILE_W, py0 = blockIdx.y * TILE_H;
    Window wd = {};
    if (LDS) {
        wd = tile_window(d, px0, py0, TILE_H);
        load_window(d, wd, grid, s_val);
        __syncthreads();
    }
    const int wi = px0 + (threadIdx.x & (TILE_W - 1)), hi = py0 + threadIdx.x / TILE_W;
    if (wi >= d.w || hi >= d.h) return;
    const size_t p = (size_t)hi * d.w + wi;
    float c[4] = {rgb[p * d.ps], rgb[p * d.ps + d.cs], rgb[p * d.ps + 2 * (size_t)d.cs], 1.f};
    if (d.clamp_input) { c[0] = clamp01(c[0]); c[1] = clamp01(c[1]); c[2] = clamp01(c[2]); }
    const Tap t = make_tap(d, hi, wi, c[0], c[1], c[2]);
    float w[8];
    corner_weights(t, w);
    const GridView<LDS> gv(d, wd, t);
    const float* src = LDS ? s_val : grid;
    float res[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int ci = 0; ci < 12; ++ci) {
        const float* pl = src + ci * gv.stride;
        float val = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) val += pl[gv.off[k]] * w[k];
        res[ci / 4] += val * c[ci % 4];
    }
    out[p * d.ps] = res[0]; out[p * d.ps + d.cs] = res[1]; out[p * d.ps + 2 * (size_t)d.cs] = res[2];
}

// ---- backward ----
// Per pixel: dL/drgb (through the affine coefficients :103-111 and through z :117-147) and the 12 products
// gw[ci] = coeff[si] * dL/dout[di] whose outer product with the 8 trilinear weights is the pixel's dL/dgrid (:85-101).
template <bool LDS>
__device__ __forceinline__ void pixel_backward(const Dims& d, const Window& wd, const Tap& t, const float* __restrict__ src, const float c[4],
                                               const float raw[3], const float go[3], float gw[12], float grad[3]) {
    float w[8];
    corner_weights(t, w);
    const GridView<LDS> gv(d, wd, t);
    float v[3] = {0.f, 0.f, 0.f}, tri[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ci = 0; ci < 12; ++ci) {
        const int si = ci % 4, di = ci / 4;
        gw[ci] = c[si] * go[di];
        float lerp = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float val = src[ci * gv.stride + gv.off[k]];
            lerp += val * w[k];
            tri[k] += val * gw[ci];
        }
        if (si < 3) v[si] += lerp * go[di];
    }
    const float ax = 1.f - t.fx, ay = 1.f - t.fy;
    const float dwdz[8] = {-ax * ay, -t.fx * ay, -ax * t.fy, -t.fx * t.fy, ax * ay, t.fx * ay, ax * t.fy, t.fx * t.fy};
    float gz = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) gz += dwdz[k] * (float)(d.L - 1) * tri[k];
    gz *= (float)((float)t.z0 != t.z && (float)t.z1 != t.z); // discontinuity mask (:150)
    const float lum[3] = {0.299f, 0.587f, 0.114f};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        grad[k] = v[k] + lum[k] * gz;
        if (d.clamp_input && !(raw[k] >= 0.f && raw[k] <= 1.f)) grad[k] = 0.f; // clamp backward (ATen: pass where min <= x <= max)
    }
}

struct Pixel { float raw[3], c[4], go[3]; };
__device__ __forceinline__ Pixel load_pixel(const Dims& d, size_t p, const float* __restrict__ rgb, const float* __restrict__ grad_out) {
    Pixel px;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        px.raw[k] = rgb[p * d.ps + k * (size_t)d.cs];
        px.go[k] = grad_out[p * d.ps + k * (size_t)d.cs];
        px.c[k] = d.clamp_input ? clamp01(px.raw[k]) : px.raw[k];
    }
    px.c[3] = 1.f;
    return px;
}

// generic path: grid read from global memory, 96 global float atomics per pixel (what the reference does everywhere)
__global__ void __launch_bounds__(THREADS) slice_bwd_generic_kernel(Dims d, const float* __restrict__ grid, const float* __restrict__ rgb,
                                                                    const float* __restrict__ grad_out, float* __restrict__ grad_grid,
                                                                    float* __restrict__ grad_rgb) {
    const int wi = blockIdx.x * TILE_W + (threadIdx.x & (TILE_W - 1)), hi = blockIdx.y * TILE_H + threadIdx.x / TILE_W;
    if (wi >= d.w || hi >= d.h) return;
    const size_t p = (size_t)hi * d.w + wi;
    const Pixel px = load_pixel(d, p, rgb, grad_out);
    const Tap t = make_tap(d, hi, wi, px.c[0], px.c[1], px.c[2]);
    const Window wd = {};
    float gw[12], g[3], w[8];
    pixel_backward<false>(d, wd, t, grid, px.c, px.raw, px.go, gw, g);
    corner_weights(t, w);
    const GridView<false> gv(d, wd, t);
#pragma unroll
    for (int ci = 0; ci < 12; ++ci) {
        if (gw[ci] == 0.f) continue;
#pragma unroll
        for (int k = 0; k < 8; ++k) unsafeAtomicAdd(grad_grid + ci * gv.stride + gv.off[k], w[k] * gw[ci]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) grad_rgb[p * d.ps + k * (size_t)d.cs] = g[k];
}

// Windowed path. dL/dgrid is a contraction over pixels: G[ci][x][y][z] = sum_p gw_p[ci] * wx_p[x] * wy_p[y] * wz_p[z]. A wavefront
// owns 8 consecutive image rows of a 64-column strip; within one row y0 / y1 / fy are wave-uniform, so per row the wavefront needs
//     S[ci][x * L + z] = sum over its 64 pixels of gw_p[ci] * (wx_p[x] * wz_p[z])        (12 x nx*L, K = 64 pixels)
// which is a dense [16 x 64] x [64 x 16 NT] product: v_mfma_f32_16x16x4_f32 (exact f32 FMA chains), 16 K-steps x NT column tiles
// per row. The pixels' gw vectors and stencil records are transposed into the MFMA operand layout through a per-wave LDS
// staging buffer. Rows are folded into two register accumulators, G[.., y0] += (1 - fy) S and G[.., y1] += fy S, that are
// flushed (LDS float atomics into the workgroup's gradient window) only when y0 changes; the window goes to global memory
// with one atomic per touched cell and workgroup. LDS float atomics per pixel were measured at ~1 ms per 1080p image (96
// ds_add_f32 per pixel, whether or not the addresses conflict); this path has none per pixel.
constexpr int BWD_ROWS = 8, BWD_TILE_H = 4 * BWD_ROWS, NT_MAX = 4, STAGE_STRIDE = 20;

__global__ void __launch_bounds__(THREADS) slice_bwd_window_kernel(Dims d, const float* __restrict__ grid, const float* __restrict__ rgb,
                                                                   const float* __restrict__ grad_out, float* __restrict__ grad_grid,
                                                                   float* __restrict__ grad_rgb) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ float s_mem[]; // value window | gradient window | 4 staging buffers [64][STAGE_STRIDE]
    const int px0 = blockIdx.x * TILE_W, py0 = blockIdx.y * BWD_TILE_H;
    const Window wd = tile_window(d, px0, py0, BWD_TILE_H);
    float* s_val = s_mem;
    float* s_grad = s_mem + 12 * wd.cells;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* stage = s_mem + 24 * wd.cells + wave * (64 * STAGE_STRIDE);
    load_window(d, wd, grid, s_val);
    for (int i = threadIdx.x; i < 12 * wd.cells; i += THREADS) s_grad[i] = 0.f;
    __syncthreads();

    const int ncols = wd.nx * d.L, nt = (ncols + 15) >> 4;
    int col_x[NT_MAX], col_z[NT_MAX]; // this lane's B / D column in each column tile
#pragma unroll
    for (int n = 0; n < NT_MAX; ++n) {
        const int j = 16 * n + (lane & 15);
        col_x[n] = j < ncols ? j / d.L : -1;
        col_z[n] = j < ncols ? j % d.L : -1;
    }
    f32x4 acc_lo[NT_MAX], acc_hi[NT_MAX];
#pragma unroll
    for (int n = 0; n < NT_MAX; ++n) { acc_lo[n] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_hi[n] = acc_lo[n]; }
    int cur_y0 = -1, cur_y1 = -1;

    auto flush = [&]() {
        if (cur_y0 < 0) return;
#pragma unroll
        for (int n = 0; n < NT_MAX; ++n) {
            if (n >= nt || col_x[n] < 0) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = 4 * (lane >> 4) + r;
                if (ci >= 12) continue;
                float* base = s_grad + ci * wd.cells + col_z[n] * (wd.ny * wd.nx) + col_x[n];
                if (acc_lo[n][r] != 0.f) atomicAdd(base + (cur_y0 - wd.ya) * wd.nx, acc_lo[n][r]);
                if (acc_hi[n][r] != 0.f) atomicAdd(base + (cur_y1 - wd.ya) * wd.nx, acc_hi[n][r]);
            }
            acc_lo[n] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_hi[n] = acc_lo[n];
        }
    };

    const int wi = px0 + lane;
    for (int row = 0; row < BWD_ROWS; ++row) {
        const int hi = py0 + wave * BWD_ROWS + row; // wave-uniform
        if (hi >= d.h) break;
        const float y = grid_coord(hi, d.h, d.H);
        const int y0 = (int)floorf(y), y1 = min(y0 + 1, d.H - 1);
        const float fy = y - (float)y0;
        if (y0 != cur_y0) { flush(); cur_y0 = y0; cur_y1 = y1; }

        float gw[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
        if (wi < d.w) {
            const size_t p = (size_t)hi * d.w + wi;
            const Pixel px = load_pixel(d, p, rgb, grad_out);
            const Tap t = make_tap(d, hi, wi, px.c[0], px.c[1], px.c[2]);
            float g[3];
            pixel_backward<true>(d, wd, t, s_val, px.c, px.raw, px.go, gw, g);
#pragma unroll
            for (int k = 0; k < 3; ++k) grad_rgb[p * d.ps + k * (size_t)d.cs] = g[k];
            rec = make_float4(__int_as_float((t.x0 - wd.xa) | ((t.x1 - wd.xa) << 8) | (t.z0 << 16) | (t.z1 << 24)), t.fx, t.fz, 0.f);
        }
        // transpose through LDS: pixel-major [64][16 gw (12 used) | record]
        float4* st = reinterpret_cast<float4*>(stage + lane * STAGE_STRIDE);
        st[0] = make_float4(gw[0], gw[1], gw[2], gw[3]); st[1] = make_float4(gw[4], gw[5], gw[6], gw[7]);
        st[2] = make_float4(gw[8], gw[9], gw[10], gw[11]); st[3] = make_float4(0.f, 0.f, 0.f, 0.f);
        st[4] = rec;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        f32x4 S[NT_MAX];
#pragma unroll
        for (int n = 0; n < NT_MAX; ++n) S[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int g = 0; g < 16; ++g) {
            const float* sp = stage + (4 * g + (lane >> 4)) * STAGE_STRIDE;
            const float a = sp[lane & 15];
            const float4 r4 = *reinterpret_cast<const float4*>(sp + 16);
            const int idx = __float_as_int(r4.x);
            const int x0r = idx & 255, x1r = (idx >> 8) & 255, z0 = (idx >> 16) & 255, z1 = (idx >> 24) & 255;
#pragma unroll
            for (int n = 0; n < NT_MAX; ++n) {
                if (n >= nt) continue; // wave-uniform
                const float wx = (col_x[n] == x0r ? 1.f - r4.y : 0.f) + (col_x[n] == x1r ? r4.y : 0.f);
                const float wz = (col_z[n] == z0 ? 1.f - r4.z : 0.f) + (col_z[n] == z1 ? r4.z : 0.f);
                S[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wx * wz, S[n], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier(); // the staging buffer is rewritten by the next row
#pragma unroll
        for (int n = 0; n < NT_MAX; ++n) {
            if (n >= nt) continue;
            acc_lo[n] += S[n] * (1.f - fy);
            acc_hi[n] += S[n] * fy;
        }
    }
    flush();
    __syncthreads();
    const int n = 12 * wd.cells, plane = d.L * d.H * d.W;
    for (int i = threadIdx.x; i < n; i += THREADS) {
        const float g = s_grad[i];
        if (g == 0.f) continue;
        int r = i;
        const int x = r % wd.nx; r /= wd.nx;
        const int y = r % wd.ny; r /= wd.ny;
        const int z = r % d.L, ci = r / d.L;
        unsafeAtomicAdd(grad_grid + ci * plane + (z * d.H + y + wd.ya) * d.W + x + wd.xa, g);
    }
}

// ---- total variation ----
// loss = 1/(12 N) * sum_dirs mean over the direction's differences of diff^2 (bilateral_grid_tv.cu:39-69)
__global__ void __launch_bounds__(256) tv_fwd_kernel(const float* __restrict__ grids, float* __restrict__ loss, uint32_t NC, int L, int H, int W,
                                                      float sx, float sy, float sz) {
    const size_t total = (size_t)NC * L * H * W;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int wi = (int)(i % W), hi = (int)((i / W) % H), li = (int)((i / ((size_t)W * H)) % L);
        const float v = grids[i];
        if (wi > 0) { const float dd = v - grids[i - 1]; acc += dd * dd * sx; }
        if (hi > 0) { const float dd = v - grids[i - W]; acc += dd * dd * sy; }
        if (li > 0) { const float dd = v - grids[i - (size_t)W * H]; acc += dd * dd * sz; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(loss, part[0] + part[1] + part[2] + part[3]);
}

template <bool ACCUM>
__global__ void __launch_bounds__(256) tv_bwd_kernel(const float* __restrict__ grids, float* __restrict__ grad, uint32_t NC, int L, int H, int W,
                                                      float sx, float sy, float sz) {
    const size_t total = (size_t)NC * L * H * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int wi = (int)(i % W), hi = (int)((i / W) % H), li = (int)((i / ((size_t)W * H)) % L);
        const float v = grids[i];
        float g = 0.f;
        if (wi > 0) g += (v - grids[i - 1]) * sx;
        if (wi < W - 1) g += (v - grids[i + 1]) * sx;
        if (hi > 0) g += (v - grids[i - W]) * sy;
        if (hi < H - 1) g += (v - grids[i + W]) * sy;
        if (li > 0) g += (v - grids[i - (size_t)W * H]) * sz;
        if (li < L - 1) g += (v - grids[i + (size_t)W * H]) * sz;
        grad[i] = ACCUM ? grad[i] + g : g;
    }
}

static int check_dims(uint32_t L, uint32_t H, uint32_t W, uint32_t h, uint32_t w) {
    if (!L || !H || !W || h < 2 || w < 2) return LFS_E_INVALID; // (h - 1), (w - 1) divide (bilateral_grid_forward.cu:34-35)
    if ((uint64_t)12 * L * H * W >= (1ull << 31) || (uint64_t)h * w >= (1ull << 31)) return LFS_E_INVALID;
    return LFS_OK;
}

// worst-case window of any tile: decides LDS vs generic path on the host (same arithmetic bound as tile_window)
static size_t window_floats(uint32_t L, uint32_t H, uint32_t W, uint32_t h, uint32_t w, uint32_t rows = TILE_H) {
    auto span = [](uint32_t tile, uint32_t n, uint32_t G) {
        const double per_px = (double)(G - 1) / (double)(n - 1);
        uint32_t s = (uint32_t)(per_px * (tile - 1) + 1e-3) + 3; // floor of the span + both partial cells + x1 (+ fp32 slack)
        return s < G ? s : G;
    };
    return (size_t)12 * L * span(TILE_W, w, W) * span(rows, h, H);
}

} // namespace bg
} // namespace lfs

using namespace lfs;
using namespace lfs::bg;

extern "C" int lfs_bilateral_slice_fwd(uint32_t L, uint32_t H, uint32_t W, uint32_t h, uint32_t w, const float* grid, const float* rgb,
                                       uint32_t chw, uint32_t clamp_input, float* output, lfs_stream_t stream) {
    if (int rc = check_dims(L, H, W, h, w)) return rc;
    if (!grid || !rgb || !output) return LFS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const Dims d{(int)L, (int)H, (int)W, (int)h, (int)w, chw ? 1u : 3u, chw ? h * w : 1u, (int)(clamp_input != 0)};
    const dim3 g((w + TILE_W - 1) / TILE_W, (h + TILE_H - 1) / TILE_H);
    const size_t wf = window_floats(L, H, W, h, w);
    lfs::ProfScope prof("bilateral_slice_fwd", s);
    if (wf <= LDS_FLOATS) hipLaunchKernelGGL(slice_fwd_kernel<true>, g, dim3(THREADS), wf * sizeof(float), s, d, grid, rgb, output);
    else hipLaunchKernelGGL(slice_fwd_kernel<false>, g, dim3(THREADS), 0, s, d, grid, rgb, output);
    return (int)hipGetLastError();
}

extern "C" int lfs_bilateral_slice_bwd(uint32_t L, uint32_t H, uint32_t W, uint32_t h, uint32_t w, const float* grid, const float* rgb,
                                       const float* grad_output, uint32_t chw, uint32_t clamp_input, float* grad_grid, float* grad_rgb,
                                       lfs_stream_t stream) {
    if (int rc = check_dims(L, H, W, h, w)) return rc;
    if (!grid || !rgb || !grad_output || !grad_grid || !grad_rgb) return LFS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const Dims d{(int)L, (int)H, (int)W, (int)h, (int)w, chw ? 1u : 3u, chw ? h * w : 1u, (int)(clamp_input != 0)};
    // windowed path: the window (values + gradient) and the staging buffers fit the default 64 KB of LDS, the x-extent of the
    // window times L fits NT_MAX column tiles, and the packed stencil record holds its indices in 8 bits
    auto span = [](uint32_t tile, uint32_t n, uint32_t G) {
        uint32_t sp = (uint32_t)((double)(G - 1) / (double)(n - 1) * (tile - 1) + 1e-3) + 3;
        return sp < G ? sp : G;
    };
    const size_t wf = window_floats(L, H, W, h, w, BWD_TILE_H);
    const size_t lds = (2 * wf + 4 * 64 * STAGE_STRIDE) * sizeof(float);
    lfs::ProfScope prof("bilateral_slice_bwd", s);
    if (lds <= 64 * 1024 && span(TILE_W, w, W) * L <= 16 * NT_MAX && L <= 255) {
        const dim3 g((w + TILE_W - 1) / TILE_W, (h + BWD_TILE_H - 1) / BWD_TILE_H);
        hipLaunchKernelGGL(slice_bwd_window_kernel, g, dim3(THREADS), lds, s, d, grid, rgb, grad_output, grad_grid, grad_rgb);
    } else {
        const dim3 g((w + TILE_W - 1) / TILE_W, (h + TILE_H - 1) / TILE_H);
        hipLaunchKernelGGL(slice_bwd_generic_kernel, g, dim3(THREADS), 0, s, d, grid, rgb, grad_output, grad_grid, grad_rgb);
    }
    return (int)hipGetLastError();
}

static bool tv_scales(uint32_t N, uint32_t L, uint32_t H, uint32_t W, float k, float& sx, float& sy, float& sz) {
    if (!N || !L || !H || !W || (uint64_t)N * 12 * L * H * W >= (1ull << 40)) return false;
    // a direction of extent 1 has no differences: the reference divides by zero there but never adds the term
    sx = W > 1 ? k / ((float)L * H * (W - 1)) : 0.f;
    sy = H > 1 ? k / ((float)L * (H - 1) * W) : 0.f;
    sz = L > 1 ? k / ((float)(L - 1) * H * W) : 0.f;
    return true;
}

extern "C" int lfs_bilateral_tv_loss_fwd(uint32_t N, uint32_t L, uint32_t H, uint32_t W, const float* grids, float weight, float* loss,
                                         lfs_stream_t stream) {
    float sx, sy, sz;
    if (!grids || !loss || !tv_scales(N, L, H, W, weight / (12.f * N), sx, sy, sz)) return LFS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const size_t total = (size_t)N * 12 * L * H * W;
    const uint32_t blocks = (uint32_t)((total + 1023) / 1024 < 2048 ? (total + 1023) / 1024 : 2048);
    lfs::ProfScope prof("bilateral_tv_fwd", s);
    hipLaunchKernelGGL(tv_fwd_kernel, dim3(blocks), dim3(256), 0, s, grids, loss, N * 12, (int)L, (int)H, (int)W, sx, sy, sz);
    return (int)hipGetLastError();
}

extern "C" int lfs_bilateral_tv_loss_bwd(uint32_t N, uint32_t L, uint32_t H, uint32_t W, const float* grids, float grad_output, uint32_t accumulate,
                                         float* grad_grids, lfs_stream_t stream) {
    float sx, sy, sz;
    if (!grids || !grad_grids || !tv_scales(N, L, H, W, grad_output / (6.f * N), sx, sy, sz)) return LFS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const size_t total = (size_t)N * 12 * L * H * W;
    const uint32_t blocks = (uint32_t)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    lfs::ProfScope prof("bilateral_tv_bwd", s);
    if (accumulate) hipLaunchKernelGGL(tv_bwd_kernel<true>, dim3(blocks), dim3(256), 0, s, grids, grad_grids, N * 12, (int)L, (int)H, (int)W, sx, sy, sz);
    else hipLaunchKernelGGL(tv_bwd_kernel<false>, dim3(blocks), dim3(256), 0, s, grids, grad_grids, N * 12, (int)L, (int)H, (int)W, sx, sy, sz);
    return (int)hipGetLastError();
}
