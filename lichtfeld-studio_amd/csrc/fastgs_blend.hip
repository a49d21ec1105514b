// The reference's default training rasterizer ("fastgs", EWA splatting) — per-pixel stages:
// alpha blending forward (replaces blend_cu, fastgs/rasterization/include/kernels_forward.cuh:353-459) and backward
// (blend_backward_cu, kernels_backward.cuh:236-448), with the same CDNA4 structure as raster.hip: per-8x8-cell culling of
// the tile lists (here with the reference's own exact ellipse/rectangle test, kernel_utils.cuh:108-148, at cell
// granularity), one wavefront per cell walking its list with the 64-byte records arriving through the scalar unit, and a
// transposed wave reduction + one 64-byte atomic per (cell, primitive) in the backward. The reference instead keeps
// per-32-primitive (colour, transmittance) checkpoints of every pixel ("buckets", 16 B x 256 pixels per bucket) and runs the
// backward primitive-parallel; the sums are the same, the traversal here is back to front from the final transmittance.
#include "lfs_fastgs.cuh"
#include "lfs_prof.h"
#include "lfs_tilelists.cuh"
#include "../../include/lfs_gsplat.h"

namespace lfs {
namespace fgs {

int launch_scatter(uint32_t N, const Frame& f, const PrimWs& w, int64_t* keys, hipStream_t s);
int launch_preprocess_bwd(uint32_t N, const float* means, const float* scales_raw, const float* rot_raw, const float* sh0, const float* sh_rest, const Frame& f, const PrimWs& w,
                          float* g_means, float* g_scales_raw, float* g_rot_raw, float* g_opac_raw, float* g_sh0, float* g_sh_rest, float* densification_info, hipStream_t s,
                          const ShAdamArgs* adam);

// cell-level version of kernel_utils.cuh:108-148 on the record's conic in bits (A, B, C) = log2(e) (a/2, b, c/2): the ratios that
// locate the maximum are scale free; thr already carries the safety margin
LFS_DI bool cell_reachable(float mx, float my, float A, float B, float C, float rx0, float ry0, float thr) {
    const float rx1 = rx0 + 7.f, ry1 = ry0 + 7.f;
    const float x_min_diff = rx0 - mx, y_min_diff = ry0 - my;
    const float x_left = x_min_diff > 0.f ? 1.f : 0.f, y_above = y_min_diff > 0.f ? 1.f : 0.f;
    const float not_in_x = x_left + (mx > rx1 ? 1.f : 0.f), not_in_y = y_above + (my > ry1 ? 1.f : 0.f);
    if (not_in_x + not_in_y == 0.f) return true;
    const float ccx = x_left > 0.f ? rx0 : rx1, ccy = y_above > 0.f ? ry0 : ry1;
    const float dfx = mx - ccx, dfy = my - ccy;
    const float dx = copysignf(7.f, x_min_diff), dy = copysignf(7.f, y_min_diff);
    const float tx = not_in_y * __saturatef((2.f * A * dfx + B * dfy) / (2.f * A * dx));
    const float ty = not_in_x * __saturatef((B * dfx + 2.f * C * dfy) / (2.f * C * dy));
    const float ddx = mx - (ccx + tx * dx), ddy = my - (ccy + ty * dy);
    return !(A * ddx * ddx + C * ddy * ddy + B * ddx * ddy > thr); // NaN -> keep
}

constexpr uint32_t WPT = 4; // 8x8 cells per 16x16 tile

__global__ void __launch_bounds__(256) fg_cull_kernel(
    const uint32_t gw, const uint32_t gh, const uint32_t width, const uint32_t height, const uint32_t cull_enabled,
    const GaussRec* __restrict__ recs, const int32_t* __restrict__ offsets, const int32_t* __restrict__ ids,
    int32_t* __restrict__ cell_count, int2* __restrict__ cell_list) {
    __shared__ float4 s_a[2][256], s_b[2][256];
    __shared__ int32_t s_g[2][256];
    const uint32_t total_tiles = gw * gh;
    const CellCtx cc = cell_ctx(total_tiles, total_tiles, gw, TILE, 1, 4);
    if (!cc.in_grid) return;
    const uint32_t lane = threadIdx.x & 63;
    const size_t cell = size_t(cc.tile_global) * WPT + cc.wl;
    const int32_t start = offsets[cc.tile_global], end = offsets[cc.tile_global + 1];
    if (end <= start) {
        if (lane == 0) cell_count[cell] = 0;
        return;
    }
    const uint32_t x0 = (cc.tile_global % gw) * TILE + (cc.wl & 1) * 8, y0 = (cc.tile_global / gw) * TILE + (cc.wl >> 1) * 8;
    const bool cell_live = x0 < width && y0 < height;
    const float rx0 = float(x0), ry0 = float(y0);
    int2* __restrict__ out = cell_list + (size_t(WPT) * size_t(start) + size_t(cc.wl) * size_t(end - start));
    int32_t count = 0;
    auto fetch = [&](int32_t base, int32_t& g, float4& a, float4& b) {
        const int32_t i = base + int32_t(threadIdx.x);
        g = i < end ? ids[i] : 0;
        const float4* r = reinterpret_cast<const float4*>(recs + g);
        a = r[0]; b = r[1];
    };
    int32_t g_reg; float4 a_reg, b_reg;
    fetch(start, g_reg, a_reg, b_reg);
    int buf = 0;
    for (int32_t base = start; base < end; base += 256, buf ^= 1) {
        s_g[buf][threadIdx.x] = g_reg; s_a[buf][threadIdx.x] = a_reg; s_b[buf][threadIdx.x] = b_reg;
        __syncthreads();
        if (base + 256 < end) fetch(base + 256, g_reg, a_reg, b_reg);
        if (!cell_live) continue;
        for (int32_t sub = 0; sub < 4; ++sub) {
            if (base + (sub << 6) >= end) break;
            const int32_t slot = (sub << 6) + int32_t(lane), my_idx = base + slot;
            const bool valid = my_idx < end;
            const int32_t my_g = s_g[buf][slot];
            bool hit = valid;
            if (cull_enabled) {
                const float4 a = s_a[buf][slot], b = s_b[buf][slot];
                hit = valid && cell_reachable(a.x - 0.5f, a.y - 0.5f, a.z, a.w, b.x, rx0, ry0, b.y * 1.001f + 1e-3f);
            }
            const uint64_t m = __ballot(hit);
            if (hit) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
                out[count + int32_t(rank)] = make_int2(my_g, my_idx);
            }
            count += __popcll(m);
        }
    }
    if (lane == 0) cell_count[cell] = count;
}

struct Frag { float dx, dy, alpha; bool ok; };
// one primitive against this lane's pixel: sigma' = A dx^2 + B dx dy + C dy^2 (bits), alpha = min(opacity 2^-sigma', 0.999)
LFS_DI Frag frag_eval(const GaussRec& rec, const float px, const float py) {
    Frag f;
    f.dx = rec.r0.x - px; f.dy = rec.r0.y - py;
    const float u = __builtin_fmaf(rec.r0.w, f.dy, rec.r0.z * f.dx);
    const float s = __builtin_fmaf(rec.r1.x * f.dy, f.dy, f.dx * u);
    f.alpha = fminf(rec.r1.z * __builtin_amdgcn_exp2f(-s), MAX_ALPHA);
    f.ok = !(s < 0.f); // kernels_forward.cuh:425-426
    return f;
}

// fwd / bwd: ONE wavefront per workgroup (the cells of a tile never cooperate here; a finished cell frees its slot at once - as raster.hip's wave_geom)
__global__ void __launch_bounds__(64) fg_blend_fwd_kernel(
    const uint32_t gw, const uint32_t gh, const uint32_t width, const uint32_t height,
    const GaussRec* __restrict__ recs, const int32_t* __restrict__ offsets, const int32_t* __restrict__ cell_count, const int2* __restrict__ cell_list,
    float* __restrict__ image, float* __restrict__ alpha_map, int32_t* __restrict__ n_contrib) {
    const uint32_t total_tiles = gw * gh;
    const CellCtx cc = cell_ctx(total_tiles, total_tiles, gw, TILE, WPT, 1);
    if (!cc.in_grid) return;
    const bool inside = cc.i < height && cc.j < width;
    const float px = float(cc.j) + 0.5f, py = float(cc.i) + 0.5f;
    const int32_t start = offsets[cc.tile_global], end = offsets[cc.tile_global + 1];
    const int2* __restrict__ cl = cell_list + (size_t(WPT) * size_t(start) + size_t(cc.wl) * size_t(end - start));
    const int32_t cnt = cell_count[size_t(cc.tile_global) * WPT + cc.wl];
    const float INF = __builtin_inff();
    float thr = inside ? MIN_ALPHA : INF; // "done" is carried as the alpha threshold (see raster.hip)
    float T = 1.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
    int32_t last = 0; // 1 + global list index of the last blended instance (0: none)
    auto eval = [&](const GaussRec& rec, const int2 e) {
        const Frag f = frag_eval(rec, px, py);
        const bool pass = f.ok && !(f.alpha < thr);
        if (__ballot(pass) == 0ull) return;
        const float next_T = T * (1.f - f.alpha);
        const bool fin = pass && next_T < T_THRESHOLD; // kernels_forward.cuh:432-436: terminates without blending
        const bool contrib = pass && !fin;
        const float w = T * f.alpha;
        if (contrib) {
            c0 = __builtin_fmaf(rec.r2.x, w, c0); c1 = __builtin_fmaf(rec.r2.y, w, c1); c2 = __builtin_fmaf(rec.r2.z, w, c2);
            T = next_T; last = e.y + 1;
        }
        thr = fin ? INF : thr;
    };
    walk_cell_list<1>(cl, recs, 0, cnt, eval, [&]() { return __ballot(thr < INF) != 0ull; });
    if (inside) {
        const size_t np = size_t(width) * height, p = size_t(cc.i) * width + cc.j;
        image[p] = c0; image[np + p] = c1; image[2 * np + p] = c2;
        alpha_map[p] = 1.f - T;
        n_contrib[p] = last;
    }
}

#ifndef LFS_FG_LDS_REDUCE
#define LFS_FG_LDS_REDUCE 1   // the nine wave sums through an LDS transpose (lfs_raster_common.cuh), as the 3DGUT backward does with its sixteen. Same-box A/B x2
                              // (profiles/r03/fastgs_blend_bwd_lds_reduce_ab.txt): fastgs_blend_bwd 0.452 - 0.457 -> 0.391 - 0.395 ms; 0 = register swaps + DPP
#endif
__global__ void __launch_bounds__(64) fg_blend_bwd_kernel(
    const uint32_t gw, const uint32_t gh, const uint32_t width, const uint32_t height,
    const GaussRec* __restrict__ recs, const int32_t* __restrict__ offsets, const int32_t* __restrict__ cell_count, const int2* __restrict__ cell_list,
    const float* __restrict__ alpha_map, const int32_t* __restrict__ n_contrib, const float* __restrict__ g_image, const float* __restrict__ g_alpha,
    float* __restrict__ acc) {
    const uint32_t total_tiles = gw * gh;
#if LFS_FG_LDS_REDUCE
    __shared__ __attribute__((aligned(16))) float s_red[RED9_SCRATCH_FLOATS]; // this wavefront's transpose block (wave_sum9_atomic_lds)
    float* const red_scratch = s_red;
#endif
    const CellCtx cc = cell_ctx(total_tiles, total_tiles, gw, TILE, WPT, 1);
    if (!cc.in_grid) return;
    const uint32_t lane = threadIdx.x & 63;
    const bool inside = cc.i < height && cc.j < width;
    const float px = float(cc.j) + 0.5f, py = float(cc.i) + 0.5f;
    const int32_t start = offsets[cc.tile_global], end = offsets[cc.tile_global + 1];
    const int2* __restrict__ cl = cell_list + (size_t(WPT) * size_t(start) + size_t(cc.wl) * size_t(end - start));
    const int32_t cnt = cell_count[size_t(cc.tile_global) * WPT + cc.wl];
    float T = 1.f, gc0 = 0.f, gc1 = 0.f, gc2 = 0.f, tail = 0.f, Bsum = 0.f;
    int32_t last = -1;
    if (inside) {
        const size_t np = size_t(width) * height, p = size_t(cc.i) * width + cc.j;
        T = 1.f - alpha_map[p];
        last = n_contrib[p] - 1;
        gc0 = g_image[p]; gc1 = g_image[np + p]; gc2 = g_image[2 * np + p];
        tail = g_alpha[p] * T; // grad_alpha * (1 - alpha_pixel), kernels_backward.cuh:352
    }
    int32_t wmax = last;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) wmax = max(wmax, __shfl_xor(wmax, m, 64));
    wmax = __builtin_amdgcn_readfirstlane(wmax);
    int32_t lo = 0, hi = cnt;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (cl[mid].y <= wmax) lo = mid + 1; else hi = mid;
    }
    if (lo <= 0) return;
    auto eval = [&](const GaussRec& rec, const int2 e) {
        const Frag f = frag_eval(rec, px, py);
        const bool valid = e.y <= last && f.ok && !(f.alpha < MIN_ALPHA);
        if (__ballot(valid) == 0ull) return;
        const float ra = fast_rcp(1.f - f.alpha);
        const float Tn = T * ra;                                  // transmittance in front of this instance
        T = valid ? Tn : T;
        const float w = valid ? Tn * f.alpha : 0.f;               // blending weight (0 masks every sum below)
        const float cv = __builtin_fmaf(rec.r2.z, gc2, __builtin_fmaf(rec.r2.y, gc1, rec.r2.x * gc0));
        // dL/dalpha = (T c - colour_behind / (1 - alpha)) . g + grad_alpha (1 - alpha_pixel) / (1 - alpha)   (kernels_backward.cuh:412-417)
        const float dl_dalpha = __builtin_fmaf(ra, tail - Bsum, Tn * cv);
        Bsum = __builtin_fmaf(w, cv, Bsum);
        const float aD = valid ? f.alpha * dl_dalpha : 0.f;      // (no clamp mask on alpha, as in the reference)
        const float hx = -aD * f.dx, hy = -aD * f.dy;
        // accumulator row: {sum hx, sum hy, sum hx dx, sum hx dy | sum hy dy, dc.r, dc.g, dc.b | sum alpha dL/dalpha}
        //   dL/dmean2d = conic . (sum hx, sum hy), dL/dconic = 0.5 (sum hx dx, sum hx dy, sum hy dy)   (fastgs_prep.hip)
        float* row = acc + size_t(e.x) * ACC_STRIDE;
#if LFS_FG_LDS_REDUCE
        const float v[9] = {hx, hy, hx * f.dx, hx * f.dy, hy * f.dy, w * gc0, w * gc1, w * gc2, aD};
        wave_sum9_atomic_lds(v, row, lane, red_scratch);
#else
        const float v[8] = {hx, hy, hx * f.dx, hx * f.dy, hy * f.dy, w * gc0, w * gc1, w * gc2};
        wave_sum8_atomic(v, row, lane);
        const float tot = wave_sum1(aD);
        if (lane == 0) unsafeAtomicAdd(row + 8, tot);
#endif
    };
    walk_cell_list<-1>(cl, recs, lo - 1, lo, eval, []() { return true; });
}

static uint32_t g_fastgs_debug = 0;

} // namespace fgs
} // namespace lfs

using namespace lfs;

extern "C" void lfs_fastgs_set_debug_flags(uint32_t flags) { fgs::g_fastgs_debug = flags; }

static fgs::Frame make_frame(const float* w2c, const float* cam_position, uint32_t active_sh_bases, uint32_t total_rest, uint32_t width, uint32_t height,
                             float fx, float fy, float cx, float cy, float near_plane, float far_plane) {
    return fgs::Frame{w2c, cam_position, active_sh_bases, total_rest, width, height, (width + fgs::TILE - 1) / fgs::TILE, (height + fgs::TILE - 1) / fgs::TILE,
                      fx, fy, cx, cy, near_plane, far_plane};
}

extern "C" int lfs_fastgs_render(
    uint32_t N, uint32_t width, uint32_t height, int64_t n_instances, void* primitive_workspace, size_t primitive_workspace_bytes,
    void* instance_workspace, size_t instance_workspace_bytes, float* image, float* alpha, lfs_stream_t stream) {
    if (!primitive_workspace || !image || !alpha || width == 0 || height == 0 || n_instances < 0 || n_instances > 0x7FFFFFFFll) return LFS_E_INVALID;
    if (N >= (1u << 26) || uint64_t(n_instances) >= (1ull << 29)) return LFS_E_UNSUPPORTED; // 32-bit byte offsets of the record walker
    fgs::PrimWs w = fgs::prim_ws(primitive_workspace, N, width, height);
    if (primitive_workspace_bytes < w.bytes) return LFS_E_WORKSPACE;
    fgs::InstWs iw = fgs::inst_ws(instance_workspace, width, height, uint64_t(n_instances));
    if (!instance_workspace || instance_workspace_bytes < iw.bytes) return LFS_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const fgs::Frame f = make_frame(nullptr, nullptr, 1, 0, width, height, 0, 0, 0, 0, 0, 0);
    const uint32_t T = f.gw * f.gh;
    if (n_instances > 0) {
        int rc = fgs::launch_scatter(N, f, w, iw.keys, s);
        if (rc) return rc;
        static bool big_lds_enabled = false;
        if (!big_lds_enabled) {
            hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_sort_lds_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
            if (ae != hipSuccess) return (int)ae;
            ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_sort_bins_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4096 * 8);
            if (ae != hipSuccess) return (int)ae;
            big_lds_enabled = true;
        }
        lfs::ProfScope prof_sort("fastgs_tile_sort", s);
        hipLaunchKernelGGL(tile_sort_bins_kernel<256>, dim3(T), dim3(256), 2 * 1024 * 8, s, 1u, 1024u, T, 0u, w.offsets, iw.keys, iw.ids);
        hipLaunchKernelGGL((tile_sort_bins_kernel<512, 512, false, 32>), dim3(T), dim3(512), 4096 * 8, s, 1025u, 4096u, T, 0u, w.offsets, iw.keys, iw.ids); // (as intersect.hip)
        hipLaunchKernelGGL(tile_sort_lds_kernel<1024>, dim3(T), dim3(1024), 16384 * 8, s, 4097u, 16384u, T, 0u, w.offsets, iw.keys, iw.ids);
        hipLaunchKernelGGL(tile_sort_global_kernel, dim3(T), dim3(1024), 0, s, 16385u, T, 0u, w.offsets, iw.keys, iw.ids);
    }
    const uint32_t grid = cell_grid_blocks(T, 1);
    {
        lfs::ProfScope prof("fastgs_cull", s);
        hipLaunchKernelGGL(fgs::fg_cull_kernel, dim3(grid), dim3(256), 0, s, f.gw, f.gh, width, height, (fgs::g_fastgs_debug & 1u) ? 0u : 1u,
                           w.rec, w.offsets, iw.ids, iw.cell_count, iw.cell_list);
    }
    lfs::ProfScope prof("fastgs_blend_fwd", s);
    const uint32_t wgrid = cell_grid_blocks(uint64_t(T) * fgs::WPT, fgs::WPT);
    hipLaunchKernelGGL(fgs::fg_blend_fwd_kernel, dim3(wgrid), dim3(64), 0, s, f.gw, f.gh, width, height, w.rec, w.offsets, iw.cell_count, iw.cell_list,
                       image, alpha, w.n_contrib);
    return (int)hipGetLastError();
}

static int fastgs_backward_impl(
    uint32_t N, const float* means, const float* scales_raw, const float* rotations_raw, const float* sh_coefficients_0, const float* sh_coefficients_rest,
    uint32_t total_bases_sh_rest, const float* w2c, const float* cam_position, uint32_t active_sh_bases, uint32_t width, uint32_t height, float fx, float fy,
    float cx, float cy, float near_plane, float far_plane, int64_t n_instances, void* primitive_workspace, size_t primitive_workspace_bytes,
    void* instance_workspace, size_t instance_workspace_bytes, const float* grad_image, const float* grad_alpha, const float* alpha,
    float* densification_info, float* grad_means, float* grad_scales_raw, float* grad_rotations_raw, float* grad_opacities_raw,
    float* grad_sh_coefficients_0, float* grad_sh_coefficients_rest, lfs_stream_t stream, const lfs::ShAdamArgs* adam) {
    if (!primitive_workspace || !w2c || !cam_position || !grad_image || !grad_alpha || !alpha || width == 0 || height == 0 || n_instances < 0) return LFS_E_INVALID;
    fgs::PrimWs w = fgs::prim_ws(primitive_workspace, N, width, height);
    if (primitive_workspace_bytes < w.bytes) return LFS_E_WORKSPACE;
    fgs::InstWs iw = fgs::inst_ws(instance_workspace, width, height, uint64_t(n_instances));
    if (!instance_workspace || instance_workspace_bytes < iw.bytes) return LFS_E_WORKSPACE;
    if (N == 0) return LFS_OK;
    if (!means || !scales_raw || !rotations_raw || !grad_means || !grad_scales_raw || !grad_rotations_raw || !grad_opacities_raw || !grad_sh_coefficients_0 ||
        (total_bases_sh_rest > 0 && (!sh_coefficients_rest || (!grad_sh_coefficients_rest && !adam)))) return LFS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const fgs::Frame f = make_frame(w2c, cam_position, active_sh_bases, total_bases_sh_rest, width, height, fx, fy, cx, cy, near_plane, far_plane);
    hipError_t e = hipMemsetAsync(w.acc, 0, sizeof(float) * ACC_STRIDE * size_t(N), s);
    if (e != hipSuccess) return (int)e;
    const uint32_t T = f.gw * f.gh;
    if (n_instances > 0) {
        lfs::ProfScope prof("fastgs_blend_bwd", s);
        const uint32_t wgrid = cell_grid_blocks(uint64_t(T) * fgs::WPT, fgs::WPT);
        hipLaunchKernelGGL(fgs::fg_blend_bwd_kernel, dim3(wgrid), dim3(64), 0, s, f.gw, f.gh, width, height, w.rec, w.offsets, iw.cell_count, iw.cell_list,
                           alpha, w.n_contrib, grad_image, grad_alpha, w.acc);
    }
    return fgs::launch_preprocess_bwd(N, means, scales_raw, rotations_raw, sh_coefficients_0, sh_coefficients_rest, f, w, grad_means, grad_scales_raw, grad_rotations_raw,
                                      grad_opacities_raw, grad_sh_coefficients_0, grad_sh_coefficients_rest, densification_info, s, adam);
}

extern "C" int lfs_fastgs_backward(
    uint32_t N, const float* means, const float* scales_raw, const float* rotations_raw, const float* sh_coefficients_0, const float* sh_coefficients_rest,
    uint32_t total_bases_sh_rest, const float* w2c, const float* cam_position, uint32_t active_sh_bases, uint32_t width, uint32_t height, float fx, float fy,
    float cx, float cy, float near_plane, float far_plane, int64_t n_instances, void* primitive_workspace, size_t primitive_workspace_bytes,
    void* instance_workspace, size_t instance_workspace_bytes, const float* grad_image, const float* grad_alpha, const float* alpha,
    float* densification_info, float* grad_means, float* grad_scales_raw, float* grad_rotations_raw, float* grad_opacities_raw,
    float* grad_sh_coefficients_0, float* grad_sh_coefficients_rest, lfs_stream_t stream) {
    return fastgs_backward_impl(N, means, scales_raw, rotations_raw, sh_coefficients_0, sh_coefficients_rest, total_bases_sh_rest, w2c, cam_position, active_sh_bases,
                                width, height, fx, fy, cx, cy, near_plane, far_plane, n_instances, primitive_workspace, primitive_workspace_bytes, instance_workspace,
                                instance_workspace_bytes, grad_image, grad_alpha, alpha, densification_info, grad_means, grad_scales_raw, grad_rotations_raw,
                                grad_opacities_raw, grad_sh_coefficients_0, grad_sh_coefficients_rest, stream, nullptr);
}

// lfs_fastgs_backward for a step with ONE view, fused with the optimizer: sh_coefficients_rest and its Adam moments are updated in place by the SH
// backward (fast_gs::optimizer::adam_step arithmetic), its gradient is never stored. Everything else as lfs_fastgs_backward.
extern "C" int lfs_fastgs_backward_adam(
    uint32_t N, const float* means, const float* scales_raw, const float* rotations_raw, const float* sh_coefficients_0, float* sh_coefficients_rest,
    uint32_t total_bases_sh_rest, const float* w2c, const float* cam_position, uint32_t active_sh_bases, uint32_t width, uint32_t height, float fx, float fy,
    float cx, float cy, float near_plane, float far_plane, int64_t n_instances, void* primitive_workspace, size_t primitive_workspace_bytes,
    void* instance_workspace, size_t instance_workspace_bytes, const float* grad_image, const float* grad_alpha, const float* alpha,
    float* densification_info, float* grad_means, float* grad_scales_raw, float* grad_rotations_raw, float* grad_opacities_raw,
    float* grad_sh_coefficients_0, float* sh_rest_exp_avg, float* sh_rest_exp_avg_sq, float lr, float beta1, float beta2, float eps,
    float bias_correction1_rcp, float bias_correction2_sqrt_rcp, lfs_stream_t stream) {
    if (total_bases_sh_rest == 0 || !sh_rest_exp_avg || !sh_rest_exp_avg_sq) return LFS_E_INVALID;
    const lfs::ShAdamArgs adam{sh_rest_exp_avg, sh_rest_exp_avg_sq, lr, beta1, beta2, eps, bias_correction1_rcp, bias_correction2_sqrt_rcp};
    return fastgs_backward_impl(N, means, scales_raw, rotations_raw, sh_coefficients_0, sh_coefficients_rest, total_bases_sh_rest, w2c, cam_position, active_sh_bases,
                                width, height, fx, fy, cx, cy, near_plane, far_plane, n_instances, primitive_workspace, primitive_workspace_bytes, instance_workspace,
                                instance_workspace_bytes, grad_image, grad_alpha, alpha, densification_info, grad_means, grad_scales_raw, grad_rotations_raw,
                                grad_opacities_raw, grad_sh_coefficients_0, nullptr, stream, &adam);
}
