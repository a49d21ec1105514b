// Workspace layout and constants shared by fastgs_prep.hip and fastgs_blend.hip (the EWA "fastgs" rasterizer,
// SURVEY.md §8f row 1; reference: fastgs/rasterization/*).
#pragma once
#include "lfs_raster_common.cuh"

namespace lfs {
// sh.hip: the SH kernels with strided colour operands (colours live inside the 64-B blend records / accumulator rows)
int sh_records_fwd(uint32_t n, uint32_t K, uint32_t degree, const float* means, const float* campos, const float* sh0, const float* shN,
                   const uint32_t* mask_u32, float* colors, uint32_t colors_stride, hipStream_t s);
// adam != NULL (single-view steps): v_shN is not written, shN and its moments are updated in place by the same kernel (sh_bwd_kernel<.., ADAM>)
struct ShAdamArgs { float* exp_avg; float* exp_avg_sq; float lr, beta1, beta2, eps, bc1_rcp, bc2_sqrt_rcp; };
int sh_records_bwd(uint32_t n, uint32_t K, uint32_t degree, const float* means, const float* campos, const float* sh0, const float* shN,
                   const uint32_t* mask_u32, const float* colors, uint32_t colors_stride, const float* v_colors, uint32_t v_stride,
                   float* v_sh0, float* v_shN, float* v_means, hipStream_t s, const ShAdamArgs* adam = nullptr);
namespace fgs {

// fastgs/rasterization/include/rasterization_config.h:14-33
constexpr float DILATION = 0.3f;
constexpr float MIN_ALPHA_RCP = 255.0f;
constexpr float MIN_ALPHA = 1.0f / 255.0f;
constexpr float MAX_ALPHA = 0.999f;
constexpr float T_THRESHOLD = 1e-4f;
constexpr uint32_t TILE = 16;
constexpr float LOG2E = 1.4426950408889634f;

struct Frame { // one camera; w2c / cam_position stay device pointers (torch tensors of the caller)
    const float* w2c; const float* cam_pos;
    uint32_t active_sh_bases, total_rest, width, height, gw, gh;
    float fx, fy, cx, cy, near_, far_;
};

inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

// Per-primitive + per-tile + per-pixel state: written by the forward, read by the backward.
//   rec[N]   64-B blend record: r0 = {mean2d.x, mean2d.y, A, B}, r1 = {C, thr', opacity, -}, r2 = {max(colour, 0), -},
//            r3 unused; with (A, B, C) = log2(e) * (conic.x / 2, conic.y, conic.z / 2) and thr' = log2(e) *
//            log(255 * opacity): sigma' = A dx^2 + B dx dy + C dy^2 is the Gaussian exponent in bits (one v_exp_f32).
struct PrimWs {
    GaussRec* rec; float2* mean2d; float4* conic_opacity; ushort4* bounds; uint32_t* n_touched; uint32_t* depth_bits;
    uint32_t* totals; uint32_t* cursor; int32_t* offsets; int64_t* n_instances; int32_t* n_contrib; float* acc; size_t bytes;
};
inline PrimWs prim_ws(void* base, uint32_t N, uint32_t width, uint32_t height) {
    const size_t T = size_t((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE), P = size_t(width) * height;
    PrimWs w; char* p = (char*)base; size_t o = 0;
    w.rec = (GaussRec*)(p + o); o += align256(sizeof(GaussRec) * N);
    w.mean2d = (float2*)(p + o); o += align256(sizeof(float2) * N);
    w.conic_opacity = (float4*)(p + o); o += align256(sizeof(float4) * N);
    w.bounds = (ushort4*)(p + o); o += align256(sizeof(ushort4) * N);
    w.n_touched = (uint32_t*)(p + o); o += align256(4 * size_t(N));
    w.depth_bits = (uint32_t*)(p + o); o += align256(4 * size_t(N));
    w.totals = (uint32_t*)(p + o); o += align256(4 * T);
    w.cursor = (uint32_t*)(p + o); o += align256(4 * T);
    w.offsets = (int32_t*)(p + o); o += align256(4 * (T + 1));
    w.n_instances = (int64_t*)(p + o); o += 256;
    w.n_contrib = (int32_t*)(p + o); o += align256(4 * P);
    w.acc = (float*)(p + o); o += align256(sizeof(float) * ACC_STRIDE * N);
    w.bytes = o;
    return w;
}
// Per-instance state: keys (depth bits << 32 | primitive) -> sorted ids, and the compacted per-8x8-cell lists.
struct InstWs { int64_t* keys; int32_t* ids; int32_t* cell_count; int2* cell_list; size_t bytes; };
inline InstWs inst_ws(void* base, uint32_t width, uint32_t height, uint64_t n_instances) {
    const size_t T = size_t((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
    InstWs w; char* p = (char*)base; size_t o = 0;
    w.keys = (int64_t*)(p + o); o += align256(8 * n_instances);
    w.ids = (int32_t*)(p + o); o += align256(4 * n_instances);
    w.cell_count = (int32_t*)(p + o); o += align256(4 * 4 * T);
    w.cell_list = (int2*)(p + o); o += align256(8 * 4 * n_instances);
    w.bytes = o;
    return w;
}

// accumulator row of the blend backward (16 floats per primitive, one 64-B line):
//   0,1 dL/dmean2d * log2(e) | 2,3,4 dL/dconic (xx, xy, yy) | 5 sum alpha dL/dalpha | 6,7,8 dL/d(clamped colour) | 9..15 unused
} // namespace fgs
} // namespace lfs
