// Data-side kernels of SURVEY.md §8f row 4 (what the reference does on the host before / around training):
//   * lfs_image_u8_to_chw_f32 — Camera::load_and_get_image (src/core/camera.cpp:101-140: u8 [h,w,3] -> permute -> float / 255)
//     fused with load_image's downscale (src/core/image_io.cpp:33-57: OpenImageIO ImageBufAlgo::resample(interpolate = true),
//     i.e. a bilinear tap at the destination pixel centre, clamped at the borders, rounded back to u8). The reference
//     resamples on the CPU and uploads the small image; here the full-size u8 image is uploaded once and one kernel writes
//     the training target: no host resample, no intermediate tensors.
//   * lfs_mean_neighbor_distances — compute_mean_neighbor_distances (src/core/splat_data.cpp:64-111): mean distance to the (up
//     to) 3 nearest neighbours among the 4 nearest results with d^2 > 1e-8, the initial Gaussian scale of
//     init_model_from_pointcloud (:550-555). The reference builds a CPU kd-tree (nanoflann); here it is an exact tiled
//     all-pairs search: candidates stream through the scalar cache into SGPRs, every thread keeps the 4 smallest squared
//     distances of its query in registers. O(N^2) but VALU-dense (3 sub, 3 mul, 2 add + a rarely taken insertion per pair): 1M points take
//     well under a second once per run, and the result does not depend on tree construction order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lfs_gsplat.h"
#include "lfs_prof.h"

namespace lfs {
namespace dataprep {

// OpenImageIO resample, interpolate = true: destination pixel (x, y) samples the source at
//   sx = (x + 0.5) / dw * sw,  sy = (y + 0.5) / dh * sh      (continuous image coordinates, pixel centres at +0.5)
// bilinearly: shift by -0.5, floor / frac, 2x2 texels with clamp addressing; the u8 source is read as v / 255 and the result
// stored as u8 with round-to-nearest (v * 255 + 0.5 truncated, clamped).
__global__ void __launch_bounds__(256) image_to_chw_kernel(const uint8_t* __restrict__ src, int sw, int sh, float* __restrict__ dst, int dw, int dh,
                                                           int resample) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    const size_t plane = (size_t)dw * dh, o = (size_t)y * dw + x;
    if (!resample) {
        const uint8_t* p = src + o * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[c * plane + o] = (float)p[c] / 255.0f;
        return;
    }
    const float s = ((float)x + 0.5f) * (1.0f / (float)dw), t = ((float)y + 0.5f) * (1.0f / (float)dh);
    const float fx = s * (float)sw - 0.5f, fy = t * (float)sh - 0.5f;
    const float flx = floorf(fx), fly = floorf(fy);
    const float ax = fx - flx, ay = fy - fly;
    const int x0 = min(max((int)flx, 0), sw - 1), x1 = min(max((int)flx + 1, 0), sw - 1);
    const int y0 = min(max((int)fly, 0), sh - 1), y1 = min(max((int)fly + 1, 0), sh - 1);
    const uint8_t* p00 = src + ((size_t)y0 * sw + x0) * 3;
    const uint8_t* p01 = src + ((size_t)y0 * sw + x1) * 3;
    const uint8_t* p10 = src + ((size_t)y1 * sw + x0) * 3;
    const uint8_t* p11 = src + ((size_t)y1 * sw + x1) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v00 = (float)p00[c] * (1.0f / 255.0f), v01 = (float)p01[c] * (1.0f / 255.0f);
        const float v10 = (float)p10[c] * (1.0f / 255.0f), v11 = (float)p11[c] * (1.0f / 255.0f);
        // bilerp as OIIO's: (1-ay) * ((1-ax) v00 + ax v01) + ay * ((1-ax) v10 + ax v11)
        const float top = (1.0f - ax) * v00 + ax * v01, bot = (1.0f - ax) * v10 + ax * v11;
        const float v = (1.0f - ay) * top + ay * bot;
        const float q = fminf(fmaxf(v * 255.0f + 0.5f, 0.0f), 255.0f);
        dst[c * plane + o] = (float)(int)q / 255.0f;
    }
}

constexpr int KNN_THREADS = 256;

// sorted insert into the 4 smallest (ascending); ties keep the earlier entry first, as a stable selection does
__device__ __forceinline__ void knn_insert(float d, float b[4]) {
    if (d < b[0]) { b[3] = b[2]; b[2] = b[1]; b[1] = b[0]; b[0] = d; }
    else if (d < b[1]) { b[3] = b[2]; b[2] = b[1]; b[1] = d; }
    else if (d < b[2]) { b[3] = b[2]; b[2] = d; }
    else b[3] = d;
}

// The candidate index is wave-uniform, so candidates are fetched with scalar loads (s_load_dwordx*, through the scalar cache) and
// enter the VALU as SGPR operands: no LDS staging, no vector memory traffic in the inner loop.
__global__ void __launch_bounds__(KNN_THREADS) knn_mean_distance_kernel(uint32_t N, const float* __restrict__ pts, float* __restrict__ out) {
    const uint32_t q = blockIdx.x * KNN_THREADS + threadIdx.x;
    const bool live = q < N;
    const uint32_t qc = live ? q : N - 1;
    const float qx = pts[3 * (size_t)qc], qy = pts[3 * (size_t)qc + 1], qz = pts[3 * (size_t)qc + 2];
    float best[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    const uint32_t n4 = N & ~3u;
    for (uint32_t i = 0; i < n4; i += 4) { // 12 consecutive floats: three s_load_dwordx4
        float c[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) c[k] = pts[3 * (size_t)i + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dx = qx - c[3 * k], dy = qy - c[3 * k + 1], dz = qz - c[3 * k + 2];
            const float d = (dx * dx + dy * dy) + dz * dz; // nanoflann L2_Simple: accumulate in dimension order
            if (d < best[3]) knn_insert(d, best);
        }
    }
    for (uint32_t i = n4; i < N; ++i) {
        const float dx = qx - pts[3 * (size_t)i], dy = qy - pts[3 * (size_t)i + 1], dz = qz - pts[3 * (size_t)i + 2];
        const float d = (dx * dx + dy * dy) + dz * dz;
        if (d < best[3]) knn_insert(d, best);
    }
    if (!live) return;
    // the query itself (d = 0) and exact duplicates are in the list: skip d^2 <= 1e-8, take up to 3 (splat_data.cpp:96-107)
    float sum = 0.f;
    int valid = 0;
    const int results = N < 4 ? (int)N : 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < results && valid < 3 && best[j] > 1e-8f) { sum += sqrtf(best[j]); ++valid; }
    out[q] = valid > 0 ? sum / (float)valid : 0.01f;
}

} // namespace dataprep
} // namespace lfs

using namespace lfs::dataprep;

extern "C" int lfs_image_u8_to_chw_f32(const uint8_t* src_hwc, uint32_t src_width, uint32_t src_height, float* dst_chw, uint32_t dst_width,
                                       uint32_t dst_height, lfs_stream_t stream) {
    if (!dst_width || !dst_height) return LFS_OK;
    if (!src_hwc || !dst_chw || !src_width || !src_height || src_width >= (1u << 30) || src_height >= (1u << 30)) return LFS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    lfs::ProfScope prof("image_to_chw", s);
    const int resample = (dst_width != src_width || dst_height != src_height) ? 1 : 0;
    hipLaunchKernelGGL(image_to_chw_kernel, dim3((dst_width + 63) / 64, (dst_height + 3) / 4), dim3(256), 0, s, src_hwc, (int)src_width, (int)src_height, dst_chw,
                       (int)dst_width, (int)dst_height, resample);
    return (int)hipGetLastError();
}

extern "C" int lfs_mean_neighbor_distances(uint32_t N, const float* points, float* out, lfs_stream_t stream) {
    if (!N) return LFS_OK;
    if (!points || !out) return LFS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    lfs::ProfScope prof("mean_neighbor_distances", s);
    // N == 1: the only result is the query itself (d = 0, skipped) -> 0.01, the reference's "num_points <= 1" case (splat_data.cpp:72-74)
    hipLaunchKernelGGL(knn_mean_distance_kernel, dim3((N + KNN_THREADS - 1) / KNN_THREADS), dim3(KNN_THREADS), 0, s, N, points, out);
    return (int)hipGetLastError();
}
