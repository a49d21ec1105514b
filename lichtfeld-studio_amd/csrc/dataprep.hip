// Data-side kernels of SURVEY.md §8f row 4 (what the reference does on the host before / around training):
//   * lfs_image_u8_to_chw_f32 — Camera::load_and_get_image (src/core/camera.cpp:101-140: u8 [h,w,3] -> permute -> float / 255)
//     fused with load_image's downscale (src/core/image_io.cpp:33-57: OpenImageIO ImageBufAlgo::resample(interpolate = true),
//     i.e. a bilinear tap at the destination pixel centre, clamped at the borders, rounded back to u8). The reference
//     resamples on the CPU and uploads the small image; here the full-size u8 image is uploaded once and one kernel writes
//     the training target: no host resample, no intermediate tensors.
//   * lfs_mean_neighbor_distances — compute_mean_neighbor_distances (src/core/splat_data.cpp:64-111): mean distance to the (up
//     to) 3 nearest neighbours among the 4 results of a kd-tree query with d^2 > 1e-8, the initial Gaussian scale of
//     init_model_from_pointcloud (:550-555). The reference queries nanoflann (vendored, v1.7.1) with SearchParameters(10), and that
//     first argument is `eps`: the search is (1 + 10)-APPROXIMATE - a far subtree is entered only if 11 x its lower bound is still
//     within the current 4th-best squared distance. Running the reference's function (oracle/_ref/libref_splat_io.so) shows what that
//     means: half of the points get a larger value than the exact 3-NN mean (+7 % on average, up to 2x). The result therefore
//     depends on the tree and on the visiting order, and both are reproduced here: the tree is built on the host exactly as
//     nanoflann's single-threaded divideTree / middleSplit_ / planeSplit do (leaf size 10; a sequential algorithm - the permutation
//     of the index array is part of the result), uploaded as a flat node array, and one GPU thread per query walks it in
//     searchLevel's order with an explicit stack. Bit-identical to the reference's output (tests/golden/ref_splat_io.npz).
//   * lfs_mean_neighbor_distances_exact — the same quantity from an exact search (what the reference's comment says it computes): a
//     tiled all-pairs scan, candidates stream through the scalar cache into SGPRs, every thread keeps the 4 smallest squared
//     distances of its query in registers. O(N^2) but VALU-dense; independent of any tree.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cfloat>
#include <utility>
#include <vector>
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

// ---- the reference's kd-tree query, reproduced ---------------------------------------------------------------------------------------------------
// Node of the flat tree. Leaf: child1 < 0, points order[lo .. hi). Inner: children, split dimension, the two sides' tight bounds along it.
struct KdNode {
    int32_t child1, child2;
    uint32_t lo, hi; // leaf: range in `order`; inner: lo = split dimension
    float divlow, divhigh;
};
struct KdBox {
    float low[3], high[3];
};
constexpr uint32_t KD_LEAF = 10;   // KDTreeSingleIndexAdaptorParams(10), splat_data.cpp:81
constexpr float KD_EPS_ERROR = 11.f; // 1 + SearchParameters(10).eps, splat_data.cpp:97 / nanoflann findNeighbors
constexpr int KD_STACK = 128;

// Host: nanoflann's KDTreeBaseClass::divideTree with middleSplit_ / planeSplit / computeMinMax (single-threaded build: n_thread_build = 1).
struct KdBuilder {
    const float* pts;
    std::vector<uint32_t> order;
    std::vector<KdNode> nodes;
    int depth = 0;

    float at(size_t k, int d) const { return pts[3 * (size_t)order[k] + d]; }

    // partition order[ind .. ind+count) around cutval along d: [ < cutval | == cutval | > cutval ), returns the two boundaries. The two sweeps and their swap
    // order are nanoflann's: the permutation they leave decides the order in which a leaf's points are offered to the result set.
    void plane_split(size_t ind, size_t count, int d, float cutval, size_t& lim1, size_t& lim2) {
        size_t l = 0, r = count - 1;
        for (;;) {
            while (l <= r && at(ind + l, d) < cutval) ++l;
            while (r && l <= r && at(ind + r, d) >= cutval) --r;
            if (l > r || !r) break;
            std::swap(order[ind + l], order[ind + r]);
            ++l, --r;
        }
        lim1 = l;
        r = count - 1;
        for (;;) {
            while (l <= r && at(ind + l, d) <= cutval) ++l;
            while (r && l <= r && at(ind + r, d) > cutval) --r;
            if (l > r || !r) break;
            std::swap(order[ind + l], order[ind + r]);
            ++l, --r;
        }
        lim2 = l;
    }

    void middle_split(size_t ind, size_t count, size_t& index, int& cutfeat, float& cutval, const KdBox& box) {
        const float EPS = 0.00001f;
        float max_span = box.high[0] - box.low[0];
        for (int d = 1; d < 3; ++d) max_span = (box.high[d] - box.low[d] > max_span) ? box.high[d] - box.low[d] : max_span;
        float max_spread = -1.f, min_elem = 0.f, max_elem = 0.f;
        cutfeat = 0;
        for (int d = 0; d < 3; ++d) {
            if (box.high[d] - box.low[d] >= (1 - EPS) * max_span) { // only the (nearly) widest dimensions of the BOX are candidates ...
                float lo = at(ind, d), hi = lo;
                for (size_t k = 1; k < count; ++k) {
                    const float v = at(ind + k, d);
                    if (v < lo) lo = v;
                    if (v > hi) hi = v;
                }
                if (hi - lo > max_spread) cutfeat = d, max_spread = hi - lo, min_elem = lo, max_elem = hi; // ... the one with the widest DATA wins
            }
        }
        const float mid = (box.low[cutfeat] + box.high[cutfeat]) / 2;
        cutval = mid < min_elem ? min_elem : mid > max_elem ? max_elem : mid;
        size_t lim1, lim2;
        plane_split(ind, count, cutfeat, cutval, lim1, lim2);
        index = lim1 > count / 2 ? lim1 : lim2 < count / 2 ? lim2 : count / 2;
    }

    int32_t divide(size_t left, size_t right, KdBox& box, int level) {
        if (level > depth) depth = level;
        const int32_t me = (int32_t)nodes.size();
        nodes.push_back(KdNode{});
        if (right - left <= KD_LEAF) {
            for (int d = 0; d < 3; ++d) box.low[d] = box.high[d] = at(left, d);
            for (size_t k = left + 1; k < right; ++k)
                for (int d = 0; d < 3; ++d) {
                    const float v = at(k, d);
                    if (box.low[d] > v) box.low[d] = v;
                    if (box.high[d] < v) box.high[d] = v;
                }
            nodes[me] = KdNode{-1, -1, (uint32_t)left, (uint32_t)right, 0.f, 0.f};
            return me;
        }
        size_t idx;
        int cutfeat;
        float cutval;
        middle_split(left, right - left, idx, cutfeat, cutval, box);
        KdBox lb = box, rb = box;
        lb.high[cutfeat] = cutval;
        const int32_t c1 = divide(left, left + idx, lb, level + 1);
        rb.low[cutfeat] = cutval;
        const int32_t c2 = divide(left + idx, right, rb, level + 1);
        nodes[me] = KdNode{c1, c2, (uint32_t)cutfeat, 0u, lb.high[cutfeat], rb.low[cutfeat]}; // the children returned their tight boxes
        for (int d = 0; d < 3; ++d) {
            box.low[d] = lb.low[d] < rb.low[d] ? lb.low[d] : rb.low[d];
            box.high[d] = lb.high[d] > rb.high[d] ? lb.high[d] : rb.high[d];
        }
        return me;
    }

    KdBox build(const float* points, uint32_t N) {
        pts = points;
        order.resize(N);
        for (uint32_t i = 0; i < N; ++i) order[i] = i;
        nodes.reserve(N / 4 + 16);
        KdBox root;
        for (int d = 0; d < 3; ++d) root.low[d] = root.high[d] = at(0, d);
        for (size_t k = 1; k < N; ++k)
            for (int d = 0; d < 3; ++d) {
                const float v = at(k, d);
                if (v < root.low[d]) root.low[d] = v;
                if (v > root.high[d]) root.high[d] = v;
            }
        KdBox box = root; // divideTree shrinks it to the union of the leaves: the same box
        divide(0, N, box, 1);
        return box;       // what nanoflann keeps as root_bbox_ (divideTree works on it in place)
    }
};

// KNNResultSet<float>(4): ascending insertion, an equal distance goes behind the ones already there
struct Knn4 {
    float d[4];
    int count, cap;
    __device__ float worst() const { return count < cap ? FLT_MAX : d[count - 1]; }
    __device__ void add(float dist) {
        int i = count;
        for (; i > 0; --i) {
            if (d[i - 1] > dist) {
                if (i < cap) d[i] = d[i - 1];
            } else break;
        }
        if (i < cap) d[i] = dist;
        if (count < cap) ++count;
    }
};

struct KdFrame {
    int32_t node;
    float mindist, side[3];
};

// One thread per query: searchLevel's recursion with an explicit stack. A frame is the far child of a visited inner node together with the lower bound and the
// per-dimension offsets it would be entered with; it is examined when everything pushed above it (the near subtree) is done - the moment the recursion returns to
// that node - and only then compared with the 4th-best distance.
__global__ void __launch_bounds__(64) kdtree_mean_distance_kernel(uint32_t N, const float* __restrict__ pts, const KdNode* __restrict__ nodes,
                                                                   const uint32_t* __restrict__ order, KdBox root, float* __restrict__ out,
                                                                   int* __restrict__ overflow) {
    const uint32_t q = blockIdx.x * 64 + threadIdx.x;
    if (q >= N) return;
    const float v[3] = {pts[3 * (size_t)q], pts[3 * (size_t)q + 1], pts[3 * (size_t)q + 2]};
    Knn4 res;
    res.count = 0, res.cap = N < 4 ? (int)N : 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) res.d[j] = 0.f; // std::vector<float> out_dists_sqr(num_results): zero where never written
    KdFrame stack[KD_STACK];
    int sp = 0;
    KdFrame cur;
    cur.node = 0, cur.mindist = 0.f;
    for (int d = 0; d < 3; ++d) { // computeInitialDistances: zero for a query inside the root box (every query is a data point)
        cur.side[d] = 0.f;
        if (v[d] < root.low[d]) cur.side[d] = (v[d] - root.low[d]) * (v[d] - root.low[d]), cur.mindist += cur.side[d];
        if (v[d] > root.high[d]) cur.side[d] = (v[d] - root.high[d]) * (v[d] - root.high[d]), cur.mindist += cur.side[d];
    }
    bool have = true;
    while (have) {
        // descend to a leaf along the near children, leaving the far ones on the stack
        for (;;) {
            const KdNode nd = nodes[cur.node];
            if (nd.child1 < 0) {
                const float worst = res.worst(); // read once per leaf, as searchLevel does
                for (uint32_t k = nd.lo; k < nd.hi; ++k) {
                    const uint32_t p = order[k];
                    const float dx = v[0] - pts[3 * (size_t)p], dy = v[1] - pts[3 * (size_t)p + 1], dz = v[2] - pts[3 * (size_t)p + 2];
                    const float dist = (dx * dx + dy * dy) + dz * dz; // L2_Simple_Adaptor::evalMetric: accumulated in dimension order
                    if (dist < worst) res.add(dist);
                }
                break;
            }
            const int dim = (int)nd.lo;
            const float val = dim == 0 ? v[0] : dim == 1 ? v[1] : v[2];
            const float diff1 = val - nd.divlow, diff2 = val - nd.divhigh;
            const bool near_is_1 = (diff1 + diff2) < 0;
            const float cut = near_is_1 ? (val - nd.divhigh) * (val - nd.divhigh) : (val - nd.divlow) * (val - nd.divlow);
            KdFrame far = cur;
            far.node = near_is_1 ? nd.child2 : nd.child1;
            const float had = dim == 0 ? cur.side[0] : dim == 1 ? cur.side[1] : cur.side[2];
            far.mindist = cur.mindist + cut - had;
            if (dim == 0) far.side[0] = cut; else if (dim == 1) far.side[1] = cut; else far.side[2] = cut;
            if (sp == KD_STACK) {
                atomicExch(overflow, 1);
                return;
            }
            stack[sp++] = far;
            cur.node = near_is_1 ? nd.child1 : nd.child2;
        }
        have = false;
        while (sp > 0) {
            cur = stack[--sp];
            if (cur.mindist * KD_EPS_ERROR <= res.worst()) {
                have = true;
                break;
            }
        }
    }
    // splat_data.cpp:99-109: up to 3 of the results with d^2 > 1e-8 (the query itself and exact duplicates are among them)
    float sum = 0.f;
    int valid = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < res.cap && valid < 3 && res.d[j] > 1e-8f) { sum += sqrtf(res.d[j]); ++valid; }
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
    if (N == 1) { // "num_points <= 1" -> 0.01 (splat_data.cpp:72-74)
        const float v = 0.01f;
        return (int)hipMemcpyAsync(out, &v, sizeof(float), hipMemcpyHostToDevice, s);
    }
    // the tree is built where the reference builds it: on the host, once per run (a sequential algorithm whose index permutation is part of the result)
    std::vector<float> host(3 * (size_t)N);
    hipError_t e = hipMemcpyAsync(host.data(), points, sizeof(float) * host.size(), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return (int)e;
    KdBuilder b;
    const KdBox root = b.build(host.data(), N);
    if (b.depth > KD_STACK) return LFS_E_UNSUPPORTED; // one stack frame per level
    KdNode* d_nodes = nullptr;
    uint32_t* d_order = nullptr;
    int* d_flag = nullptr;
    const size_t nb = sizeof(KdNode) * b.nodes.size(), ob = sizeof(uint32_t) * (size_t)N;
    char* mem = nullptr;
    e = hipMalloc((void**)&mem, nb + ob + 16);
    if (e != hipSuccess) return (int)e;
    d_nodes = (KdNode*)mem, d_order = (uint32_t*)(mem + nb), d_flag = (int*)(mem + nb + ob);
    int flag = 0;
    e = hipMemcpyAsync(d_nodes, b.nodes.data(), nb, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_order, b.order.data(), ob, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_flag, 0, sizeof(int), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(kdtree_mean_distance_kernel, dim3((N + 63) / 64), dim3(64), 0, s, N, points, d_nodes, d_order, root, out, d_flag);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, s);
    const hipError_t e2 = hipStreamSynchronize(s); // the host vectors and the allocation live until here
    (void)hipFree(mem);
    if (e != hipSuccess) return (int)e;
    if (e2 != hipSuccess) return (int)e2;
    return flag ? LFS_E_UNSUPPORTED : LFS_OK;
}

extern "C" int lfs_mean_neighbor_distances_exact(uint32_t N, const float* points, float* out, lfs_stream_t stream) {
    if (!N) return LFS_OK;
    if (!points || !out) return LFS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    lfs::ProfScope prof("mean_neighbor_distances_exact", s);
    // N == 1: the only result is the query itself (d = 0, skipped) -> 0.01, the reference's "num_points <= 1" case (splat_data.cpp:72-74)
    hipLaunchKernelGGL(knn_mean_distance_kernel, dim3((N + KNN_THREADS - 1) / KNN_THREADS), dim3(KNN_THREADS), 0, s, N, points, out);
    return (int)hipGetLastError();
}
