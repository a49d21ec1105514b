// K3-K6 — tile intersection, depth ordering and per-tile offsets (replace
// gsplat::intersect_tile / intersect_offset; reference: gsplat/IntersectTile.cu
// :24-113 count/emit, :206-252 offsets, :290-342 CUB radix sort of the 64-bit
// keys; host gsplat/Intersect.cpp:15-137).
//
// Integer stage — outputs are BIT-EXACT with the reference given the same
// (means2d, radii, depths). The reference emits (key,value) pairs Gaussian-major
// and runs a 6-pass global LSD radix sort over 46 key bits. Here the tile id is
// never sorted at all:
//   count   : per-tile intersection counts (LDS histogram per workgroup, one
//             global atomic per touched (workgroup, tile))          -> totals[T]
//   scan    : exclusive scan of totals                              -> tile_offsets[T+1]
//   scatter : every intersection goes straight into its tile's bucket
//             (LDS ranks + one reserving atomic per (workgroup, tile))
//   sort    : one workgroup per tile sorts its bucket in LDS on the composite
//             (depth bits << 32 | flatten id).
// A flatten id occurs at most once per tile and the reference's emission order
// inside a tile is ascending flatten id, so "stable sort by (tile, depth)" and
// "sort by (tile, depth, flatten id)" are the same total order: the result does
// not depend on the (atomic, unordered) bucket fill. tile_offsets falls out of
// the scan, so intersect_offset is free on this path.
#include "lfs_math.cuh"
#include "lfs_prof.h"
#include "lfs_tilelists.cuh"
#include "lfs_step_internal.h"
#include "../../include/lfs_gsplat.h"

namespace lfs {

struct TileRect { uint32_t x0, y0, x1, y1; };

// float -> uint32 with CUDA's saturating semantics (negative / NaN -> 0)
LFS_DI uint32_t sat_u32(float v) {
    if (!(v > 0.f)) return 0u;
    if (v >= 4294967296.f) return 0xFFFFFFFFu;
    return uint32_t(v);
}

LFS_DI bool tile_rect(const float* __restrict__ means2d, const int32_t* __restrict__ radii, size_t idx,
                      float tile_size_f, uint32_t tw, uint32_t th, TileRect& r) {
    const int2 rr = reinterpret_cast<const int2*>(radii)[idx];
    const float rx = float(rr.x), ry = float(rr.y);
    if (rx <= 0.f || ry <= 0.f) return false;
    const float2 m = reinterpret_cast<const float2*>(means2d)[idx];
    const float trx = rx / tile_size_f, try_ = ry / tile_size_f;
    const float tx = m.x / tile_size_f, ty = m.y / tile_size_f;
    r.x0 = min(sat_u32(floorf(tx - trx)), tw);
    r.y0 = min(sat_u32(floorf(ty - try_)), th);
    r.x1 = min(sat_u32(ceilf(tx + trx)), tw);
    r.y1 = min(sat_u32(ceilf(ty + try_)), th);
    return true;
}

// number of bits of v: floor(log2 v) + 1 (the reference evaluates this with double log2 on the host)
static inline uint32_t bit_width_u32(uint32_t v) { uint32_t b = 0; while (v) { ++b; v >>= 1; } return b; }

// ---------------------------------------------------------------------------
// count: tiles_per_gauss + per-tile totals
// ---------------------------------------------------------------------------
template <bool LDS_HIST>
__global__ void __launch_bounds__(1024) isect_count_kernel(
    const uint32_t C, const uint32_t N, const uint32_t per_block,
    const float* __restrict__ means2d, const int32_t* __restrict__ radii,
    const float tile_size_f, const uint32_t tw, const uint32_t th,
    int32_t* __restrict__ tiles_per_gauss, uint32_t* __restrict__ totals) {
    LFS_DYN_LDS(uint32_t, hist);
    const uint32_t T = C * tw * th, n_tiles = tw * th;
    if (LDS_HIST) {
        for (uint32_t t = threadIdx.x; t < T; t += blockDim.x) hist[t] = 0u;
        __syncthreads();
    }
    const size_t total = size_t(C) * N;
    const size_t begin = size_t(blockIdx.x) * per_block;
    const size_t end = min(begin + per_block, total);
    for (size_t idx = begin + threadIdx.x; idx < end; idx += blockDim.x) {
        TileRect r;
        int32_t n = 0;
        if (tile_rect(means2d, radii, idx, tile_size_f, tw, th, r)) {
            n = int32_t((r.y1 - r.y0) * (r.x1 - r.x0));
            const uint32_t base = uint32_t(idx / N) * n_tiles;
            for (uint32_t i = r.y0; i < r.y1; ++i)
                for (uint32_t j = r.x0; j < r.x1; ++j) {
                    if (LDS_HIST) atomicAdd(&hist[base + i * tw + j], 1u);
                    else atomicAdd(&totals[base + i * tw + j], 1u);
                }
        }
        tiles_per_gauss[idx] = n;
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
// scatter: write (key, flatten id) of every intersection into its tile bucket
// ---------------------------------------------------------------------------
template <bool LDS_HIST>
__global__ void __launch_bounds__(1024) isect_scatter_kernel(
    const uint32_t C, const uint32_t N, const uint32_t per_block,
    const float* __restrict__ means2d, const int32_t* __restrict__ radii, const float* __restrict__ depths,
    const float tile_size_f, const uint32_t tw, const uint32_t th, const uint32_t tile_n_bits,
    const int32_t* __restrict__ offsets, uint32_t* __restrict__ cursor,
    int64_t* __restrict__ isect_ids, int32_t* __restrict__ flatten_ids) {
    LFS_DYN_LDS(uint32_t, lds);
    const uint32_t T = C * tw * th, n_tiles = tw * th;
    uint32_t* cnt = lds;        // [T] local count, then local rank counter
    uint32_t* base_s = lds + T; // [T] start of this workgroup's slice inside the tile bucket
    const size_t total = size_t(C) * N;
    const size_t begin = size_t(blockIdx.x) * per_block;
    const size_t end = min(begin + per_block, total);
    if (LDS_HIST) {
        for (uint32_t t = threadIdx.x; t < T; t += blockDim.x) cnt[t] = 0u;
        __syncthreads();
        for (size_t idx = begin + threadIdx.x; idx < end; idx += blockDim.x) {
            TileRect r;
            if (!tile_rect(means2d, radii, idx, tile_size_f, tw, th, r)) continue;
            const uint32_t cb = uint32_t(idx / N) * n_tiles;
            for (uint32_t i = r.y0; i < r.y1; ++i)
                for (uint32_t j = r.x0; j < r.x1; ++j) atomicAdd(&cnt[cb + i * tw + j], 1u);
        }
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < T; t += blockDim.x) {
            const uint32_t c = cnt[t];
            base_s[t] = c ? atomicAdd(&cursor[t], c) : 0u;
            cnt[t] = 0u;
        }
        __syncthreads();
    }
    for (size_t idx = begin + threadIdx.x; idx < end; idx += blockDim.x) {
        TileRect r;
        if (!tile_rect(means2d, radii, idx, tile_size_f, tw, th, r)) continue;
        const uint32_t cid = uint32_t(idx / N);
        const uint32_t cb = cid * n_tiles;
        const uint64_t dbits = uint64_t(__float_as_uint(depths[idx]));
        for (uint32_t i = r.y0; i < r.y1; ++i)
            for (uint32_t j = r.x0; j < r.x1; ++j) {
                const uint32_t tile = i * tw + j, t = cb + tile;
                uint32_t slot;
                if (LDS_HIST) slot = base_s[t] + atomicAdd(&cnt[t], 1u);
                else slot = atomicAdd(&cursor[t], 1u);
                const size_t pos = size_t(offsets[t]) + slot;
                // ONE 8-byte store per intersection: (depth bits << 32 | flatten id) parked in the isect_ids output; the per-tile
                // sort kernel turns it into the reference's (camera | tile | depth) key and the flatten id. (Two stores of 8 + 4
                // bytes doubled the number of scattered partial-line writes, the cost of this kernel: WRITE_SIZE 297 MB for 53 MB.)
                isect_ids[pos] = int64_t((dbits << 32) | uint64_t(uint32_t(idx)));
            }
    }
}

// ---------------------------------------------------------------------------
// Two-pass scatter (the default when the caller provides a scratch array): the one-pass kernel above writes every intersection as an isolated
// 8-byte store (a workgroup's ~9 000 intersections fall into ~8 000 different tile buckets), and HBM writes in 32-byte sectors: 158 MB written for
// 36 MB of payload (PMC, SYN-B). Binning in two steps keeps every store part of a run of hundreds of bytes:
//   pass 1 (isect_rows_kernel) : Gaussians -> tile ROW buckets (C * tile_h <= 512 of them) in `scratch`; a workgroup stages its intersections in LDS
//                                sorted by row and copies each row's run out contiguously. The entry carries the tile column in the bits above
//                                the flatten id: (depth bits << 32 | tile_x << idx_bits | flatten id).
//   pass 2 (isect_tiles_kernel): a workgroup takes 8192 consecutive entries of `scratch` (one or two rows), bins them by tile in LDS and copies each
//                                tile's run into the tile's bucket of isect_ids, as (depth bits << 32 | flatten id) - what the one-pass kernel
//                                leaves there. The row range of the scratch array IS the row's range of the final array (tile buckets are laid
//                                out in tile order), so a position alone identifies the row.
// The per-tile sort that follows orders each bucket by (depth, flatten id): the result does not depend on how the bucket was filled.
// ---------------------------------------------------------------------------
#ifndef LFS_ROWS_STAGE
#define LFS_ROWS_STAGE 8192
#endif
#ifndef LFS_TILES_CHUNK
#define LFS_TILES_CHUNK 8192
#endif
#ifndef LFS_TILES_SPAN
#define LFS_TILES_SPAN 1024
#endif
#ifndef LFS_SORT_2CLASS
#define LFS_SORT_2CLASS 1   // the <= 1024 and <= 4096 classes of the per-tile sort as ONE launch (lfs_tilelists.cuh tile_sort_bins_2class_kernel); 0 = two launches (rounds 2 - 4)
#endif
constexpr uint32_t ROWS_MAX = 512;                // pass 1: row buckets per launch (LDS tables)
constexpr uint32_t ROWS_STAGE = LFS_ROWS_STAGE;   // pass 1: staged entries per workgroup (64 KiB); more than that -> direct stores
constexpr uint32_t ROWS_PER_BLOCK = 1024;         // pass 1: Gaussians per workgroup (one per thread)
constexpr uint32_t TILES_CHUNK = LFS_TILES_CHUNK; // pass 2: entries per workgroup (64 KiB staged)
constexpr uint32_t TILES_SPAN = LFS_TILES_SPAN;   // pass 2: tiles a chunk may span with LDS binning (two workgroups per CU at 1024); more (many empty rows) -> direct stores

// exclusive scan of a[0..L) (LDS) into out[0..L] (out[L] = total); all 1024 threads call it; tmp: 17 words of LDS
LFS_DI void block_scan_1024(const uint32_t* a, uint32_t* out, uint32_t L, uint32_t* tmp) {
    const uint32_t ipt = (L + 1023u) / 1024u, b = threadIdx.x * ipt;
    uint32_t local = 0;
    for (uint32_t k = 0; k < ipt; ++k) if (b + k < L) local += a[b + k];
    uint32_t s = local;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(s, d, 64); if (int(lane) >= d) s += o; }
    if (lane == 63) tmp[wave] = s;
    __syncthreads();
    uint32_t off = s - local;
    for (uint32_t w = 0; w < wave; ++w) off += tmp[w];
    __syncthreads(); // a may alias out: every thread has read its items before anyone writes
    for (uint32_t k = 0; k < ipt; ++k) if (b + k < L) { const uint32_t v = a[b + k]; out[b + k] = off; off += v; }
    if (threadIdx.x == 1023) out[L] = off;
    __syncthreads();
}

__global__ void __launch_bounds__(1024) isect_rows_kernel(
    const uint32_t C, const uint32_t N, const float* __restrict__ means2d, const int32_t* __restrict__ radii, const float* __restrict__ depths,
    const float tile_size_f, const uint32_t tw, const uint32_t th, const uint32_t idx_bits,
    const int32_t* __restrict__ offsets, uint32_t* __restrict__ row_cursor, uint64_t* __restrict__ scratch, const int32_t* __restrict__ abort_flag = nullptr) {
    LFS_DYN_LDS(uint64_t, stage); // [ROWS_STAGE]
    if (abort_flag != nullptr && *abort_flag != 0) return; // speculative step: the lists do not fit the caller's buffers (tile_scan_kernel) - this kernel derives its
                                                           // write positions from the Gaussians, not from the (zeroed) offsets, so it has to stop itself
    __shared__ uint32_t cnt[ROWS_MAX], lbase[ROWS_MAX + 1], gbase[ROWS_MAX], tmp[17];
    const uint32_t R = C * th;
    const size_t total = size_t(C) * N;
    const size_t begin = size_t(blockIdx.x) * ROWS_PER_BLOCK + threadIdx.x; // one Gaussian per thread
    for (uint32_t r = threadIdx.x; r < R; r += 1024) cnt[r] = 0u;
    __syncthreads();
    TileRect rc{0, 0, 0, 0};
    const bool have = begin < total && tile_rect(means2d, radii, begin, tile_size_f, tw, th, rc) && rc.x1 > rc.x0 && rc.y1 > rc.y0;
    const uint32_t nx = rc.x1 - rc.x0, rb = uint32_t(begin / N) * th;
    if (have) for (uint32_t i = rc.y0; i < rc.y1; ++i) atomicAdd(&cnt[rb + i], nx);
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < R; r += 1024) { const uint32_t c = cnt[r]; gbase[r] = c ? atomicAdd(&row_cursor[r], c) : 0u; }
    block_scan_1024(cnt, lbase, R, tmp);
    const uint32_t E = lbase[R];
    const bool staged = E <= ROWS_STAGE;
    for (uint32_t r = threadIdx.x; r < R; r += 1024) cnt[r] = 0u; // now the running rank inside the row
    __syncthreads();
    if (have) {
        const uint64_t hi = uint64_t(__float_as_uint(depths[begin])) << 32 | uint64_t(uint32_t(begin));
        for (uint32_t i = rc.y0; i < rc.y1; ++i) {
            const uint32_t row = rb + i;
            const uint32_t k = atomicAdd(&cnt[row], nx);
            if (staged) {
                uint64_t* dst = stage + lbase[row] + k;
                for (uint32_t j = 0; j < nx; ++j) dst[j] = hi | (uint64_t(rc.x0 + j) << idx_bits);
            } else {
                uint64_t* dst = scratch + size_t(offsets[size_t(row) * tw]) + gbase[row] + k;
                for (uint32_t j = 0; j < nx; ++j) dst[j] = hi | (uint64_t(rc.x0 + j) << idx_bits);
            }
        }
    }
    if (!staged) return;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t r = wave; r < R; r += 16) {
        const uint32_t lb = lbase[r], n = lbase[r + 1] - lb;
        if (n == 0) continue;
        uint64_t* dst = scratch + size_t(offsets[size_t(r) * tw]) + gbase[r];
        for (uint32_t k = lane; k < n; k += 64) dst[k] = stage[lb + k];
    }
}

__global__ void __launch_bounds__(1024) isect_tiles_kernel(
    const uint32_t R, const uint32_t tw, const uint32_t idx_bits, const int64_t n_isects_arg,
    const int32_t* __restrict__ offsets, uint32_t* __restrict__ cursor, const uint64_t* __restrict__ scratch, int64_t* __restrict__ isect_ids) {
    LFS_DYN_LDS(uint64_t, stage); // [TILES_CHUNK]
    __shared__ uint32_t cnt[TILES_SPAN], lpre[TILES_SPAN + 1], gb[TILES_SPAN], tmp[17];
    __shared__ uint32_t rows_s[2];
    constexpr uint32_t PER = TILES_CHUNK / 1024;
    const int64_t n_isects = n_isects_arg >= 0 ? n_isects_arg : int64_t(offsets[size_t(R) * tw]); // < 0: the count lives on the device (offsets[T]); the grid covers the caller's capacity
    const int64_t p0 = int64_t(blockIdx.x) * TILES_CHUNK, p1 = min(p0 + int64_t(TILES_CHUNK), n_isects);
    if (p0 >= n_isects) return; // (uniform)
    if (threadIdx.x < 2) { // the row that holds p0 / p1 - 1: the largest r with row_start(r) <= p
        const int64_t p = threadIdx.x == 0 ? p0 : p1 - 1;
        uint32_t lo = 0, hi = R; // invariant: row_start(lo) <= p < row_start(hi) (row_start(R) = n_isects)
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (int64_t(offsets[size_t(mid) * tw]) <= p) lo = mid; else hi = mid; }
        rows_s[threadIdx.x] = lo;
    }
    __syncthreads();
    const uint32_t r_lo = rows_s[0], r_hi = rows_s[1];
    const uint32_t t_lo = r_lo * tw, span = (r_hi - r_lo + 1) * tw;
    const uint64_t idx_mask = (uint64_t(1) << idx_bits) - 1;
    if (span > TILES_SPAN) { // a chunk across many (mostly empty) rows: plain scatter
        for (int64_t p = p0 + threadIdx.x; p < p1; p += 1024) {
            const uint64_t e = scratch[p];
            uint32_t r = r_lo;
            while (r < r_hi && int64_t(offsets[size_t(r + 1) * tw]) <= p) ++r;
            const uint32_t t = r * tw + uint32_t((e & 0xFFFFFFFFull) >> idx_bits);
            const uint32_t slot = atomicAdd(&cursor[t], 1u);
            isect_ids[size_t(offsets[t]) + slot] = int64_t((e & 0xFFFFFFFF00000000ull) | (e & idx_mask));
        }
        return;
    }
    for (uint32_t t = threadIdx.x; t < span; t += 1024) cnt[t] = 0u;
    __syncthreads();
    uint64_t ent[PER]; uint32_t where[PER]; // where = local tile << 16 | rank   (rank < 8192, local tile < 2048)
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        const int64_t p = p0 + threadIdx.x + int64_t(k) * 1024;
        where[k] = 0xFFFFFFFFu;
        if (p < p1) {
            const uint64_t e = scratch[p];
            uint32_t r = r_lo;
            while (r < r_hi && int64_t(offsets[size_t(r + 1) * tw]) <= p) ++r;
            const uint32_t tl = (r - r_lo) * tw + uint32_t((e & 0xFFFFFFFFull) >> idx_bits);
            ent[k] = (e & 0xFFFFFFFF00000000ull) | (e & idx_mask);
            where[k] = tl << 16 | atomicAdd(&cnt[tl], 1u);
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < span; t += 1024) { const uint32_t c = cnt[t]; gb[t] = c ? atomicAdd(&cursor[t_lo + t], c) : 0u; }
    block_scan_1024(cnt, lpre, span, tmp);
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k)
        if (where[k] != 0xFFFFFFFFu) stage[lpre[where[k] >> 16] + (where[k] & 0xFFFFu)] = ent[k];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t t = wave; t < span; t += 16) {
        const uint32_t lb = lpre[t], n = lpre[t + 1] - lb;
        if (n == 0) continue;
        int64_t* dst = isect_ids + size_t(offsets[t_lo + t]) + gb[t];
        for (uint32_t k = lane; k < n; k += 64) dst[k] = int64_t(stage[lb + k]);
    }
}

// ---------------------------------------------------------------------------
// sort == false path: reference emission order (Gaussian-major), needs the
// inclusive scan of tiles_per_gauss.
// ---------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 2048; // per workgroup of 256 threads (8 per thread)

__global__ void __launch_bounds__(256) scan_block_sums_kernel(const size_t n, const int32_t* __restrict__ in, int64_t* __restrict__ block_sums) {
    __shared__ int64_t ws[4];
    const size_t base = size_t(blockIdx.x) * SCAN_ITEMS;
    int64_t s = 0;
    for (int i = threadIdx.x; i < SCAN_ITEMS; i += 256) { const size_t p = base + i; if (p < n) s += in[p]; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ void __launch_bounds__(1024) scan_sums_kernel(const uint32_t nb, int64_t* __restrict__ block_sums) {
    // exclusive scan of block_sums in place, single workgroup
    __shared__ int64_t wave_sums[16];
    __shared__ int64_t carry_s;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += 1024) {
        const uint32_t t = base + threadIdx.x;
        const int64_t v = t < nb ? block_sums[t] : 0;
        int64_t s = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int64_t o = __shfl_up(s, d, 64); if (int(lane) >= d) s += o; }
        if (lane == 63) wave_sums[wave] = s;
        __syncthreads();
        int64_t wave_off = 0;
        for (uint32_t w = 0; w < wave; ++w) wave_off += wave_sums[w];
        const int64_t carry = carry_s;
        if (t < nb) block_sums[t] = carry + wave_off + s - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wave_off + s;
        __syncthreads();
    }
}
// each thread owns 8 consecutive Gaussians: exclusive start offsets -> emit
__global__ void __launch_bounds__(256) isect_emit_unsorted_kernel(
    const uint32_t C, const uint32_t N, const float* __restrict__ means2d, const int32_t* __restrict__ radii,
    const float* __restrict__ depths, const int32_t* __restrict__ tiles_per_gauss, const int64_t* __restrict__ block_offs,
    const float tile_size_f, const uint32_t tw, const uint32_t th, const uint32_t tile_n_bits,
    int64_t* __restrict__ isect_ids, int32_t* __restrict__ flatten_ids) {
    __shared__ int64_t ws[4];
    const size_t total = size_t(C) * N;
    const size_t base = size_t(blockIdx.x) * SCAN_ITEMS + size_t(threadIdx.x) * 8;
    int32_t cnt[8]; int64_t local = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { cnt[i] = (base + i < total) ? tiles_per_gauss[base + i] : 0; local += cnt[i]; }
    int64_t s = local;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int64_t o = __shfl_up(s, d, 64); if (int(lane) >= d) s += o; }
    if (lane == 63) ws[wave] = s;
    __syncthreads();
    int64_t off = block_offs[blockIdx.x] + s - local;
    for (uint32_t w = 0; w < wave; ++w) off += ws[w];
    const uint32_t n_tiles = tw * th;
    (void)n_tiles;
#pragma unroll 1
    for (int i = 0; i < 8; ++i) {
        const size_t idx = base + i;
        if (idx >= total || cnt[i] == 0) continue;
        TileRect r;
        if (!tile_rect(means2d, radii, idx, tile_size_f, tw, th, r)) continue;
        const uint64_t cid_enc = uint64_t(idx / N) << (32 + tile_n_bits);
        const uint64_t dbits = uint64_t(__float_as_uint(depths[idx]));
        int64_t cur = off;
        for (uint32_t y = r.y0; y < r.y1; ++y)
            for (uint32_t x = r.x0; x < r.x1; ++x) {
                isect_ids[cur] = int64_t(cid_enc | (uint64_t(y * tw + x) << 32) | dbits);
                flatten_ids[cur] = int32_t(idx);
                ++cur;
            }
        off += cnt[i];
    }
}

// ---------------------------------------------------------------------------
// K6 standalone: lower-bound offsets from sorted keys
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) isect_offset_kernel(
    const int64_t n_isects, const int64_t* __restrict__ isect_ids, const uint32_t n_tiles, const uint32_t tile_n_bits,
    const int64_t T, int32_t* __restrict__ offsets) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_isects) return;
    auto flat = [&](int64_t key) { const int64_t hi = key >> 32; return (hi >> tile_n_bits) * n_tiles + (hi & ((int64_t(1) << tile_n_bits) - 1)); };
    const int64_t cur = flat(isect_ids[i]);
    const int64_t prev = i > 0 ? flat(isect_ids[i - 1]) : -1;
    for (int64_t t = prev + 1; t <= cur; ++t) offsets[t] = int32_t(i);
    if (i == n_isects - 1) for (int64_t t = cur + 1; t < T; ++t) offsets[t] = int32_t(n_isects);
}

// ---- workspace layout ------------------------------------------------------
struct IsectWs {
    uint32_t* totals;   // [T]
    uint32_t* cursor;   // [T]
    uint32_t* row_cursor; // [C * tile_h]   (two-pass scatter)
    int32_t* offsets;   // [T+1]
    int64_t* block_sums; // [ceil(CN / SCAN_ITEMS) + 1]
    size_t bytes;
};
static inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }
static IsectWs isect_ws(void* base, uint32_t C, uint32_t N, uint32_t tw, uint32_t th) {
    const size_t T = size_t(C) * tw * th;
    const size_t nb = (size_t(C) * N + SCAN_ITEMS - 1) / SCAN_ITEMS + 1;
    IsectWs w; char* p = (char*)base; size_t o = 0;
    w.totals = (uint32_t*)(p + o); o += align256(T * 4);
    w.cursor = (uint32_t*)(p + o); o += align256(T * 4);
    w.row_cursor = (uint32_t*)(p + o); o += align256(size_t(C) * th * 4);
    w.offsets = (int32_t*)(p + o); o += align256((T + 1) * 4);
    w.block_sums = (int64_t*)(p + o); o += align256(nb * 8);
    w.bytes = o;
    return w;
}
static inline uint32_t isect_per_block(size_t total) {
    // ~512 workgroups of 1024 threads on a big problem, never fewer than 1024 Gaussians each
    size_t pb = (total + 511) / 512;
    if (pb < 1024) pb = 1024;
    return uint32_t((pb + 1023) / 1024 * 1024);
}
constexpr size_t LDS_HIST_LIMIT = 64 * 1024; // bytes for the scatter kernel's two [T] arrays

} // namespace lfs

using namespace lfs;

extern "C" size_t lfs_intersect_tile_workspace_bytes(uint32_t C, uint32_t N, uint32_t tile_width, uint32_t tile_height) {
    return isect_ws(nullptr, C, N, tile_width, tile_height).bytes;
}

int lfs::isect_count_impl(
    uint32_t C, uint32_t N, const float* means2d, const int32_t* radii,
    uint32_t tile_size, uint32_t tile_width, uint32_t tile_height,
    int32_t* tiles_per_gauss, int64_t* n_isects, int64_t* max_tile_isects, int32_t* tile_offsets, uint32_t flags, int64_t* stamp_out, int64_t stamp,
    void* workspace, size_t workspace_bytes, hipStream_t s, const IsectGuard* guard) {
    if (!n_isects || !workspace || tile_size == 0 || tile_width == 0 || tile_height == 0 || C == 0) return LFS_E_INVALID;
    if (bit_width_u32(tile_width * tile_height) + bit_width_u32(C) > 32) return LFS_E_UNSUPPORTED; // IntersectTile.cu:154
    IsectWs w = isect_ws(workspace, C, N, tile_width, tile_height);
    if (workspace_bytes < w.bytes) return LFS_E_WORKSPACE;
    const uint32_t T = C * tile_width * tile_height;
    // the scan kernel leaves totals[] zero again and zeroes both cursors itself: a caller that OWNS the workspace and passes the one of its previous call
    // (same C, N, tile grid) may set LFS_ISECT_COUNTERS_ZERO and save the memset (nothing in this repository does any more: 32 KB, ~2 us)
    if (!(flags & LFS_ISECT_COUNTERS_ZERO)) {
        hipError_t e = hipMemsetAsync(w.totals, 0, size_t(T) * 4, s);
        if (e != hipSuccess) return (int)e;
    }
    const size_t total = size_t(C) * N;
    // Round 6, measured and removed (profiles/r06/lease16_count_scan_tail_ab.txt): the scan as the tail of the count kernel's LAST workgroup (ticket counter behind a
    // device-scope fence, totals read with device-scope atomic loads) - count + scan 0.031 -> 0.140 ms. A device-scope release on this part writes the L2 of the XCD
    // back (the eight L2s are not coherent with each other), and every one of the 512 workgroups paid for it behind the projection kernel's 200 MB of dirty lines:
    // "last workgroup finishes the job" is not a pattern for an eight-XCD device; the second launch (8 us + gap) stays.
    lfs::ProfScope prof("isect_count_scan", s);
    if (total > 0) {
        if (!means2d || !radii || !tiles_per_gauss) return LFS_E_INVALID;
        const uint32_t pb = isect_per_block(total);
        const uint32_t blocks = uint32_t((total + pb - 1) / pb);
        if (size_t(T) * 8 <= LDS_HIST_LIMIT)
            hipLaunchKernelGGL(isect_count_kernel<true>, dim3(blocks), dim3(1024), T * 4, s, C, N, pb, means2d, radii,
                               float(tile_size), tile_width, tile_height, tiles_per_gauss, w.totals);
        else
            hipLaunchKernelGGL(isect_count_kernel<false>, dim3(blocks), dim3(1024), 0, s, C, N, pb, means2d, radii,
                               float(tile_size), tile_width, tile_height, tiles_per_gauss, w.totals);
    }
    if (guard != nullptr) {
        if (guard->capacity < 0 || !guard->abort_flag) return LFS_E_INVALID;
        hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, s, T, w.totals, w.offsets, n_isects, true, w.cursor, w.row_cursor, C * tile_height, tile_offsets,
                           max_tile_isects, stamp_out, stamp, guard->capacity, sort_class_limit(guard->assumed_longest), guard->abort_flag);
    } else
        hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, s, T, w.totals, w.offsets, n_isects, true, w.cursor, w.row_cursor, C * tile_height, tile_offsets,
                           max_tile_isects, stamp_out, stamp);
    return (int)hipGetLastError();
}

// what the guarded (speculative) emit needs: the two-pass binned scatter (the one-pass kernels do not watch the abort flag)
bool lfs::isect_two_pass_supported(uint32_t C, uint32_t N, uint32_t tile_width, uint32_t tile_height) {
    const size_t total = size_t(C) * N;
    if (total == 0 || tile_width == 0 || tile_height == 0) return false;
    const uint32_t idx_bits = bit_width_u32(uint32_t(total - 1)) ? bit_width_u32(uint32_t(total - 1)) : 1u;
    return C * tile_height <= ROWS_MAX && total <= 0xFFFFFFFFull && idx_bits + bit_width_u32(tile_width - 1) <= 32 && !(lfs_get_debug_flags() & 32u);
}

uint32_t* lfs::isect_workspace_totals(void* workspace, uint32_t C, uint32_t N, uint32_t tile_width, uint32_t tile_height) {
    return isect_ws(workspace, C, N, tile_width, tile_height).totals;
}
const int32_t* lfs::isect_workspace_offsets(void* workspace, uint32_t C, uint32_t N, uint32_t tile_width, uint32_t tile_height) {
    return isect_ws(workspace, C, N, tile_width, tile_height).offsets;
}

extern "C" int lfs_intersect_tile_count_ex(
    uint32_t C, uint32_t N, const float* means2d, const int32_t* radii,
    uint32_t tile_size, uint32_t tile_width, uint32_t tile_height,
    int32_t* tiles_per_gauss, int64_t* n_isects, int64_t* max_tile_isects, int32_t* tile_offsets, uint32_t flags, int64_t* stamp_out, int64_t stamp,
    void* workspace, size_t workspace_bytes, lfs_stream_t stream) {
    return lfs::isect_count_impl(C, N, means2d, radii, tile_size, tile_width, tile_height, tiles_per_gauss, n_isects, max_tile_isects, tile_offsets, flags, stamp_out,
                                 stamp, workspace, workspace_bytes, (hipStream_t)stream, nullptr);
}

extern "C" int lfs_intersect_tile_count(
    uint32_t C, uint32_t N, const float* means2d, const int32_t* radii,
    uint32_t tile_size, uint32_t tile_width, uint32_t tile_height,
    int32_t* tiles_per_gauss, int64_t* n_isects, void* workspace, size_t workspace_bytes, lfs_stream_t stream) {
    return lfs_intersect_tile_count_ex(C, N, means2d, radii, tile_size, tile_width, tile_height, tiles_per_gauss, n_isects, nullptr, nullptr, 0u, nullptr, 0, workspace,
                                       workspace_bytes, stream);
}

int lfs::isect_emit_impl(
    uint32_t C, uint32_t N, const float* means2d, const int32_t* radii, const float* depths,
    uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, int sort, int64_t n_isects,
    const int32_t* tiles_per_gauss, int64_t* isect_ids, int32_t* flatten_ids, int32_t* tile_offsets, int64_t* scratch, int64_t max_tile_isects,
    void* workspace, size_t workspace_bytes, hipStream_t s, const IsectGuard* guard) {
    if (!workspace || C == 0 || tile_size == 0) return LFS_E_INVALID;
    IsectWs w = isect_ws(workspace, C, N, tile_width, tile_height);
    if (workspace_bytes < w.bytes) return LFS_E_WORKSPACE;
    const uint32_t T = C * tile_width * tile_height;
    // guarded (speculative) call: the count is on the device. Grids cover the caller's capacity, the kernels read offsets[T]; the sort classes are the ones
    // tile_scan_kernel was told about (anything longer raised the abort flag and emptied the lists)
    const bool guarded = guard != nullptr;
    const int64_t n_grid = guarded ? guard->capacity : n_isects;     // what the launch geometry covers
    const int64_t n_arg = guarded ? int64_t(-1) : n_isects;          // what the kernels receive
    if (guarded) {
        if (!sort || !scratch || guard->capacity < 0) return LFS_E_INVALID;
        n_isects = guard->capacity;
        max_tile_isects = int64_t(sort_class_limit(guard->assumed_longest));
        if (max_tile_isects == int64_t(0xFFFFFFFFu)) max_tile_isects = -1;
    }
    if (tile_offsets) {
        hipError_t e = hipMemcpyAsync(tile_offsets, w.offsets, size_t(T) * 4, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return (int)e;
    }
    if (n_isects <= 0) return LFS_OK;
    if (n_isects > 0x7FFFFFFFll) return LFS_E_UNSUPPORTED; // offsets / last_ids are int32 in the reference API
    if (!means2d || !radii || !depths || !isect_ids || !flatten_ids) return LFS_E_INVALID;
    const uint32_t tile_n_bits = bit_width_u32(tile_width * tile_height);
    const size_t total = size_t(C) * N;
    if (sort) {
        static bool big_lds_enabled = false; // > 64 KiB of dynamic LDS has to be opted into once per process
        if (!big_lds_enabled) {
            hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_sort_lds_kernel<1024>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
            if (ae != hipSuccess) return (int)ae;
            ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_sort_bins_kernel<256>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4096 * 8);
            if (ae != hipSuccess) return (int)ae;
            ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_sort_bins_kernel<1024, 1024, false, 64>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
            if (ae != hipSuccess) return (int)ae;
            ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&isect_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ROWS_STAGE * 8);
            if (ae != hipSuccess) return (int)ae;
            ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&isect_tiles_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TILES_CHUNK * 8);
            if (ae != hipSuccess) return (int)ae;
            big_lds_enabled = true;
        }
        const uint32_t pb = isect_per_block(total);
        const uint32_t blocks = uint32_t((total + pb - 1) / pb);
        int tok = lfs::prof_begin("isect_scatter", s);
        const uint32_t R = C * tile_height;
        const uint32_t idx_bits = bit_width_u32(uint32_t(total - 1)) ? bit_width_u32(uint32_t(total - 1)) : 1u;
        const bool two_pass = scratch != nullptr && isect_two_pass_supported(C, N, tile_width, tile_height);
        if (two_pass) {
            hipLaunchKernelGGL(isect_rows_kernel, dim3(uint32_t((total + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(1024), ROWS_STAGE * 8, s, C, N, means2d, radii,
                               depths, float(tile_size), tile_width, tile_height, idx_bits, w.offsets, w.row_cursor, reinterpret_cast<uint64_t*>(scratch),
                               guarded ? guard->abort_flag : nullptr);
            hipLaunchKernelGGL(isect_tiles_kernel, dim3(uint32_t((n_grid + TILES_CHUNK - 1) / TILES_CHUNK)), dim3(1024), TILES_CHUNK * 8, s, R, tile_width, idx_bits,
                               n_arg, w.offsets, w.cursor, reinterpret_cast<const uint64_t*>(scratch), isect_ids);
        } else if (guarded) { lfs::prof_end(tok, s); return LFS_E_UNSUPPORTED; } // (the one-pass scatter does not watch the abort flag)
        else if (size_t(T) * 8 <= LDS_HIST_LIMIT)
            hipLaunchKernelGGL(isect_scatter_kernel<true>, dim3(blocks), dim3(1024), size_t(T) * 8, s, C, N, pb, means2d, radii, depths,
                               float(tile_size), tile_width, tile_height, tile_n_bits, w.offsets, w.cursor, isect_ids, flatten_ids);
        else
            hipLaunchKernelGGL(isect_scatter_kernel<false>, dim3(blocks), dim3(1024), 0, s, C, N, pb, means2d, radii, depths,
                               float(tile_size), tile_width, tile_height, tile_n_bits, w.offsets, w.cursor, isect_ids, flatten_ids);
        lfs::prof_end(tok, s);
        lfs::ProfScope prof_sort("isect_tile_sort", s);
        const uint32_t n_tiles_ = tile_width * tile_height;
        // size classes (LDS sized to the class so that small tiles do not cap the occupancy): <= 1024 entries with the counting kernel on 256 bins
        // (256 threads, keys staged in LDS), <= 4096 on 512 bins with 512 threads and only the binned copy in LDS (32 KiB; measured against the staged
        // 256-thread form: 0.277 -> 0.127 ms per view at 12 M intersections, 0.058 -> 0.042 at 4.4 M; 1024 threads or the same change for the first class:
        // no further gain), <= 16384 on 1024 bins (1024 threads, 128 KiB LDS; a bin of more than 64 keys falls back to the bitonic network inside the
        // kernel), larger -> bitonic on global memory
        // (max_tile_isects >= 0: the longest tile list, from lfs_intersect_tile_count_ex - classes no tile falls into are not launched: ~9 us each at T = 8160)
        const int64_t longest = max_tile_isects >= 0 ? max_tile_isects : INT64_MAX;
#if LFS_SORT_2CLASS
        // (round 5) both classes in one launch whenever the second one is needed at all: its few long lists then run beside the many short ones instead of after them
        if (longest > 1024)
            hipLaunchKernelGGL(tile_sort_bins_2class_kernel, dim3(T), dim3(512), 4096 * 8, s, n_tiles_, tile_n_bits, w.offsets, isect_ids, flatten_ids);
        else
            hipLaunchKernelGGL(tile_sort_bins_kernel<256>, dim3(T), dim3(256), 2 * 1024 * 8, s, 1u, 1024u, n_tiles_, tile_n_bits, w.offsets, isect_ids, flatten_ids);
#else
        hipLaunchKernelGGL(tile_sort_bins_kernel<256>, dim3(T), dim3(256), 2 * 1024 * 8, s, 1u, 1024u, n_tiles_, tile_n_bits, w.offsets, isect_ids, flatten_ids);
        if (longest > 1024)
            hipLaunchKernelGGL((tile_sort_bins_kernel<512, 512, false, 32>), dim3(T), dim3(512), 4096 * 8, s, 1025u, 4096u, n_tiles_, tile_n_bits, w.offsets, isect_ids, flatten_ids);
#endif
        if (longest > 4096)
            hipLaunchKernelGGL((tile_sort_bins_kernel<1024, 1024, false, 64>), dim3(T), dim3(1024), 16384 * 8, s, 4097u, 16384u, n_tiles_, tile_n_bits, w.offsets, isect_ids, flatten_ids);
        if (longest > 16384)
            hipLaunchKernelGGL(tile_sort_global_kernel, dim3(T), dim3(1024), 0, s, 16385u, n_tiles_, tile_n_bits, w.offsets, isect_ids, flatten_ids);
    } else {
        if (!tiles_per_gauss) return LFS_E_INVALID;
        const uint32_t nb = uint32_t((total + SCAN_ITEMS - 1) / SCAN_ITEMS);
        hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nb), dim3(256), 0, s, total, tiles_per_gauss, w.block_sums);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, s, nb, w.block_sums);
        hipLaunchKernelGGL(isect_emit_unsorted_kernel, dim3(nb), dim3(256), 0, s, C, N, means2d, radii, depths, tiles_per_gauss,
                           w.block_sums, float(tile_size), tile_width, tile_height, tile_n_bits, isect_ids, flatten_ids);
    }
    return (int)hipGetLastError();
}

extern "C" int lfs_intersect_tile_emit_ex(
    uint32_t C, uint32_t N, const float* means2d, const int32_t* radii, const float* depths,
    uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, int sort, int64_t n_isects,
    const int32_t* tiles_per_gauss, int64_t* isect_ids, int32_t* flatten_ids, int32_t* tile_offsets, int64_t* scratch, int64_t max_tile_isects,
    void* workspace, size_t workspace_bytes, lfs_stream_t stream) {
    return lfs::isect_emit_impl(C, N, means2d, radii, depths, tile_size, tile_width, tile_height, sort, n_isects, tiles_per_gauss, isect_ids, flatten_ids, tile_offsets,
                                scratch, max_tile_isects, workspace, workspace_bytes, (hipStream_t)stream, nullptr);
}

extern "C" int lfs_intersect_tile_emit(
    uint32_t C, uint32_t N, const float* means2d, const int32_t* radii, const float* depths,
    uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, int sort, int64_t n_isects,
    const int32_t* tiles_per_gauss, int64_t* isect_ids, int32_t* flatten_ids, int32_t* tile_offsets,
    void* workspace, size_t workspace_bytes, lfs_stream_t stream) {
    return lfs_intersect_tile_emit_ex(C, N, means2d, radii, depths, tile_size, tile_width, tile_height, sort, n_isects, tiles_per_gauss, isect_ids, flatten_ids,
                                      tile_offsets, nullptr, -1, workspace, workspace_bytes, stream);
}

extern "C" int lfs_intersect_offset(
    int64_t n_isects, const int64_t* isect_ids, uint32_t C, uint32_t tile_width, uint32_t tile_height,
    int32_t* offsets, lfs_stream_t stream) {
    if (!offsets) return LFS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int64_t T = int64_t(C) * tile_width * tile_height;
    if (n_isects <= 0) { // IntersectTile.cu:268-271
        hipError_t e = hipMemsetAsync(offsets, 0, size_t(T) * 4, s);
        return (int)e;
    }
    if (!isect_ids) return LFS_E_INVALID;
    const uint32_t n_tiles = tile_width * tile_height;
    hipLaunchKernelGGL(isect_offset_kernel, dim3(uint32_t((n_isects + 255) / 256)), dim3(256), 0, s,
                       n_isects, isect_ids, n_tiles, bit_width_u32(n_tiles), T, offsets);
    return (int)hipGetLastError();
}
