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
#include <algorithm>
#include "lfs_math.cuh"
#include "lfs_prof.h"
#include "lfs_tilelists.cuh"
#include "lfs_raster_pack.cuh"
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
// MASKED lists (round 4; the training step only - never behind gsplat::intersect_tile, whose outputs stay bit-exact with the reference).
// The reference lists a Gaussian in every tile of the bounding rectangle of its radii; for the small Gaussians of a trained scene most of those tiles are
// never touched by the alpha >= 1/255 ellipse (SYN-B: the per-cell conic test of raster_cull_kernel removed 86 % of the 4 cells x entries), yet every
// entry was scattered, sorted and then gathered again (a 32-byte culling record per entry, 1.36 x the algorithmic bytes of that kernel) only to be dropped.
// Here the conic test runs where the Gaussian's culling record is read ONCE, coalesced: per (Gaussian, tile) a 4-bit mask says which of the tile's 8x8
// cells can reach the alpha threshold (lfs_cull_conic.cuh: conservative, so the image and the gradients are exactly those of the full lists); tiles with an
// empty mask are not listed at all, the others carry the mask through binning and sort in the four bits below the Gaussian index, and the per-cell lists
// fall out of a gather-free compaction (raster.hip: cells_from_masks_kernel). What stays defined as in the reference: tiles_per_gauss / n_isects are still
// the rectangle counts (reported to the caller), the order inside a tile is (depth, Gaussian index).
//   isect_count_masked_kernel : lanes = (Gaussian, tile) PAIRS of the wavefront's 64 Gaussians (prefix sum + search: converged evaluation of the 4 conic tests,
//                               where a per-Gaussian loop over its rectangle would run every wavefront at the pace of its largest Gaussian); masks of
//                               rectangles of <= 16 tiles are stored (64 bits per Gaussian), larger ones are re-derived by the binning kernel.
// ---------------------------------------------------------------------------
struct CellFrame { float ax, bx, ay, by, mx, my; uint32_t cull_on; }; // u(j) = ax * j + bx at pixel column j's left edge + 0.5 (normalised camera x); margins mx, my
LFS_DI CellFrame cell_frame(const CamDev& cam, uint32_t cull_on) {
    CellFrame f;
    f.ax = 1.f / cam.fx; f.bx = (0.5f - cam.cx) / cam.fx; f.ay = 1.f / cam.fy; f.by = (0.5f - cam.cy) / cam.fy;
    f.mx = 0.25f / cam.fx; f.my = 0.25f / cam.fy; f.cull_on = cull_on;   // a quarter pixel of numerical margin, as raster_cull_kernel
    return f;
}
constexpr uint32_t NREF_SLOTS = 64; // the rectangle count (the reference's n_isects) is accumulated over this many addresses

// The cells of one ROW of 8x8 cells (cell row `crow` of the image) the Gaussian can reach: the global cell columns [clo, chi] (empty: clo > chi), from ONE
// interval computation (lfs_cull_conic.cuh: conic_strip_interval) - a Gaussian costs (cell rows of its rectangle) x ~50 operations, its tiles only integer
// compares. History of this stage on SYN-B (profiles/r04): per-(Gaussian, tile) 4-cell conic tests in a per-Gaussian loop 0.168 + 0.241 ms (count + binning:
// the few Gaussians with 20 - 60 tile rectangles hold their wavefronts), the same tests with lanes = (Gaussian, tile) pairs and stored masks 0.077 + 0.049 ms,
// this form (nothing stored, both kernels re-derive the row ranges) see DESIGN.md. `per_cell`: the strip form is degenerate for this record (a needle whose leading
// coefficient is within the tolerance of zero): the caller tests the row's cells one by one with conic_culled.
struct CellRowRange { int32_t clo, chi; bool per_cell; };
LFS_DI CellRowRange cell_row_range(const CellFrame& f, const ConicRec& k, const uint32_t crow) {
    CellRowRange r{-(1 << 28), 1 << 28, false};
    if (!f.cull_on) return r;
    const float i0 = float(crow * 8u);
    const float v0 = f.ay * i0 + f.by - f.my, v1 = f.ay * (i0 + 7.f) + f.by + f.my;
    float lo, hi;
    if (!conic_strip_interval(k, v0, v1, lo, hi)) { r.clo = 1; r.chi = 0; return r; }
    if (!(lo > -INFINITY) || !(hi < INFINITY)) { r.per_cell = k.g < INFINITY; return r; } // (a "never cull" record: every cell)
    // cell column c spans u in [ax 8c + bx - mx, ax (8c + 7) + bx + mx]; it meets [lo + px, hi + px] iff tl <= c <= th (a thousandth of a cell of slack for the rounding here)
    const float inv = 1.f / (8.f * f.ax);
    const float tl = (lo + k.px - f.bx - f.mx - 7.f * f.ax) * inv - 1e-3f, th = (hi + k.px - f.bx + f.mx) * inv + 1e-3f;
    r.clo = int32_t(ceilf(fminf(fmaxf(tl, -1e6f), 1e6f)));
    r.chi = int32_t(floorf(fminf(fmaxf(th, -1e6f), 1e6f)));
    return r;
}
// the 4-bit (WPS = 2) / 1-bit (WPS = 1) cell mask of tile column x from the ranges of the tile's cell rows; bit (cr * WPS + cc)
template <int WPS>
LFS_DI uint32_t tile_mask_from_rows(const CellFrame& f, const ConicRec& k, const CellRowRange (&rr)[WPS], const uint32_t x, const uint32_t y) {
    uint32_t mask = 0;
#pragma unroll
    for (int cr = 0; cr < WPS; ++cr)
#pragma unroll
        for (int cc = 0; cc < WPS; ++cc) {
            const int32_t col = int32_t(x * WPS + cc);
            bool on = col >= rr[cr].clo && col <= rr[cr].chi;
            if (rr[cr].per_cell) {
                const float j0 = float(col * 8), i0 = float((y * WPS + cr) * 8u);
                on = !conic_culled(k, f.ax * j0 + f.bx - f.mx, f.ax * (j0 + 7.f) + f.bx + f.mx, f.ay * i0 + f.by - f.my, f.ay * (i0 + 7.f) + f.by + f.my);
            }
            mask |= on ? 1u << (cr * WPS + cc) : 0u;
        }
    return mask;
}
// the tile columns [jlo, jhi) of the rectangle [x0, x1) a tile row can list at all (saves the empty iterations of wide rectangles)
template <int WPS>
LFS_DI void tile_cols_of_rows(const CellRowRange (&rr)[WPS], const uint32_t x0, const uint32_t x1, uint32_t& jlo, uint32_t& jhi) {
    int32_t lo = 1 << 28, hi = -(1 << 28);
    bool any_per_cell = false;
#pragma unroll
    for (int cr = 0; cr < WPS; ++cr) { any_per_cell |= rr[cr].per_cell; if (rr[cr].clo <= rr[cr].chi) { lo = min(lo, rr[cr].clo); hi = max(hi, rr[cr].chi); } }
    if (any_per_cell) { jlo = x0; jhi = x1; return; }
    if (lo > hi) { jlo = x0; jhi = x0; return; }
    const int32_t tlo = lo >= 0 ? lo / WPS : 0, thi = hi >= 0 ? hi / WPS : -1;
    jlo = uint32_t(min(max(tlo, int32_t(x0)), int32_t(x1)));
    jhi = uint32_t(min(max(thi + 1, int32_t(jlo)), int32_t(x1)));
}

template <int WPS, bool LDS_HIST>
__global__ void __launch_bounds__(1024) isect_count_masked_kernel(
    const uint32_t N, const uint32_t per_block, const float* __restrict__ means2d, const int32_t* __restrict__ radii, const CullRec* __restrict__ cull,
    const CamDev* __restrict__ cams, const float tile_size_f, const uint32_t tw, const uint32_t th, const uint32_t cull_on,
    uint32_t* __restrict__ totals, unsigned long long* __restrict__ n_ref_slots) {
    LFS_DYN_LDS(uint32_t, hist); // [T] (LDS_HIST)
    __shared__ unsigned long long s_ref[16];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t T = tw * th;
    if (LDS_HIST) {
        for (uint32_t t = threadIdx.x; t < T; t += 1024) hist[t] = 0u;
        __syncthreads();
    }
    const CellFrame fr = cell_frame(cams[0], cull_on);
    unsigned long long ref_sum = 0;
    const uint32_t begin = blockIdx.x * per_block, end = min(begin + per_block, N);
    for (uint32_t idx = begin + threadIdx.x; idx < end; idx += 1024) {
        TileRect rc;
        if (!tile_rect(means2d, radii, idx, tile_size_f, tw, th, rc) || rc.x1 <= rc.x0 || rc.y1 <= rc.y0) continue;
        ref_sum += (unsigned long long)(rc.x1 - rc.x0) * (rc.y1 - rc.y0);
        const CullRec cr = cull[idx];
        const ConicRec kr{cr.a.x, cr.a.y, cr.a.z, cr.a.w, cr.b.x, cr.b.y, cr.b.z, cr.b.w};
        for (uint32_t i = rc.y0; i < rc.y1; ++i) {
            CellRowRange rr[WPS];
#pragma unroll
            for (int c = 0; c < WPS; ++c) rr[c] = cell_row_range(fr, kr, i * WPS + c);
            uint32_t jlo, jhi;
            tile_cols_of_rows<WPS>(rr, rc.x0, rc.x1, jlo, jhi);
            for (uint32_t x = jlo; x < jhi; ++x)
                if (tile_mask_from_rows<WPS>(fr, kr, rr, x, i)) { if (LDS_HIST) atomicAdd(&hist[i * tw + x], 1u); else atomicAdd(&totals[i * tw + x], 1u); }
        }
    }
    if (LDS_HIST) {
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < T; t += 1024) { const uint32_t c = hist[t]; if (c) atomicAdd(&totals[t], c); }
    }
    // the reference's intersection count (rectangle areas) of this workgroup -> one of NREF_SLOTS addresses
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) ref_sum += __shfl_xor(ref_sum, m, 64);
    if (lane == 0) s_ref[wave] = ref_sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < 16; ++w) t += s_ref[w];
        if (t) atomicAdd(&n_ref_slots[blockIdx.x & (NREF_SLOTS - 1)], t);
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

// Entry layout of the two passes (64 bits):  depth bits << dshift | tile_x << xshift | Gaussian index << pshift | cell mask.
//   reference lists (gsplat::intersect_tile): dshift 32, xshift = bits(C N - 1), pshift 0, no mask
//   masked lists (training step)            : pshift 4, xshift = bits(N - 1) + 4, dshift = max(32, xshift + bits(tile_w - 1)) <= 33 - a depth behind the near
//                                             plane is positive, its sign bit is free: one more bit for the payload (3 M Gaussians at 100 tile columns need 33)
struct EntryFmt { uint32_t dshift, xshift, pshift; };
template <bool MASKED, int WPS>
__global__ void __launch_bounds__(1024) isect_rows_kernel(
    const uint32_t C, const uint32_t N, const float* __restrict__ means2d, const int32_t* __restrict__ radii, const float* __restrict__ depths,
    const float tile_size_f, const uint32_t tw, const uint32_t th, const EntryFmt fmt,
    const int32_t* __restrict__ offsets, uint32_t* __restrict__ row_cursor, uint64_t* __restrict__ scratch, const int32_t* __restrict__ abort_flag = nullptr,
    const CullRec* __restrict__ cull = nullptr, const CamDev* __restrict__ cams = nullptr, const uint32_t cull_on = 1) {
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
    // MASKED: the cell ranges of a tile row are re-derived here from the culling record (cell_row_range: one interval per row of cells) - twice, for the count
    // and for the placement: cheaper than storing them between the two kernels and between the two phases
    ConicRec kr{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    CellFrame fr{};
    if (MASKED && have) { const CullRec cr = cull[begin]; kr = ConicRec{cr.a.x, cr.a.y, cr.a.z, cr.a.w, cr.b.x, cr.b.y, cr.b.z, cr.b.w}; fr = cell_frame(cams[0], cull_on); }
    if (have) for (uint32_t i = rc.y0; i < rc.y1; ++i) {
        uint32_t c = nx;
        if (MASKED) {
            CellRowRange rr[WPS];
#pragma unroll
            for (int q = 0; q < WPS; ++q) rr[q] = cell_row_range(fr, kr, i * WPS + q);
            uint32_t jlo, jhi;
            tile_cols_of_rows<WPS>(rr, rc.x0, rc.x1, jlo, jhi);
            c = 0;
            for (uint32_t x = jlo; x < jhi; ++x) c += tile_mask_from_rows<WPS>(fr, kr, rr, x, i) != 0u;
        }
        if (c) atomicAdd(&cnt[rb + i], c);
    }
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < R; r += 1024) { const uint32_t c = cnt[r]; gbase[r] = c ? atomicAdd(&row_cursor[r], c) : 0u; }
    block_scan_1024(cnt, lbase, R, tmp);
    const uint32_t E = lbase[R];
    const bool staged = E <= ROWS_STAGE;
    for (uint32_t r = threadIdx.x; r < R; r += 1024) cnt[r] = 0u; // now the running rank inside the row
    __syncthreads();
    if (have) {
        const uint64_t hi = uint64_t(__float_as_uint(depths[begin])) << fmt.dshift | uint64_t(uint32_t(begin)) << fmt.pshift;
        for (uint32_t i = rc.y0; i < rc.y1; ++i) {
            const uint32_t row = rb + i;
            if (!MASKED) {
                const uint32_t k = atomicAdd(&cnt[row], nx);
                uint64_t* dst = staged ? stage + lbase[row] + k : scratch + size_t(offsets[size_t(row) * tw]) + gbase[row] + k;
                for (uint32_t j = 0; j < nx; ++j) dst[j] = hi | (uint64_t(rc.x0 + j) << fmt.xshift);
            } else {
                CellRowRange rr[WPS];
#pragma unroll
                for (int q = 0; q < WPS; ++q) rr[q] = cell_row_range(fr, kr, i * WPS + q);
                uint32_t jlo, jhi;
                tile_cols_of_rows<WPS>(rr, rc.x0, rc.x1, jlo, jhi);
                uint32_t c = 0, masks = 0; // (up to 8 tiles' masks are kept from the counting pass of this row; longer rows re-derive them)
                for (uint32_t x = jlo; x < jhi; ++x) { const uint32_t m = tile_mask_from_rows<WPS>(fr, kr, rr, x, i); c += m != 0u; if (x - jlo < 8u) masks |= m << ((x - jlo) * 4u); }
                if (c == 0) continue;
                const uint32_t k = atomicAdd(&cnt[row], c);
                uint64_t* dst = staged ? stage + lbase[row] + k : scratch + size_t(offsets[size_t(row) * tw]) + gbase[row] + k;
                uint32_t w = 0;
                for (uint32_t x = jlo; x < jhi; ++x) {
                    const uint32_t m = (x - jlo < 8u) ? (masks >> ((x - jlo) * 4u)) & 15u : tile_mask_from_rows<WPS>(fr, kr, rr, x, i);
                    if (m) dst[w++] = hi | (uint64_t(x) << fmt.xshift) | uint64_t(m);
                }
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
    const uint32_t R, const uint32_t tw, const EntryFmt fmt, const int64_t n_isects_arg,
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
    // out: depth bits << dshift | payload (everything below the tile column: Gaussian index [<< 4 | cell mask])
    const uint64_t low_mask = (uint64_t(1) << fmt.dshift) - 1, pay_mask = (uint64_t(1) << fmt.xshift) - 1;
    if (span > TILES_SPAN) { // a chunk across many (mostly empty) rows: plain scatter
        for (int64_t p = p0 + threadIdx.x; p < p1; p += 1024) {
            const uint64_t e = scratch[p];
            uint32_t r = r_lo;
            while (r < r_hi && int64_t(offsets[size_t(r + 1) * tw]) <= p) ++r;
            const uint32_t t = r * tw + uint32_t((e & low_mask) >> fmt.xshift);
            const uint32_t slot = atomicAdd(&cursor[t], 1u);
            isect_ids[size_t(offsets[t]) + slot] = int64_t((e & ~low_mask) | (e & pay_mask));
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
            const uint32_t tl = (r - r_lo) * tw + uint32_t((e & low_mask) >> fmt.xshift);
            ent[k] = (e & ~low_mask) | (e & pay_mask);
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
    unsigned long long* nref; // [NREF_SLOTS] partial sums of the rectangle counts (masked lists); directly behind totals: one clear covers both
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
    w.nref = (unsigned long long*)(p + o); o += align256(NREF_SLOTS * 8);
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
                           max_tile_isects, stamp_out, stamp, guard->capacity, sort_class_limit(guard->assumed_longest), guard->abort_flag,
                           (unsigned long long*)nullptr, 0u, guard->n_ref_out);
    } else
        hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, s, T, w.totals, w.offsets, n_isects, true, w.cursor, w.row_cursor, C * tile_height, tile_offsets,
                           max_tile_isects, stamp_out, stamp);
    return (int)hipGetLastError();
}

uint32_t* lfs::isect_workspace_totals(void* workspace, uint32_t C, uint32_t N, uint32_t tile_width, uint32_t tile_height, uint32_t* n_words) {
    const IsectWs w = isect_ws(workspace, C, N, tile_width, tile_height);
    // the per-tile totals and, directly behind them, the rectangle-count slots of the masked lists: what a step's first kernel clears
    if (n_words) *n_words = uint32_t((reinterpret_cast<const char*>(w.nref + NREF_SLOTS) - reinterpret_cast<const char*>(w.totals)) / 4);
    return w.totals;
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

static int enable_big_lds() { // > 64 KiB of dynamic LDS has to be opted into once per process
    static bool big_lds_enabled = false;
    if (big_lds_enabled) return LFS_OK;
    hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_sort_lds_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
    if (ae != hipSuccess) return (int)ae;
    ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_sort_bins_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4096 * 8);
    if (ae != hipSuccess) return (int)ae;
    ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_sort_bins_kernel<1024, 1024, false, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
    if (ae != hipSuccess) return (int)ae;
    ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&isect_rows_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, ROWS_STAGE * 8);
    if (ae != hipSuccess) return (int)ae;
    ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&isect_rows_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, ROWS_STAGE * 8);
    if (ae != hipSuccess) return (int)ae;
    ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&isect_rows_kernel<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, ROWS_STAGE * 8);
    if (ae != hipSuccess) return (int)ae;
    ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&isect_tiles_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TILES_CHUNK * 8);
    if (ae != hipSuccess) return (int)ae;
    big_lds_enabled = true;
    return LFS_OK;
}

// The per-tile sort, by size class (LDS sized to the class so that small tiles do not cap the occupancy): <= 1024 entries with the counting kernel on 256 bins
// (256 threads, keys staged in LDS), <= 4096 on 512 bins with 512 threads and only the binned copy in LDS (32 KiB; measured against the staged
// 256-thread form: 0.277 -> 0.127 ms per view at 12 M intersections, 0.058 -> 0.042 at 4.4 M; 1024 threads or the same change for the first class:
// no further gain), <= 16384 on 1024 bins (1024 threads, 128 KiB LDS; a bin of more than 64 keys falls back to the bitonic network inside the
// kernel), larger -> bitonic on global memory. `longest` (the longest tile list when known): classes no tile falls into are not launched (~9 us each at T = 8160).
static void launch_tile_sorts(uint32_t T, uint32_t n_tiles, uint32_t tile_n_bits, int64_t longest, const int32_t* offsets, int64_t* isect_ids, int32_t* flatten_ids,
                              uint32_t dshift, uint32_t payload_only, hipStream_t s) {
    hipLaunchKernelGGL(tile_sort_bins_kernel<256>, dim3(T), dim3(256), 2 * 1024 * 8, s, 1u, 1024u, n_tiles, tile_n_bits, offsets, isect_ids, flatten_ids, dshift, payload_only);
    if (longest > 1024)
        hipLaunchKernelGGL((tile_sort_bins_kernel<512, 512, false, 32>), dim3(T), dim3(512), 4096 * 8, s, 1025u, 4096u, n_tiles, tile_n_bits, offsets, isect_ids, flatten_ids, dshift, payload_only);
    if (longest > 4096)
        hipLaunchKernelGGL((tile_sort_bins_kernel<1024, 1024, false, 64>), dim3(T), dim3(1024), 16384 * 8, s, 4097u, 16384u, n_tiles, tile_n_bits, offsets, isect_ids, flatten_ids, dshift, payload_only);
    if (longest > 16384)
        hipLaunchKernelGGL(tile_sort_global_kernel, dim3(T), dim3(1024), 0, s, 16385u, n_tiles, tile_n_bits, offsets, isect_ids, flatten_ids, dshift, payload_only);
}

// ---- masked lists of the training step (see isect_count_masked_kernel) ------------------------------------------------------------------------------
// One camera, pinhole, tile size 8 or 16 (at most 4 cells per tile: the 4 mask bits), bits(N - 1) + 4 + bits(tile_w - 1) <= 33.
bool lfs::isect_masked_supported(uint32_t N, uint32_t tile_size, uint32_t tile_width, uint32_t tile_height) {
    if (tile_size != 16 && tile_size != 8) return false;
    if (N == 0 || tile_height > ROWS_MAX) return false;
    const uint32_t idx_bits = bit_width_u32(N - 1) ? bit_width_u32(N - 1) : 1u;
    return idx_bits + 4 + bit_width_u32(tile_width - 1) <= 33 && !(lfs_get_debug_flags() & (32u | 64u)); // debug bit 6: the reference lists + cull kernel (A/B, tests)
}

// counts: [0] intersections LISTED (what `capacity` has to hold), [1] the longest tile list, [2] stamp, [3] the reference's n_isects (rectangle areas)
int lfs::isect_masked_lists_impl(uint32_t N, const float* means2d, const int32_t* radii, const float* depths, const void* cull_recs, const void* cams_dev,
                                 uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, int64_t* bucket /* [capacity] 8-byte entries */,
                                 int32_t* payload_out /* [capacity]: sorted (Gaussian index << 4 | cell mask) */, int64_t* scratch /* [capacity] */, int64_t* counts,
                                 int64_t stamp, void* workspace, size_t workspace_bytes, hipStream_t s, const IsectGuard* guard) {
    if (!guard || !guard->abort_flag || guard->capacity <= 0 || !counts || !workspace || !bucket || !payload_out || !scratch || !cull_recs || !cams_dev) return LFS_E_INVALID;
    if (!isect_masked_supported(N, tile_size, tile_width, tile_height)) return LFS_E_UNSUPPORTED;
    if (guard->capacity > 0x7FFFFFFFll) return LFS_E_UNSUPPORTED;
    IsectWs w = isect_ws(workspace, 1, N, tile_width, tile_height);
    if (workspace_bytes < w.bytes) return LFS_E_WORKSPACE;
    int rc = enable_big_lds();
    if (rc) return rc;
    const uint32_t T = tile_width * tile_height, R = tile_height;
    const uint32_t idx_bits = bit_width_u32(N - 1) ? bit_width_u32(N - 1) : 1u;
    EntryFmt fmt; fmt.pshift = 4; fmt.xshift = idx_bits + 4; fmt.dshift = fmt.xshift + bit_width_u32(tile_width - 1) > 32 ? 33u : 32u;
    const uint32_t cull_on = (lfs_get_debug_flags() & 1u) ? 0u : 1u;
    const CullRec* cull = static_cast<const CullRec*>(cull_recs);
    const CamDev* cams = static_cast<const CamDev*>(cams_dev);
    {
        lfs::ProfScope prof("isect_count_scan", s);
        const uint32_t pb = isect_per_block(N);
        const dim3 grid((N + pb - 1) / pb);
        const bool lds_hist = size_t(T) * 4 <= LDS_HIST_LIMIT;
#define LFS_COUNT_MASKED(WPS, LH)                                                                                                                         \
    hipLaunchKernelGGL((isect_count_masked_kernel<WPS, LH>), grid, dim3(1024), (LH) ? size_t(T) * 4 : 0, s, N, pb, means2d, radii, cull, cams, float(tile_size), \
                       tile_width, tile_height, cull_on, w.totals, w.nref)
        if (tile_size == 16) { if (lds_hist) LFS_COUNT_MASKED(2, true); else LFS_COUNT_MASKED(2, false); }
        else { if (lds_hist) LFS_COUNT_MASKED(1, true); else LFS_COUNT_MASKED(1, false); }
#undef LFS_COUNT_MASKED
        hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, s, T, w.totals, w.offsets, counts, true, w.cursor, w.row_cursor, R, (int32_t*)nullptr,
                           counts + 1, counts + 2, stamp, guard->capacity, sort_class_limit(guard->assumed_longest), guard->abort_flag, w.nref, NREF_SLOTS, counts + 3);
    }
    {
        lfs::ProfScope prof("isect_scatter", s);
        const dim3 rgrid(uint32_t((size_t(N) + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
        if (tile_size == 16)
            hipLaunchKernelGGL((isect_rows_kernel<true, 2>), rgrid, dim3(1024), ROWS_STAGE * 8, s, 1u, N, means2d, radii, depths, float(tile_size), tile_width, tile_height, fmt,
                               w.offsets, w.row_cursor, reinterpret_cast<uint64_t*>(scratch), guard->abort_flag, cull, cams, cull_on);
        else
            hipLaunchKernelGGL((isect_rows_kernel<true, 1>), rgrid, dim3(1024), ROWS_STAGE * 8, s, 1u, N, means2d, radii, depths, float(tile_size), tile_width, tile_height, fmt,
                               w.offsets, w.row_cursor, reinterpret_cast<uint64_t*>(scratch), guard->abort_flag, cull, cams, cull_on);
        hipLaunchKernelGGL(isect_tiles_kernel, dim3(uint32_t((guard->capacity + TILES_CHUNK - 1) / TILES_CHUNK)), dim3(1024), TILES_CHUNK * 8, s, R, tile_width, fmt,
                           int64_t(-1), w.offsets, w.cursor, reinterpret_cast<const uint64_t*>(scratch), bucket);
    }
    lfs::ProfScope prof_sort("isect_tile_sort", s);
    int64_t longest = int64_t(sort_class_limit(guard->assumed_longest));
    if (longest == int64_t(0xFFFFFFFFu)) longest = INT64_MAX;
    launch_tile_sorts(T, T, bit_width_u32(T), longest, w.offsets, bucket, payload_out, fmt.dshift, 1u, s);
    return (int)hipGetLastError();
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
        { const int lrc = enable_big_lds(); if (lrc) return lrc; }
        const uint32_t pb = isect_per_block(total);
        const uint32_t blocks = uint32_t((total + pb - 1) / pb);
        int tok = lfs::prof_begin("isect_scatter", s);
        const uint32_t R = C * tile_height;
        const uint32_t idx_bits = bit_width_u32(uint32_t(total - 1)) ? bit_width_u32(uint32_t(total - 1)) : 1u;
        const bool two_pass = scratch != nullptr && R <= ROWS_MAX && total <= 0xFFFFFFFFull && idx_bits + bit_width_u32(tile_width - 1) <= 32 &&
                              !(lfs_get_debug_flags() & 32u);
        if (two_pass) {
            const EntryFmt fmt{32u, idx_bits, 0u};
            hipLaunchKernelGGL((isect_rows_kernel<false, 2>), dim3(uint32_t((total + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(1024), ROWS_STAGE * 8, s, C, N, means2d, radii,
                               depths, float(tile_size), tile_width, tile_height, fmt, w.offsets, w.row_cursor, reinterpret_cast<uint64_t*>(scratch),
                               guarded ? guard->abort_flag : nullptr);
            hipLaunchKernelGGL(isect_tiles_kernel, dim3(uint32_t((n_grid + TILES_CHUNK - 1) / TILES_CHUNK)), dim3(1024), TILES_CHUNK * 8, s, R, tile_width, fmt,
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
        // (max_tile_isects >= 0: the longest tile list, from lfs_intersect_tile_count_ex)
        launch_tile_sorts(T, n_tiles_, tile_n_bits, max_tile_isects >= 0 ? max_tile_isects : INT64_MAX, w.offsets, isect_ids, flatten_ids, 32u, 0u, s);
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
