// Per-tile list machinery shared by intersect.hip (gsplat::intersect_tile) and fastgs.hip (the EWA rasterizer's instance lists):
// the exclusive scan of per-tile counts and the per-tile sort of (depth bits << 32 | id) keys that a scatter kernel parked in
// a 64-bit array. The sort kernels write the reference's (camera | tile | depth) key back into that array and the id into a
// 32-bit array. See intersect.hip for the design notes.
#pragma once
#include "lfs_math.cuh"

namespace lfs {

// ---------------------------------------------------------------------------
// single-workgroup exclusive scan of totals[T] -> offsets[T+1] (int32), n_isects (int64)
// ---------------------------------------------------------------------------
// zero_totals: totals[] is left zero for the next call (zero-on-consume: the caller may then skip its memset); cursor / aux (nullable) are zeroed for
// the scatter kernels of THIS call; offsets_out (nullable, [T]) receives a second copy of the offsets (the caller's tile_offsets tensor); max_total
// (nullable) the longest tile list (the caller skips the sort launches of the size classes no tile falls into)
static __global__ void __launch_bounds__(1024) tile_scan_kernel(
    const uint32_t T, uint32_t* __restrict__ totals, int32_t* __restrict__ offsets, int64_t* __restrict__ n_isects,
    const bool zero_totals = false, uint32_t* __restrict__ cursor = nullptr, uint32_t* __restrict__ aux = nullptr, const uint32_t n_aux = 0,
    int32_t* __restrict__ offsets_out = nullptr, int64_t* __restrict__ max_total = nullptr, int64_t* __restrict__ stamp_out = nullptr, const int64_t stamp = 0,
    const int64_t capacity = -1, const uint32_t max_list = 0xFFFFFFFFu, int32_t* __restrict__ abort_flag = nullptr) {
    // capacity >= 0 (the speculative training step, csrc/gut_step.hip): the caller sized its list buffers for `capacity` intersections and launched the
    // per-tile sort classes up to `max_list` entries BEFORE these counts existed. When either assumption fails, *abort_flag = 1, every offset is
    // rewritten to 0 (all lists empty: nothing downstream indexes past its buffers) and the true counts are still reported - the host sees them after
    // it has enqueued the rest of the step, and runs the step again with buffers that fit.
    // slices of 8192 tiles staged in LDS: coalesced loads, every thread scans 8 consecutive values, coalesced stores (one slice = the whole array
    // at 1080p; the round-1 version walked slices of 1024 with three barriers each: 13 us at T = 8160; a register-blocked version without the LDS
    // transpose was slower still - 8-word strides between lanes make every store a partial 32-byte sector)
    constexpr uint32_t SLICE = 8192, PER = SLICE / 1024;
    __shared__ uint32_t vals[SLICE];
    __shared__ uint64_t wave_sums[16];
    __shared__ uint32_t wave_max[16];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (aux != nullptr) for (uint32_t i = threadIdx.x; i < n_aux; i += 1024) aux[i] = 0u;
    uint64_t carry = 0; uint32_t vmax = 0;
    for (uint32_t base = 0; base < T; base += SLICE) {
        const uint32_t n = min(SLICE, T - base);
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) { const uint32_t i = threadIdx.x + k * 1024; vals[i] = i < n ? totals[base + i] : 0u; }
        __syncthreads();
        uint32_t v[PER]; uint32_t local = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) { v[k] = vals[threadIdx.x * PER + k]; local += v[k]; vmax = max(vmax, v[k]); }
        uint64_t s = local; // inclusive wave scan of the threads' sums
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t o = __shfl_up(s, d, 64);
            if (int(lane) >= d) s += o;
        }
        if (lane == 63) wave_sums[wave] = s;
        __syncthreads();
        uint64_t run = carry + s - local, total = 0;
        for (uint32_t w = 0; w < 16; ++w) { if (w < wave) run += wave_sums[w]; total += wave_sums[w]; }
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) { vals[threadIdx.x * PER + k] = uint32_t(run); run += v[k]; } // (offsets are int32 in the reference API)
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t i = threadIdx.x + k * 1024;
            if (i < n) {
                const int32_t o = int32_t(vals[i]);
                offsets[base + i] = o;
                if (offsets_out != nullptr) offsets_out[base + i] = o;
                if (zero_totals) totals[base + i] = 0u;
                if (cursor != nullptr) cursor[base + i] = 0u;
            }
        }
        carry += total;
        __syncthreads();
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) vmax = max(vmax, uint32_t(__shfl_xor(int(vmax), m, 64)));
    if (lane == 0) wave_max[wave] = vmax;
    __syncthreads();
    uint32_t longest = 0;
    for (int w = 0; w < 16; ++w) longest = max(longest, wave_max[w]);
    const bool over = capacity >= 0 && (carry > uint64_t(capacity) || longest > max_list); // (uniform: every thread has the same carry and maxima)
    if (over) {
        for (uint32_t i = threadIdx.x; i < T; i += 1024) { offsets[i] = 0; if (offsets_out != nullptr) offsets_out[i] = 0; }
    }
    if (threadIdx.x == 0) {
        offsets[T] = over ? 0 : int32_t(carry); *n_isects = int64_t(carry);
        if (abort_flag != nullptr) *abort_flag = over ? 1 : 0;
        if (max_total != nullptr) *max_total = int64_t(longest);
        if (stamp_out != nullptr) { // the counts may live in pinned HOST memory: the stamp is written last, behind a system-scope fence - a host that
            LFS_SYSTEM_FENCE();     // sees the stamp of this call also sees its counts (it does not rely on the completion event alone)
            *reinterpret_cast<volatile int64_t*>(stamp_out) = stamp;
        }
    }
}

// Ascending bitonic network over n_pad (power of two) 64-bit keys in LDS; all THREADS threads of the workgroup call it.
// Barrier elision: pair index i = it * THREADS + wave * 64 + lane. For a flip with k <= 128 and for half-cleaners
// with j <= 64 the 64 pairs of a wave live inside ONE aligned block of 128 keys, the same block in every such stage,
// so consecutive wave-local stages only need the in-order LDS pipeline of that wave. A workgroup barrier is needed
// only around the stages that cross 128-key blocks (9 of the 55 stages at n_pad = 1024).
template <int THREADS>
LFS_DI void bitonic_sort_lds(uint64_t* __restrict__ keys, const uint32_t n_pad) {
    bool prev_cross = false;
    for (uint32_t k = 2; k <= n_pad; k <<= 1) {
        const bool flip_cross = k > 128;
        if (flip_cross || prev_cross) __syncthreads(); else LFS_WAVE_LOCKSTEP();
        prev_cross = flip_cross;
        for (uint32_t i = threadIdx.x; i < n_pad / 2; i += THREADS) { // flip step
            const uint32_t blk = i / (k >> 1), off = i % (k >> 1);
            const uint32_t a = blk * k + off, b = blk * k + (k - 1 - off);
            const uint64_t ka = keys[a], kb = keys[b];
            if (ka > kb) { keys[a] = kb; keys[b] = ka; }
        }
        for (uint32_t j = k >> 2; j >= 1; j >>= 1) {
            const bool cross = j >= 128;
            if (cross || prev_cross) __syncthreads(); else LFS_WAVE_LOCKSTEP();
            prev_cross = cross;
            for (uint32_t i = threadIdx.x; i < n_pad / 2; i += THREADS) {
                const uint32_t a = ((i / j) * (j << 1)) + (i % j), b = a + j;
                const uint64_t ka = keys[a], kb = keys[b];
                if (ka > kb) { keys[a] = kb; keys[b] = ka; }
            }
        }
    }
    __syncthreads();
}

// Per-tile sort, fast path: the keys of one tile are spread over a depth range, so ONE counting pass on a monotone
// NBINS-level quantisation of the (unsigned) depth bits leaves NBINS bins of ~n/NBINS keys, each finished by a single-thread
// rank count: ~10 LDS operations per key instead of the ~110 of the bitonic network (which was LDS-pipe bound:
// 0.17 ms at 4.4 M intersections). Exact: the quantisation is monotone in the key, ties fall into one bin, and a bin
// with more than BIN_LIMIT keys (degenerate depth distributions) sends the tile through the bitonic network instead.
// COPY: the tile's keys are staged in LDS (A) next to the binned copy (B); COPY = false (round 2: lists of 4 097 .. 16 384 keys, the
// common case at 3 M Gaussians / 1600x1200 where this stage was 0.53 ms per view on the bitonic path) re-reads them from global memory
// in the two passes instead, so only B (128 KB at 16 384 keys) lives in LDS.
// (the body: one workgroup of THREADS threads sorts the bucket [start, start + n) of tile t; lds64 = the workgroup's dynamic LDS block)
template <int THREADS, int NBINS, bool COPY, uint32_t BIN_LIMIT>
LFS_DI void tile_sort_bins_body(const uint32_t t, const uint32_t start, const uint32_t n, const uint32_t n_tiles, const uint32_t tile_n_bits,
                                int64_t* __restrict__ isect_ids, int32_t* __restrict__ flatten_ids, uint64_t* __restrict__ lds64) {
    __shared__ uint32_t s_hist[NBINS], s_off[NBINS + 1], s_minmax[2], s_big;
    static_assert(NBINS % 64 == 0, "one wave scans the bin counts");
    constexpr int PER = NBINS / 64;
    uint32_t n_pad = 2; while (n_pad < n) n_pad <<= 1;
    uint64_t* A = lds64;                          // [n_pad] input copy (COPY only)
    uint64_t* B = COPY ? lds64 + n_pad : lds64;   // [n_pad] binned / sorted
    const uint64_t hi_bits = ((uint64_t(t / n_tiles) << tile_n_bits) | uint64_t(t % n_tiles)) << 32;
    for (uint32_t b = threadIdx.x; b < uint32_t(NBINS); b += THREADS) s_hist[b] = 0u;
    if (threadIdx.x == 0) { s_minmax[0] = 0xFFFFFFFFu; s_minmax[1] = 0u; s_big = 0u; }
    __syncthreads();
    auto key_at = [&](uint32_t i) { return COPY ? A[i] : uint64_t(isect_ids[start + i]); };
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (uint32_t i = threadIdx.x; i < n; i += THREADS) {
        const uint64_t k = uint64_t(isect_ids[start + i]);
        if (COPY) A[i] = k;
        const uint32_t d = uint32_t(k >> 32);
        lo = min(lo, d); hi = max(hi, d);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { lo = min(lo, uint32_t(__shfl_xor(int(lo), m, 64))); hi = max(hi, uint32_t(__shfl_xor(int(hi), m, 64))); }
    if ((threadIdx.x & 63) == 0) { atomicMin(&s_minmax[0], lo); atomicMax(&s_minmax[1], hi); }
    __syncthreads();
    const uint32_t dmin = s_minmax[0];
    const float scale = float(NBINS) / (float(s_minmax[1] - dmin) + 1.f); // monotone map of the unsigned depth bits onto [0, NBINS)
    auto bin_of = [&](uint64_t k) { return min(uint32_t(NBINS - 1), uint32_t(float(uint32_t(k >> 32) - dmin) * scale)); };
    for (uint32_t i = threadIdx.x; i < n; i += THREADS) atomicAdd(&s_hist[bin_of(key_at(i))], 1u);
    __syncthreads();
    if (threadIdx.x < 64) { // exclusive scan of the NBINS counts by one wave (PER per lane)
        const uint32_t l = threadIdx.x;
        uint32_t c[PER], tot = 0, big = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) { c[q] = s_hist[PER * l + q]; tot += c[q]; big = max(big, c[q]); }
        uint32_t inc = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = uint32_t(__shfl_up(int(inc), d, 64)); if (int(l) >= d) inc += o; }
        uint32_t run = inc - tot;
#pragma unroll
        for (int q = 0; q < PER; ++q) { s_off[PER * l + q] = run; run += c[q]; s_hist[PER * l + q] = 0u; } // (s_hist: reused as fill cursors)
        if (l == 63) s_off[NBINS] = inc;
        if (big > BIN_LIMIT) atomicOr(&s_big, 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += THREADS) {
        const uint64_t k = key_at(i);
        const uint32_t b = bin_of(k);
        B[s_off[b] + atomicAdd(&s_hist[b], 1u)] = k;
    }
    for (uint32_t i = n + threadIdx.x; i < n_pad; i += THREADS) B[i] = ~0ull;
    __syncthreads();
    if (s_big) {
        bitonic_sort_lds<THREADS>(B, n_pad);
        for (uint32_t i = threadIdx.x; i < n; i += THREADS) {
            const uint64_t k = B[i];
            isect_ids[start + i] = int64_t(hi_bits | (k >> 32));
            flatten_ids[start + i] = int32_t(uint32_t(k));
        }
        return;
    }
    // rank inside the bin = number of smaller keys in it (keys are unique: a flatten id occurs once per tile); every key
    // goes straight to its final global position - no serial insertion chain, no further barrier
    for (uint32_t i = threadIdx.x; i < n; i += THREADS) {
        const uint64_t k = B[i];
        const uint32_t b = bin_of(k);
        const uint32_t o = s_off[b], m = s_off[b + 1] - o;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < m; ++j) rank += B[o + j] < k ? 1u : 0u;
        isect_ids[start + o + rank] = int64_t(hi_bits | (k >> 32));
        flatten_ids[start + o + rank] = int32_t(uint32_t(k));
    }
}

template <int THREADS, int NBINS = 256, bool COPY = true, uint32_t BIN_LIMIT = 32>
__global__ void __launch_bounds__(THREADS) tile_sort_bins_kernel(
    const uint32_t n_min, const uint32_t n_max, const uint32_t n_tiles, const uint32_t tile_n_bits, const int32_t* __restrict__ offsets,
    int64_t* __restrict__ isect_ids, int32_t* __restrict__ flatten_ids) {
    LFS_DYN_LDS(uint64_t, lds64);
    const uint32_t t = blockIdx.x;
    const uint32_t start = uint32_t(offsets[t]);
    const uint32_t n = uint32_t(offsets[t + 1]) - start;
    if (n < n_min || n > n_max) return;
    tile_sort_bins_body<THREADS, NBINS, COPY, BIN_LIMIT>(t, start, n, n_tiles, tile_n_bits, isect_ids, flatten_ids, lds64);
}

// The two size classes that hold every tile of a typical view - 1 .. 1024 entries (256 bins, keys staged in LDS) and 1025 .. 4096 (512 bins, only the binned copy in LDS) -
// in ONE launch of 512-thread workgroups (round 5): as two launches the second class - few, long lists - ran on its own for 23 us on SYN-B behind the 16 us of the first, most of
// it the tail of its longest tiles on a mostly idle chip; in one launch those tiles overlap with the thousands of short ones. Same bodies, same results (the body does not
// depend on which other tiles run beside it); 32 KiB of dynamic LDS per workgroup (the larger class) = 5 workgroups per CU.
static __global__ void __launch_bounds__(512) tile_sort_bins_2class_kernel(
    const uint32_t n_tiles, const uint32_t tile_n_bits, const int32_t* __restrict__ offsets, int64_t* __restrict__ isect_ids, int32_t* __restrict__ flatten_ids) {
    LFS_DYN_LDS(uint64_t, lds64);
    const uint32_t t = blockIdx.x;
    const uint32_t start = uint32_t(offsets[t]);
    const uint32_t n = uint32_t(offsets[t + 1]) - start;
    if (n < 1u || n > 4096u) return;   // (longer lists: the 1024-thread class / the global-memory network, launched separately when a tile needs them)
    if (n <= 1024u) tile_sort_bins_body<512, 256, true, 32>(t, start, n, n_tiles, tile_n_bits, isect_ids, flatten_ids, lds64);
    else tile_sort_bins_body<512, 512, false, 32>(t, start, n, n_tiles, tile_n_bits, isect_ids, flatten_ids, lds64);
}

// ---------------------------------------------------------------------------
// per-tile sort. Ascending-comparator bitonic network ("flip" formulation: the
// first step of each merge mirrors inside the block, the rest are half-cleaners),
// so +inf padding above n never moves below n.
// ---------------------------------------------------------------------------
template <int THREADS>
__global__ void __launch_bounds__(THREADS) tile_sort_lds_kernel(
    const uint32_t n_min, const uint32_t n_max, const uint32_t n_tiles, const uint32_t tile_n_bits, const int32_t* __restrict__ offsets,
    int64_t* __restrict__ isect_ids, int32_t* __restrict__ flatten_ids) {
    LFS_DYN_LDS(uint64_t, keys);
    const uint32_t t = blockIdx.x;
    const uint32_t start = uint32_t(offsets[t]);
    const uint32_t n = uint32_t(offsets[t + 1]) - start;
    if (n < n_min || n > n_max) return;
    uint32_t n_pad = 2; while (n_pad < n) n_pad <<= 1;
    // camera | tile: the same for the whole bucket (the scatter kernel left (depth << 32 | flatten id) in isect_ids)
    const uint64_t hi_bits = ((uint64_t(t / n_tiles) << tile_n_bits) | uint64_t(t % n_tiles)) << 32;
    for (uint32_t i = threadIdx.x; i < n_pad; i += THREADS) keys[i] = i < n ? uint64_t(isect_ids[start + i]) : ~0ull;
    __syncthreads();
    bitonic_sort_lds<THREADS>(keys, n_pad);
    for (uint32_t i = threadIdx.x; i < n; i += THREADS) {
        const uint64_t k = keys[i];
        isect_ids[start + i] = int64_t(hi_bits | (k >> 32));
        flatten_ids[start + i] = int32_t(uint32_t(k));
    }
}

// buckets too large for LDS: same network directly on the combined keys in global memory, then the same conversion
static __global__ void __launch_bounds__(1024) tile_sort_global_kernel(
    const uint32_t n_min, const uint32_t n_tiles, const uint32_t tile_n_bits, const int32_t* __restrict__ offsets, int64_t* isect_ids, int32_t* flatten_ids) {
    const uint32_t t = blockIdx.x;
    const uint32_t start = uint32_t(offsets[t]);
    const uint32_t n = uint32_t(offsets[t + 1]) - start;
    if (n < n_min) return;
    uint32_t n_pad = 2; while (n_pad < n) n_pad <<= 1;
    uint64_t* K = reinterpret_cast<uint64_t*>(isect_ids + start);
    auto cas = [&](uint32_t a, uint32_t b) {
        if (b >= n) return; // virtual +inf padding
        const uint64_t ka = K[a], kb = K[b];
        if (ka > kb) { K[a] = kb; K[b] = ka; }
    };
    for (uint32_t k = 2; k <= n_pad; k <<= 1) {
        for (uint32_t i = threadIdx.x; i < n_pad / 2; i += blockDim.x) {
            const uint32_t blk = i / (k >> 1), off = i % (k >> 1);
            cas(blk * k + off, blk * k + (k - 1 - off));
        }
        __syncthreads();
        for (uint32_t j = k >> 2; j >= 1; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < n_pad / 2; i += blockDim.x) {
                const uint32_t a = ((i / j) * (j << 1)) + (i % j);
                cas(a, a + j);
            }
            __syncthreads();
        }
    }
    const uint64_t hi_bits = ((uint64_t(t / n_tiles) << tile_n_bits) | uint64_t(t % n_tiles)) << 32;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint64_t k = K[i];
        flatten_ids[start + i] = int32_t(uint32_t(k));
        K[i] = hi_bits | (k >> 32);
    }
}

} // namespace lfs
