// ORACLE / TEST INFRASTRUCTURE ONLY - the part of CUDA's <cooperative_groups.h> the reference kernels use, on top of cuda_emul.h.
#pragma once
#include "cuda_emul.h"

namespace cooperative_groups {

struct grid_group {
    unsigned long long thread_rank() const {
        const auto& b = cuemu::blk();
        const unsigned long long block_rank = (static_cast<unsigned long long>(b.bid.z) * b.gdim.y + b.bid.y) * b.gdim.x + b.bid.x;
        return block_rank * (b.bdim.x * b.bdim.y * b.bdim.z) + cuemu::cur()->flat;
    }
};
inline grid_group this_grid() { return grid_group(); }

struct thread_block {
    void sync() const { cuemu::block_barrier(0); }
    unsigned thread_rank() const { return unsigned(cuemu::cur()->flat); }
    unsigned size() const { const auto& b = cuemu::blk(); return b.bdim.x * b.bdim.y * b.bdim.z; }
    dim3 group_index() const { const auto& b = cuemu::blk(); return dim3(b.bid.x, b.bid.y, b.bid.z); }
    dim3 thread_index() const { const auto& t = cuemu::cur()->tid; return dim3(t.x, t.y, t.z); }
    dim3 group_dim() const { const auto& b = cuemu::blk(); return dim3(b.bdim.x, b.bdim.y, b.bdim.z); }
};
inline thread_block this_thread_block() { return thread_block(); }

template <unsigned SIZE, typename ParentT = void> struct thread_block_tile {
    static_assert(SIZE == 32, "the emulator models full 32-lane warps");
    unsigned thread_rank() const { return unsigned(cuemu::cur()->lane); }
    unsigned size() const { return SIZE; }
    bool any(int pred) const {
        return cuemu::warp_collective(pred ? 1 : 0, [](cuemu::WarpState& w) {
            uint64_t r = 0;
            for (int l = 0; l < cuemu::WARP; ++l) if ((w.live >> l) & 1) r |= w.slot[l];
            for (int l = 0; l < cuemu::WARP; ++l) w.result[l] = r;
        }) != 0;
    }
    bool all(int pred) const {
        return cuemu::warp_collective(pred ? 1 : 0, [](cuemu::WarpState& w) {
            uint64_t r = 1;
            for (int l = 0; l < cuemu::WARP; ++l) if ((w.live >> l) & 1) r &= w.slot[l];
            for (int l = 0; l < cuemu::WARP; ++l) w.result[l] = r;
        }) != 0;
    }
    void sync() const { cuemu::warp_collective(0, [](cuemu::WarpState& w) { for (int l = 0; l < cuemu::WARP; ++l) w.result[l] = 0; }); }
    unsigned meta_group_rank() const { return unsigned(cuemu::cur()->warp); }   // index of this tile inside its thread block
    // shfl / shfl_up: every lane deposits, the last arriver snapshots the slots, every lane then reads its source's slot (the next collective cannot
    // complete - and overwrite the snapshot - before every lane of the warp has arrived at it, i.e. has read this one)
    template <typename T> T shfl(T v, unsigned src) const {
        static thread_local uint64_t snap[cuemu::MAX_THREADS / cuemu::WARP][cuemu::WARP];
        uint64_t u = 0; static_assert(sizeof(T) <= 8, "payload"); memcpy(&u, &v, sizeof(T));
        const int wi = cuemu::cur()->warp;
        cuemu::warp_collective(u, [wi](cuemu::WarpState& w) { for (int l = 0; l < cuemu::WARP; ++l) { snap[wi][l] = w.slot[l]; w.result[l] = 0; } });
        T out; const uint64_t r = snap[wi][src & (cuemu::WARP - 1)]; memcpy(&out, &r, sizeof(T)); return out;
    }
    template <typename T> T shfl_up(T v, unsigned delta) const {   // lanes below delta keep their own value
        const unsigned me = unsigned(cuemu::cur()->lane);
        const T got = shfl(v, me >= delta ? me - delta : me);
        return me >= delta ? got : v;
    }
    template <typename T> T shfl_down(T v, unsigned delta) const {
        const unsigned me = unsigned(cuemu::cur()->lane);
        const T got = shfl(v, me + delta < SIZE ? me + delta : me);
        return me + delta < SIZE ? got : v;
    }
    unsigned ballot(int pred) const {
        return unsigned(cuemu::warp_collective(pred ? 1 : 0, [](cuemu::WarpState& w) {
            uint64_t r = 0;
            for (int l = 0; l < cuemu::WARP; ++l) if (((w.live >> l) & 1) && w.slot[l]) r |= 1ull << l;
            for (int l = 0; l < cuemu::WARP; ++l) w.result[l] = r;
        }));
    }
};
template <unsigned SIZE> inline thread_block_tile<SIZE> tiled_partition(const thread_block&) { return thread_block_tile<SIZE>(); }

template <typename T> struct plus { T operator()(T a, T b) const { return a + b; } };
template <typename T> struct greater { T operator()(T a, T b) const { return a > b ? a : b; } };   // cg::greater = max
template <typename T> struct less { T operator()(T a, T b) const { return a < b ? a : b; } };      // cg::less = min

namespace detail {
template <typename T> inline uint64_t pack(T v) { uint64_t u = 0; static_assert(sizeof(T) <= 8, "payload"); memcpy(&u, &v, sizeof(T)); return u; }
template <typename T> inline T unpack(uint64_t u) { T v; memcpy(&v, &u, sizeof(T)); return v; }
}

// reduce over a 32-lane tile: the xor butterfly (offsets 16, 8, 4, 2, 1) every lane of which ends with the same bits. A lane that has left
// the kernel contributes nothing (the reference's kernels keep whole warps alive through their reductions).
template <unsigned SIZE, typename P, typename T, typename Op> inline T reduce(const thread_block_tile<SIZE, P>&, T val, Op op) {
    const uint64_t r = cuemu::warp_collective(detail::pack(val), [op](cuemu::WarpState& w) {
        T v[cuemu::WARP];
        if (w.live != 0xffffffffu) { fprintf(stderr, "cuemu: reduce over a partially exited warp\n"); abort(); }
        for (int l = 0; l < cuemu::WARP; ++l) v[l] = detail::unpack<T>(w.slot[l]);
        for (int m = cuemu::WARP / 2; m >= 1; m >>= 1) {
            T n[cuemu::WARP];
            for (int l = 0; l < cuemu::WARP; ++l) n[l] = op(v[l], v[l ^ m]);
            for (int l = 0; l < cuemu::WARP; ++l) v[l] = n[l];
        }
        for (int l = 0; l < cuemu::WARP; ++l) w.result[l] = detail::pack(v[l]);
    });
    return detail::unpack<T>(r);
}

} // namespace cooperative_groups
