// ORACLE / TEST INFRASTRUCTURE ONLY. The slice of CUB (third-party, part of the CUDA toolkit, not in the reference tree) that the reference's fastgs host code
// calls, restated from its published semantics: cub::DoubleBuffer, DeviceRadixSort::SortPairs (a STABLE ascending sort of the key bits [begin_bit, end_bit),
// result in the buffer the selector points to afterwards), DeviceScan::ExclusiveSum / InclusiveSum. A null workspace pointer = size query.
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>
#include "../cuda_emul.h"
namespace cub {
template <typename T> struct DoubleBuffer {
    T* d_buffers[2]; int selector;
    DoubleBuffer() : d_buffers{nullptr, nullptr}, selector(0) {}
    DoubleBuffer(T* cur, T* alt) : d_buffers{cur, alt}, selector(0) {}
    T* Current() { return d_buffers[selector]; }
    T* Alternate() { return d_buffers[selector ^ 1]; }
};
struct DeviceRadixSort {
    template <typename K, typename V>
    static int SortPairs(void* ws, size_t& ws_bytes, DoubleBuffer<K>& keys, DoubleBuffer<V>& values, int n, int begin_bit = 0, int end_bit = int(sizeof(K) * 8), void* = nullptr) {
        if (ws == nullptr) { ws_bytes = 1; return 0; }
        const unsigned long long mask = (end_bit - begin_bit >= 64) ? ~0ull : ((1ull << (end_bit - begin_bit)) - 1ull);
        std::vector<int> order(size_t(n > 0 ? n : 0));
        std::iota(order.begin(), order.end(), 0);
        K* kc = keys.Current(); V* vc = values.Current();
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ((static_cast<unsigned long long>(kc[a]) >> begin_bit) & mask) < ((static_cast<unsigned long long>(kc[b]) >> begin_bit) & mask); });
        K* ka = keys.Alternate(); V* va = values.Alternate();
        for (int i = 0; i < n; ++i) { ka[i] = kc[order[size_t(i)]]; va[i] = vc[order[size_t(i)]]; }
        keys.selector ^= 1; values.selector ^= 1;   // (CUB leaves the result in either buffer and reports it through the selector)
        return 0;
    }
};
struct DeviceScan {
    template <typename In, typename Out> static int ExclusiveSum(void* ws, size_t& ws_bytes, In in, Out out, int n, void* = nullptr) {
        if (ws == nullptr) { ws_bytes = 1; return 0; }
        auto run = decltype(+in[0])(0);
        for (int i = 0; i < n; ++i) { const auto v = in[i]; out[i] = run; run += v; }
        return 0;
    }
    template <typename In, typename Out> static int InclusiveSum(void* ws, size_t& ws_bytes, In in, Out out, int n, void* = nullptr) {
        if (ws == nullptr) { ws_bytes = 1; return 0; }
        auto run = decltype(+in[0])(0);
        for (int i = 0; i < n; ++i) { run += in[i]; out[i] = run; }
        return 0;
    }
};
// cub::BlockReduce<T, BLOCK_X, ALGORITHM, BLOCK_Y>(temp).Reduce(value, op): the block-wide reduction, valid in thread 0 (linear thread rank)
enum BlockReduceAlgorithm { BLOCK_REDUCE_RAKING_COMMUTATIVE_ONLY, BLOCK_REDUCE_RAKING, BLOCK_REDUCE_WARP_REDUCTIONS };
template <typename T, int BX, BlockReduceAlgorithm = BLOCK_REDUCE_WARP_REDUCTIONS, int BY = 1, int BZ = 1> struct BlockReduce {
    struct TempStorage { T vals[BX * BY * BZ]; };
    TempStorage& t;
    explicit BlockReduce(TempStorage& ts) : t(ts) {}
    template <typename Op> T Reduce(T v, Op op) {
        const int me = cuemu::cur()->flat;
        t.vals[me] = v;
        cuemu::block_barrier(0);
        T r = v;
        if (me == 0) { r = t.vals[0]; for (int i = 1; i < BX * BY * BZ; ++i) r = op(r, t.vals[i]); }
        cuemu::block_barrier(0);
        return r;
    }
    T Sum(T v) { return Reduce(v, [](T a, T b) { return a + b; }); }
};
} // namespace cub
namespace thrust {
template <typename T> struct maximum { T operator()(const T& a, const T& b) const { return a < b ? b : a; } };
} // namespace thrust
