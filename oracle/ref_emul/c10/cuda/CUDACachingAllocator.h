// ORACLE / TEST INFRASTRUCTURE ONLY: stands in for the CUDA-only torch header of the same name (IntersectTile.cu includes it for CUB_WRAPPER, whose call is in
// the launcher part that is not compiled here)
#pragma once
#include <cstddef>
#include <vector>
namespace c10::cuda::CUDACachingAllocator { // CUB_WRAPPER of gsplat/Common.h: get()->allocate(bytes).get()
    struct Block {
        std::vector<char> mem;
        void* get() { return mem.data(); }
    };
    struct Allocator {
        Block allocate(size_t n) { return Block{std::vector<char>(n ? n : 1)}; }
    };
    inline Allocator* get() {
        static Allocator a;
        return &a;
    }
} // namespace c10::cuda::CUDACachingAllocator
