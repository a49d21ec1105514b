// ORACLE / TEST INFRASTRUCTURE ONLY: stands in for the CUDA-only torch header of the same name when the reference kernels are compiled as host code
#pragma once
#ifndef REF_NO_CUDA_EMUL // (host-only translation units - the reference's Camera class - take the stream stand-ins below without the kernel emulator)
#include "cuda_emul.h"
#endif
// core/camera.hpp holds a stream from the pool and guards one upload with it (ref_raster_shim.cpp: the reference's Camera class on the CPU)
namespace at::cuda {
    struct CUDAStream {
        void synchronize() const {}
        operator void*() const { return nullptr; } // as the stream argument of a (stand-in) cub call
    };
    inline CUDAStream getStreamFromPool(bool = false) { return {}; }
    struct CUDAStreamGuard {
        explicit CUDAStreamGuard(const CUDAStream&) {}
    };
    struct OptionalCUDAGuard { // DEVICE_GUARD of gsplat/Common.h
        template <class T> explicit OptionalCUDAGuard(T&&) {}
    };
    inline CUDAStream getCurrentCUDAStream() { return {}; }
} // namespace at::cuda
