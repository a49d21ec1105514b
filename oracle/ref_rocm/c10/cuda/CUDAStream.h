// ORACLE/_ref - TEST INFRASTRUCTURE ONLY (make reflink: the reference's own L2 host code compiled against ROCm libtorch and LINKED to the product backend).
// PyTorch-ROCm ships the CUDA-named stream / guard classes under their "masquerading" HIP names (torch's own hipify pass renames them when it builds extension
// host code); this header is that rename for the two names core/camera.hpp and core/camera.cpp use, so the reference's sources compile unmodified with g++.
#pragma once
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/hip/HIPStream.h>
namespace at::cuda {
    using CUDAStream = c10::hip::HIPStreamMasqueradingAsCUDA;
    inline CUDAStream getStreamFromPool(const bool isHighPriority = false, c10::DeviceIndex device = -1) {
        return c10::hip::getStreamFromPoolMasqueradingAsCUDA(isHighPriority, device);
    }
    inline CUDAStream getCurrentCUDAStream(c10::DeviceIndex device = -1) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(device); }
    using CUDAStreamGuard = c10::hip::HIPStreamGuardMasqueradingAsCUDA;
    using OptionalCUDAGuard = c10::hip::OptionalHIPGuardMasqueradingAsCUDA;
    using CUDAGuard = c10::hip::HIPGuardMasqueradingAsCUDA;
} // namespace at::cuda
