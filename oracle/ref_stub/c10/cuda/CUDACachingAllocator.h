// ORACLE/_ref - TEST INFRASTRUCTURE ONLY. default_strategy.cpp includes this header for an emptyCache() call that only exists under _WIN32.
#pragma once
namespace c10::cuda::CUDACachingAllocator { inline void emptyCache() {} }
