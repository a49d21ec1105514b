// ORACLE / TEST INFRASTRUCTURE ONLY - never part of the product.
// <cuda_runtime.h> for the reference's fastgs rasterizer (fastgs/rasterization: kernels_*.cuh, forward.cu, backward.cu, compiled in place as host C++
// by `make -C oracle refk_fastgs`): CUDA's built-in vector types and their make_* constructors, the handful of runtime calls the host code makes
// (device memory IS host memory here), and the warp-level intrinsics of the kernels on top of cuda_emul.h's fibers (32-lane warps).
#pragma once
#include "cuda_emul.h"

#define __align__(n) alignas(n)
// static __shared__ arrays: ONE workgroup runs at a time, so function-local static storage is exactly what the fibers of that workgroup share
// (cuda_emul.h leaves __shared__ empty for the `extern __shared__` arrays of the gsplat kernels; the fastgs kernels declare sized arrays only)
#undef __shared__
#define __shared__ static
typedef unsigned int uint;
typedef unsigned short ushort;

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct int4 { int x, y, z, w; };
struct uint4 { unsigned x, y, z, w; };
struct ushort4 { unsigned short x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline int3 make_int3(int x, int y, int z) { return {x, y, z}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline uint3 make_uint3(unsigned x, unsigned y, unsigned z) { return {x, y, z}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }
static inline ushort4 make_ushort4(unsigned short x, unsigned short y, unsigned short z, unsigned short w) { return {x, y, z, w}; }

// ---- runtime: one address space, no streams -----------------------------------------------------------------------------------------
typedef void* cudaStream_t;
enum cudaMemcpyKind { cudaMemcpyHostToHost, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice };
static inline cudaError_t cudaMemset(void* p, int v, size_t n) { memset(p, v, n); return cudaSuccess; }
static inline cudaError_t cudaMemsetAsync(void* p, int v, size_t n, cudaStream_t = nullptr) { memset(p, v, n); return cudaSuccess; }
static inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { memcpy(d, s, n); return cudaSuccess; }
static inline cudaError_t cudaStreamCreate(cudaStream_t* s) { *s = nullptr; return cudaSuccess; }
static inline cudaError_t cudaStreamSynchronize(cudaStream_t) { return cudaSuccess; }
static inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
static inline const char* cudaGetErrorString(cudaError_t) { return "no error"; }

static inline size_t max(size_t a, size_t b) { return a > b ? a : b; }
static inline size_t min(size_t a, size_t b) { return a < b ? a : b; }
static inline unsigned atomicAdd(unsigned* p, int v) { const unsigned o = *p; *p = o + unsigned(v); return o; }

// ---- kernel<<<grid, block>>>(args...): the recipe rewrites it to cuemu::launcher(kernel, grid, block)(args...) ------------------------
namespace cuemu {
template <class K> struct Launcher {
    K k; dim3 g, b;
    template <class... A> void operator()(A... a) const { if (g.x * g.y * g.z == 0) return; launch(g, b, true, [&]() { k(a...); }); }
};
template <class K> Launcher<K> launcher(K k, dim3 g, dim3 b) { return Launcher<K>{k, g, b}; }
template <class K> Launcher<K> launcher(K k, long long g, long long b) { return Launcher<K>{k, dim3(unsigned(g)), dim3(unsigned(b))}; }
} // namespace cuemu
enum cudaFuncAttribute { cudaFuncAttributeMaxDynamicSharedMemorySize = 8 };
template <class K> static inline cudaError_t cudaFuncSetAttribute(K, cudaFuncAttribute, long long) { return cudaSuccess; } // the emulator's arena is static
namespace cuemu {
} // namespace cuemu

// ---- warp intrinsics (full-mask forms; a lane that has returned contributes 0 / false, as on the GPU for an exited lane) ----------------
static inline unsigned __ballot_sync(unsigned, int pred) {
    return unsigned(cuemu::warp_collective(pred ? 1 : 0, [](cuemu::WarpState& w) {
        uint64_t r = 0;
        for (int l = 0; l < cuemu::WARP; ++l) if (((w.live >> l) & 1) && w.slot[l]) r |= 1ull << l;
        for (int l = 0; l < cuemu::WARP; ++l) w.result[l] = r;
    }));
}
template <typename T> static inline T __shfl_sync(unsigned, T v, int src, int = 32) {
    uint64_t u = 0; static_assert(sizeof(T) <= 8, "payload"); memcpy(&u, &v, sizeof(T));
    const int me = cuemu::cur()->lane;
    // two-phase: everybody deposits, the result for lane l is the slot of ITS source lane - sources differ per lane, so the functor copies all slots
    // and each lane picks after the collective
    static thread_local uint64_t snap[cuemu::MAX_THREADS / cuemu::WARP][cuemu::WARP];
    const int wi = cuemu::cur()->warp;
    cuemu::warp_collective(u, [wi](cuemu::WarpState& w) { for (int l = 0; l < cuemu::WARP; ++l) { snap[wi][l] = w.slot[l]; w.result[l] = 0; } });
    (void)me;
    T out; const uint64_t r = snap[wi][src & (cuemu::WARP - 1)]; memcpy(&out, &r, sizeof(T)); return out;
}
static inline void __syncwarp(unsigned = 0xffffffffu) { cuemu::warp_collective(0, [](cuemu::WarpState& w) { for (int l = 0; l < cuemu::WARP; ++l) w.result[l] = 0; }); }
// x >> n with CUDA's semantics (shr.b32 clamps the count: n >= 32 gives 0); C++ leaves that undefined and x86 masks the count
static inline unsigned cuemu_shr_u32(unsigned x, unsigned n) { return n >= 32u ? 0u : x >> n; }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
// __fns(mask, base, offset): position of the |offset|-th set bit of mask counting from bit `base` upwards (offset > 0; the kernels use base 0)
static inline unsigned __fns(unsigned mask, unsigned base, int offset) {
    if (offset > 0) { for (unsigned p = base; p < 32; ++p) if ((mask >> p) & 1) { if (--offset == 0) return p; } return 0xffffffffu; }
    if (offset < 0) { for (int p = int(base); p >= 0; --p) if ((mask >> p) & 1) { if (++offset == 0) return unsigned(p); } return 0xffffffffu; }
    return ((mask >> base) & 1) ? base : 0xffffffffu;
}
static inline float __saturatef(float x) { return x != x ? 0.f : fminf(fmaxf(x, 0.f), 1.f); }
static inline float __uint2float_rn(unsigned v) { return float(v); }
static inline int __float2int_ru(float x) { return int(ceilf(x)); }
static inline int __float2int_rd(float x) { return int(floorf(x)); }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
