// ORACLE / TEST INFRASTRUCTURE ONLY - never part of the product.
//
// A host stand-in for the CUDA execution model, just large enough to run the REFERENCE's own kernels
// (/root/reference/gsplat/*.cu and fastgs/optimizer/include/adam_kernels.cuh, compiled in place as host C++ by `make -C oracle refk`)
// on the CPU and produce golden vectors from them (oracle/make_golden_refk.py -> tests/golden/refk_*.npz).  What it models:
//   * __global__/__device__ qualifiers, threadIdx / blockIdx / blockDim / gridDim, dynamic shared memory (`extern __shared__ int s[]`);
//   * one workgroup at a time, its threads as fibers (a 6-register x86-64 context switch) scheduled round-robin by one OS thread;
//   * __syncthreads(), __syncthreads_count(), cooperative_groups: this_grid().thread_rank(), this_thread_block() (sync, thread_rank, size,
//     group_index, thread_index), tiled_partition<32>, reduce(plus / greater) as the 16-8-4-2-1 xor butterfly, any(), thread_rank();
//     threads that have returned count as arrived, as on the GPU;
//   * gpuAtomicAdd / atomicAdd (plain adds: one OS thread), the fast-math intrinsics as their IEEE counterparts (__expf = expf, rsqrtf =
//     1/sqrtf, __frcp_rn = 1/x, __logf = logf).  Results are therefore the reference's ALGORITHM in IEEE fp32, not bit-equal to an NVIDIA
//     GPU (nvcc --use_fast_math contracts and approximates); the parity tests compare with tolerances.
// Kernels without any block-level or warp-level communication are run as plain function calls per thread (no fibers).
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#if !defined(__x86_64__)
#error "oracle/ref_emul/cuda_emul.h: the fiber switch is written for x86-64 (the build container)"
#endif

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__
#define __constant__ static

struct uint2 { unsigned x, y; };
struct uint3 { unsigned x, y, z; };
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };

namespace cuemu {

constexpr int MAX_THREADS = 1024, WARP = 32, STACK_BYTES = 256 * 1024;

extern "C" void cuemu_switch(void** save_sp, void* load_sp);   // (weak: several emulator translation units may meet in one library)
asm(R"(
.text
.weak cuemu_switch
.type cuemu_switch,@function
cuemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size cuemu_switch,.-cuemu_switch
)");

struct Fiber { void* sp; char* stack; uint3 tid; int flat, lane, warp; bool done; };
struct WarpState { uint64_t slot[WARP]; uint32_t live; int nlive, arrived; unsigned gen; uint64_t result[WARP]; };
struct Block {
    uint3 bid, bdim, gdim;
    Fiber fib[MAX_THREADS];
    WarpState warps[MAX_THREADS / WARP];
    int nthreads = 0, nlive = 0;
    int bar_arrived = 0, bar_count = 0, bar_result = 0; unsigned bar_gen = 0;
    void* main_sp = nullptr;
    uint64_t progress = 0;
    void (*body)(void*) = nullptr; void* body_arg = nullptr;
    bool fibers = false;
};
inline Block& blk() { static Block* b = new Block(); return *b; }
inline Fiber*& cur() { static Fiber* f = nullptr; return f; }
inline void yield() { Fiber* f = cur(); cuemu_switch(&f->sp, blk().main_sp); }

// a thread that leaves the kernel counts as arrived at whatever the others are waiting for
inline void release_waiters_after_exit(Block& b, Fiber* f);
inline void fiber_main() {
    Block& b = blk(); Fiber* f = cur();
    b.body(b.body_arg);
    f->done = true;
    release_waiters_after_exit(b, f);
    b.progress++;
    cuemu_switch(&f->sp, b.main_sp);
    abort();
}

// ---- block barrier (optionally counting a predicate) ---------------------------------------------------------------
inline void barrier_complete(Block& b) { b.bar_result = b.bar_count; b.bar_arrived = 0; b.bar_count = 0; b.bar_gen++; b.progress++; }
inline int block_barrier(int pred) {
    Block& b = blk();
    if (!b.fibers) { fprintf(stderr, "cuemu: __syncthreads in a kernel launched without fibers\n"); abort(); }
    const unsigned g = b.bar_gen;
    b.bar_arrived++; b.bar_count += pred ? 1 : 0;
    if (b.bar_arrived >= b.nlive) barrier_complete(b);
    else while (b.bar_gen == g) yield();
    return b.bar_result;
}
// ---- warp collective: every live lane deposits a 64-bit payload, the last arriver runs `fin` over the slots ------------
template <class Fin> inline uint64_t warp_collective(uint64_t payload, Fin fin) {
    Block& b = blk(); Fiber* f = cur();
    if (!b.fibers) { fprintf(stderr, "cuemu: warp collective in a kernel launched without fibers\n"); abort(); }
    WarpState& w = b.warps[f->warp];
    const unsigned g = w.gen;
    w.slot[f->lane] = payload; w.arrived++;
    if (w.arrived >= w.nlive) { fin(w); w.arrived = 0; w.gen++; b.progress++; }
    else while (w.gen == g) yield();
    return w.result[f->lane];
}
inline void release_waiters_after_exit(Block& b, Fiber* f) {
    WarpState& w = b.warps[f->warp];
    w.live &= ~(1u << f->lane); w.nlive--; b.nlive--;
    // (the kernels of the reference never leave a warp collective half-entered: a lane returns before the first collective or not at all.
    //  If lanes were waiting, the collective could not be finished without its functor, so say so instead of hanging.)
    if (w.nlive > 0 && w.arrived >= w.nlive && w.arrived > 0) { fprintf(stderr, "cuemu: a lane exited while its warp waits in a collective\n"); abort(); }
    if (b.nlive > 0 && b.bar_arrived >= b.nlive && b.bar_arrived > 0) barrier_complete(b);
}

template <class F> void trampoline(void* p) { (*static_cast<F*>(p))(); }

// launch<<<grid, block>>>: `fibers` = the kernel uses barriers / warp collectives
template <class F> void launch(dim3 grid, dim3 block, bool fibers, F&& f) {
    Block& b = blk();
    const int nt = int(block.x * block.y * block.z);
    if (nt > MAX_THREADS) { fprintf(stderr, "cuemu: workgroup too large\n"); abort(); }
    b.body = &trampoline<typename std::remove_reference<F>::type>; b.body_arg = (void*)&f;
    b.bdim = {block.x, block.y, block.z}; b.gdim = {grid.x, grid.y, grid.z};
    b.fibers = fibers;
    static Fiber plain;
    for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) {
        b.bid = {bx, by, bz};
        if (!fibers) {
            cur() = &plain;
            for (int t = 0; t < nt; ++t) {
                plain.tid = {unsigned(t) % block.x, (unsigned(t) / block.x) % block.y, unsigned(t) / (block.x * block.y)};
                plain.flat = t; plain.lane = t % WARP; plain.warp = t / WARP;
                f();
            }
            continue;
        }
        b.nthreads = nt; b.nlive = nt; b.bar_arrived = b.bar_count = 0;
        for (int w = 0; w < (nt + WARP - 1) / WARP; ++w) { WarpState& ws = b.warps[w]; ws.live = 0; ws.nlive = 0; ws.arrived = 0; }
        for (int t = 0; t < nt; ++t) {
            Fiber& fb = b.fib[t];
            if (!fb.stack) fb.stack = (char*)aligned_alloc(64, STACK_BYTES);
            fb.flat = t; fb.lane = t % WARP; fb.warp = t / WARP; fb.done = false;
            fb.tid = {unsigned(t) % block.x, (unsigned(t) / block.x) % block.y, unsigned(t) / (block.x * block.y)};
            b.warps[fb.warp].live |= 1u << fb.lane; b.warps[fb.warp].nlive++;
            // initial frame: six callee-saved registers (zero), then the entry address that `ret` jumps to, then a dummy return address
            void** top = reinterpret_cast<void**>(fb.stack + STACK_BYTES);   // 64-byte aligned
            top[-1] = nullptr;
            top[-2] = reinterpret_cast<void*>(&fiber_main);
            for (int k = 3; k <= 8; ++k) top[-k] = nullptr;
            fb.sp = &top[-8];
        }
        int remaining = nt;
        while (remaining > 0) {
            const uint64_t before = b.progress;
            remaining = 0;
            for (int t = 0; t < nt; ++t) {
                Fiber& fb = b.fib[t];
                if (fb.done) continue;
                cur() = &fb;
                cuemu_switch(&b.main_sp, fb.sp);
                if (!fb.done) ++remaining;
            }
            if (remaining > 0 && b.progress == before) { fprintf(stderr, "cuemu: deadlock in block (%u,%u,%u)\n", bx, by, bz); abort(); }
        }
    }
}

inline float u2f(uint64_t u) { float f; uint32_t v = uint32_t(u); memcpy(&f, &v, 4); return f; }
inline uint64_t f2u(float f) { uint32_t v; memcpy(&v, &f, 4); return v; }

} // namespace cuemu

#define threadIdx (cuemu::cur()->tid)
#define blockIdx (cuemu::blk().bid)
#define blockDim (cuemu::blk().bdim)
#define gridDim (cuemu::blk().gdim)

static inline void __syncthreads() { cuemu::block_barrier(0); }
static inline int __syncthreads_count(int pred) { return cuemu::block_barrier(pred); }

// ---- device math: CUDA spells these without std:: --------------------------------------------------------------------
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float rsqrt(float x) { return 1.0f / sqrtf(x); }
static inline double rsqrt(double x) { return 1.0 / sqrt(x); }
// (glibc declares __expf / __logf / __powf itself: route the CUDA intrinsics of the same names through macros)
static inline float cuemu_expf(float x) { return expf(x); }
static inline float cuemu_logf(float x) { return logf(x); }
static inline float cuemu_powf(float x, float y) { return powf(x, y); }
#define __expf cuemu_expf
#define __logf cuemu_logf
#define __powf cuemu_powf
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline double pow(double a, int b) { return ::pow(a, double(b)); }
static inline float pow(float a, int b) { return powf(a, float(b)); }   // CUDA's math overload set has pow(float, int) -> float (host <cmath> would promote to double)
static inline float __fdividef(float a, float b) { return a / b; }
// float -> unsigned conversion as the GPU does it (cvt.rzi.u32.f32 saturates: negative and NaN -> 0, too large -> 0xffffffff); in C++ it is undefined for those
static inline unsigned cuemu_f2u(float v) { return !(v > 0.f) ? 0u : v >= 4294967296.f ? 0xffffffffu : (unsigned)v; }
static inline unsigned cuemu_f2u_floor(float v) { return cuemu_f2u(floorf(v)); }
static inline unsigned cuemu_f2u_ceil(float v) { return cuemu_f2u(ceilf(v)); }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline double min(double a, double b) { return fmin(a, b); }
static inline double max(double a, double b) { return fmax(a, b); }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
// CUDA's mixed overloads convert the int to unsigned (crt/math_functions.hpp)
static inline unsigned min(unsigned a, int b) { return min(a, unsigned(b)); }
static inline unsigned min(int a, unsigned b) { return min(unsigned(a), b); }
static inline unsigned max(unsigned a, int b) { return max(a, unsigned(b)); }
static inline unsigned max(int a, unsigned b) { return max(unsigned(a), b); }

template <typename T, typename U> static inline T gpuAtomicAdd(T* p, U v) { const T o = *p; *p = o + T(v); return o; }   // (ATen's overload set converts the addend)
template <typename T> static inline T atomicAdd(T* p, T v) { const T o = *p; *p = o + v; return o; }

enum cudaError_t { cudaSuccess = 0 };
