// Wavefront emulator: a stand-in for <hip/hip_runtime.h> that lets the rasterizer's HIP sources (csrc/raster.hip and its headers) be compiled
// as HOST C++ (clang, -DLFS_EMULATE) and run on the CPU - test infrastructure only (tests/test_emulated_raster.py), never part of the product.
// A workgroup's threads are ucontext fibers scheduled round-robin by one OS thread; a cross-lane operation (ballot, shuffle, DPP, permlane
// swap, readlane, barrier) parks the fiber until every live lane of its wavefront (workgroup for __syncthreads) has arrived, exchanges the
// operands and lets them all continue: exactly the lock-step semantics the kernels rely on, as long as cross-lane operations sit in
// wave-converged code (which they must on the GPU as well; a wave that diverges around one is reported as a deadlock). Lane-private
// arithmetic is ordinary float code (fmaf = fused, exp2f / 1/x stand in for v_exp_f32 / v_rcp_f32), so results are not bit-equal to the
// GPU's, but two kernels that perform the same per-pixel operations stay bit-equal to each other.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <setjmp.h>
#include <ucontext.h>
#include <vector>

// ---- HIP types ------------------------------------------------------------------------------------------------------
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct uint3_emu { unsigned x, y, z; };
// Vector types carry DWORD alignment here: gfx950 serves a global_load / store_dwordx2 / x4 at any 4-byte-aligned address, and the product relies on it in one place -
// the data-parallel path hands the kernels gradient rows that are views into the flat all-reduce bucket (dist.GradBucket), whose segment offsets are multiples of 4 bytes
// only when the Gaussian count is odd (found by tests/test_emulated_dp.py: with 16-byte float4 the host compiler emitted an aligned store and the row write of
// raster_finish_adam_kernel faulted). A host build must therefore not assume more than the GPU guarantees.
struct float2 { float x, y; };
struct alignas(4) float4 { float x, y, z, w; };
struct alignas(4) int2 { int x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
struct alignas(4) ushort4 { unsigned short x, y, z, w; };
static inline ushort4 make_ushort4(unsigned short x, unsigned short y, unsigned short z, unsigned short w) { return ushort4{x, y, z, w}; }
typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
enum { hipMemcpyDeviceToDevice = 3, hipMemcpyDeviceToHost = 2, hipMemcpyHostToDevice = 1, hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? hipSuccess : 2; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
// events and pinned host memory (csrc/fastgs_prep.hip's count read-back): everything is synchronous here
typedef void* hipEvent_t;
enum { hipHostMallocDefault = 0, hipEventDisableTiming = 2 };
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n); return *p ? hipSuccess : 2; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = malloc(1); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
enum { hipStreamNonBlocking = 1 };
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }   // (everything is synchronous here: an event has always happened)
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__ __restrict
#define __constant__

namespace emu {
constexpr int MAX_THREADS = 1024, STACK_BYTES = 512 * 1024;
struct Wave { uint64_t slot[2][64]; uint64_t live = 0; int nlive = 0, arrived = 0, departed = 0; unsigned gen = 0; };
// Fiber switches: swapcontext() saves and restores the signal mask with a system call on every switch (~1 us; a rasterizer evaluation makes a dozen cross-lane
// operations per lane). Each fiber is therefore only ENTERED through its ucontext; every later switch is _setjmp / _longjmp (no signal mask). Under AddressSanitizer
// (LFS_EMUL_SANITIZE) the plain swapcontext path is kept - ASan follows stack switches through its swapcontext interceptor only.
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define LFS_EMU_UCONTEXT_ONLY 1
#endif
#endif
#ifndef LFS_EMU_UCONTEXT_ONLY
#define LFS_EMU_UCONTEXT_ONLY 0
#endif
struct Fiber { ucontext_t ctx; jmp_buf jb; uint3_emu tid; int flat, lane, wave; bool done, started; char* stack; };
struct Block {
    uint3_emu bid, bdim, gdim;
    Fiber fib[MAX_THREADS];
    Wave waves[MAX_THREADS / 64];
    int nthreads = 0, nlive = 0, bar_arrived = 0, bar_departed = 0; unsigned bar_gen = 0;
    ucontext_t main_ctx; jmp_buf main_jb;
    uint64_t progress = 0;
    void (*body)(void*) = nullptr; void* body_arg = nullptr;
};
inline Block*& blk() { static Block* b = nullptr; return b; }
inline std::vector<uint64_t>& dyn_store() { static std::vector<uint64_t> v; return v; }
inline void* dyn_lds() { return dyn_store().data(); }   // the workgroup's dynamic LDS block (LFS_DYN_LDS in csrc/lfs_math.cuh)
inline Fiber*& cur() { static Fiber* f = nullptr; return f; }
inline void yield() {
    Fiber* f = cur();
#if LFS_EMU_UCONTEXT_ONLY
    swapcontext(&f->ctx, &blk()->main_ctx);
#else
    if (_setjmp(f->jb) == 0) _longjmp(blk()->main_jb, 1);
#endif
}
inline void fiber_entry() {
    Block* b = blk(); Fiber* f = cur();
    b->body(b->body_arg);
    f->done = true;
    Wave& w = b->waves[f->wave];
    w.live &= ~(1ull << f->lane); w.nlive--; b->nlive--; b->progress++;
#if LFS_EMU_UCONTEXT_ONLY
    swapcontext(&f->ctx, &b->main_ctx);
#else
    _longjmp(b->main_jb, 1);
#endif
}
// all live lanes of the wavefront exchange up to two 64-bit payloads; out[k][l] is meaningful where live has bit l
inline uint64_t wave_exchange(uint64_t a, uint64_t b2, uint64_t (&out)[2][64]) {
    Block* b = blk(); Fiber* f = cur(); Wave& w = b->waves[f->wave];
    const unsigned g = w.gen;
    w.slot[0][f->lane] = a; w.slot[1][f->lane] = b2; w.arrived++;
    while (w.arrived < w.nlive) yield();
    const uint64_t live = w.live;
    for (int l = 0; l < 64; ++l) { out[0][l] = (live >> l) & 1 ? w.slot[0][l] : 0; out[1][l] = (live >> l) & 1 ? w.slot[1][l] : 0; }
    b->progress++;
    w.departed++;
    if (w.departed >= w.nlive) { w.arrived = 0; w.departed = 0; w.gen++; }
    else while (w.gen == g) yield();
    return live;
}
inline void block_barrier() {
    Block* b = blk();
    const unsigned g = b->bar_gen;
    b->bar_arrived++;
    while (b->bar_arrived < b->nlive) yield();
    b->progress++;
    b->bar_departed++;
    if (b->bar_departed >= b->nlive) { b->bar_arrived = 0; b->bar_departed = 0; b->bar_gen++; }
    else while (b->bar_gen == g) yield();
}
template <class F> void trampoline(void* p) { (*static_cast<F*>(p))(); }
template <class F>
void launch(dim3 grid, dim3 block, size_t shm, F&& f) {
    static Block* b = nullptr;
    if (dyn_store().size() * 8 < shm + 16) dyn_store().resize((shm + 16) / 8 + 1);
    if (!b) {
        b = new Block();
        for (int i = 0; i < MAX_THREADS; ++i) b->fib[i].stack = (char*)malloc(STACK_BYTES);
    }
    blk() = b;
    const int nt = int(block.x * block.y * block.z);
    if (nt > MAX_THREADS) { fprintf(stderr, "emu: workgroup too large\n"); abort(); }
    b->body = &trampoline<typename std::remove_reference<F>::type>; b->body_arg = (void*)&f;
    b->bdim = {block.x, block.y, block.z}; b->gdim = {grid.x, grid.y, grid.z};
    for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) {
        b->bid = {bx, by, bz}; b->nthreads = nt; b->nlive = nt; b->bar_arrived = b->bar_departed = 0;
        for (int w = 0; w < (nt + 63) / 64; ++w) { Wave& wv = b->waves[w]; wv.live = 0; wv.nlive = 0; wv.arrived = wv.departed = 0; }
        for (int t = 0; t < nt; ++t) {
            Fiber& fb = b->fib[t];
            fb.flat = t; fb.lane = t & 63; fb.wave = t >> 6; fb.done = false; fb.started = false;
            fb.tid = {unsigned(t) % block.x, (unsigned(t) / block.x) % block.y, unsigned(t) / (block.x * block.y)};
            b->waves[fb.wave].live |= 1ull << fb.lane; b->waves[fb.wave].nlive++;
            getcontext(&fb.ctx);
            fb.ctx.uc_stack.ss_sp = fb.stack; fb.ctx.uc_stack.ss_size = STACK_BYTES; fb.ctx.uc_link = &b->main_ctx;
            makecontext(&fb.ctx, (void (*)())fiber_entry, 0);
        }
        int remaining = nt;
        while (remaining > 0) {
            const uint64_t before = b->progress;
            remaining = 0;
            for (int t = 0; t < nt; ++t) {
                Fiber& fb = b->fib[t];
                if (fb.done) continue;
                cur() = &fb;
#if LFS_EMU_UCONTEXT_ONLY
                swapcontext(&b->main_ctx, &fb.ctx);
#else
                if (_setjmp(b->main_jb) == 0) {
                    if (!fb.started) { fb.started = true; setcontext(&fb.ctx); }
                    _longjmp(fb.jb, 1);
                }
#endif
                if (!fb.done) ++remaining;
            }
            if (remaining > 0 && b->progress == before) {
                fprintf(stderr, "emu: deadlock in workgroup (%u,%u,%u): a cross-lane operation was reached by only part of a wavefront\n", bx, by, bz);
                abort();
            }
        }
    }
}
// ---- cross-lane primitives ----------------------------------------------------------------------------------------------
inline uint64_t ballot(bool p) { uint64_t o[2][64]; const uint64_t live = wave_exchange(p ? 1 : 0, 0, o); uint64_t m = 0; for (int l = 0; l < 64; ++l) if (((live >> l) & 1) && o[0][l]) m |= 1ull << l; return m; }
inline uint32_t lane_id() { return uint32_t(cur()->lane); }
inline uint32_t xlane_u32(uint32_t v, int src) { uint64_t o[2][64]; wave_exchange(v, 0, o); return uint32_t(o[0][src & 63]); }
inline uint32_t readfirstlane_u32(uint32_t v) { uint64_t o[2][64]; const uint64_t live = wave_exchange(v, 0, o); return uint32_t(o[0][__builtin_ctzll(live)]); }
inline int dpp_src(int lane, int ctrl) {
    const int row = lane & ~15, l = lane & 15;
    if (ctrl >= 0 && ctrl <= 0xFF) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);       // quad_perm
    if (ctrl >= 0x121 && ctrl <= 0x12F) return row | ((l - (ctrl - 0x120)) & 15);                 // row_ror:n (lane i <- lane i - n)
    if (ctrl == 0x140) return row | (15 - l);                                                      // row_mirror
    if (ctrl == 0x141) return row | (l & 8) | (7 - (l & 7));                                       // row_half_mirror
    if (ctrl >= 0x150 && ctrl <= 0x15F) return row | (ctrl - 0x150);                               // row_newbcast:n
    fprintf(stderr, "emu: DPP control 0x%x is not modelled\n", ctrl); abort();
}
inline uint32_t update_dpp(uint32_t, uint32_t src, int ctrl, int, int, bool) { uint64_t o[2][64]; wave_exchange(src, 0, o); return uint32_t(o[0][dpp_src(cur()->lane, ctrl)]); }
struct u32x2 { uint32_t v[2]; uint32_t operator[](int i) const { return v[i]; } };
// v_permlane32_swap: the upper half of vdst is exchanged with the lower half of src;  v_permlane16_swap: odd rows of vdst with even rows of src
inline u32x2 permlane32_swap(uint32_t vdst, uint32_t src, bool, bool) {
    uint64_t o[2][64]; wave_exchange(vdst, src, o); const int l = cur()->lane;
    return l < 32 ? u32x2{{uint32_t(o[0][l]), uint32_t(o[0][l + 32])}} : u32x2{{uint32_t(o[1][l - 32]), uint32_t(o[1][l])}};
}
inline u32x2 permlane16_swap(uint32_t vdst, uint32_t src, bool, bool) {
    uint64_t o[2][64]; wave_exchange(vdst, src, o); const int l = cur()->lane;
    return ((l >> 4) & 1) == 0 ? u32x2{{uint32_t(o[0][l]), uint32_t(o[0][l + 16])}} : u32x2{{uint32_t(o[1][l - 16]), uint32_t(o[1][l])}};
}
} // namespace emu

#define threadIdx (emu::cur()->tid)
#define blockIdx (emu::blk()->bid)
#define blockDim (emu::blk()->bdim)
#define gridDim (emu::blk()->gdim)
#define hipLaunchKernelGGL(kern, grid, block, shm, stream, ...) emu::launch((grid), (block), size_t(shm), [&]() { kern(__VA_ARGS__); })

static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline float __saturatef(float x) { return fminf(fmaxf(x, 0.f), 1.f); }   // (NaN -> 0, as the clamp modifier)
static inline int __float2int_rd(float x) { return int(floorf(x)); }
static inline int __float2int_ru(float x) { return int(ceilf(x)); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline uint64_t __ballot(bool p) { return emu::ballot(p); }
static inline void __syncthreads() { emu::block_barrier(); }
static inline float __shfl_xor(float v, int m, int = 64) { return __uint_as_float(emu::xlane_u32(__float_as_uint(v), int(emu::lane_id()) ^ m)); }
static inline int __shfl_xor(int v, int m, int = 64) { return int(emu::xlane_u32(uint32_t(v), int(emu::lane_id()) ^ m)); }
static inline uint32_t __shfl_xor(uint32_t v, int m, int = 64) { return emu::xlane_u32(v, int(emu::lane_id()) ^ m); }
// __shfl: every lane reads the lane it names (ds_bpermute)
static inline int __shfl(int v, int src, int = 64) { uint64_t o[2][64]; emu::wave_exchange(uint64_t(uint32_t(v)), 0, o); return int(uint32_t(o[0][src & 63])); }
static inline uint32_t __shfl(uint32_t v, int src, int = 64) { uint64_t o[2][64]; emu::wave_exchange(uint64_t(v), 0, o); return uint32_t(o[0][src & 63]); }
static inline float __shfl(float v, int src, int = 64) { uint64_t o[2][64]; emu::wave_exchange(uint64_t(__float_as_uint(v)), 0, o); return __uint_as_float(uint32_t(o[0][src & 63])); }
static inline unsigned long long __shfl_xor(unsigned long long v, int m, int = 64) { uint64_t o[2][64]; emu::wave_exchange(uint64_t(v), 0, o); return o[0][(int(emu::lane_id()) ^ m) & 63]; }
// On the GPU the lanes of a wavefront execute in lock-step, so LDS written before a wave barrier by one lane is visible to the others after it. Fibers switch only
// at cross-lane operations: the barrier is one here (every live lane arrives before any continues). It must sit in wave-converged code - as on the GPU.
#define __builtin_amdgcn_wave_barrier() ((void)emu::ballot(true))
#define __builtin_amdgcn_fence(...) do { } while (0)
// __shfl_up: lane i reads lane i - d; lanes below d keep their own value
static inline uint64_t emu_shfl_up64(uint64_t v, int d) { uint64_t o[2][64]; emu::wave_exchange(v, 0, o); const int l = int(emu::lane_id()); return l >= d ? o[0][l - d] : v; }
static inline uint64_t __shfl_up(uint64_t v, int d, int = 64) { return emu_shfl_up64(v, d); }
static inline int64_t __shfl_up(int64_t v, int d, int = 64) { return int64_t(emu_shfl_up64(uint64_t(v), d)); }
static inline uint32_t __shfl_up(uint32_t v, int d, int = 64) { return uint32_t(emu_shfl_up64(v, d)); }
static inline int __shfl_up(int v, int d, int = 64) { return int(uint32_t(emu_shfl_up64(uint32_t(v), d))); }
static inline int64_t __shfl_xor(int64_t v, int m, int = 64) { uint64_t o[2][64]; emu::wave_exchange(uint64_t(v), 0, o); return int64_t(o[0][(int(emu::lane_id()) ^ m) & 63]); }
// fibers switch only at cross-lane operations and barriers: a plain read-modify-write is atomic
template <class T> static inline T atomicAdd(T* p, T v) { const T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { const T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { const T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicOr(T* p, T v) { const T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { const T o = *p; *p = v; return o; }
static inline size_t min(size_t a, size_t b) { return a < b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long min(long a, long b) { return a < b ? a : b; }
static inline float unsafeAtomicAdd(float* p, float v) { const float o = *p; *p = o + v; return o; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __expf(x) expf(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_rsqf(x) (1.0f / sqrtf(x))
#define __builtin_amdgcn_logf(x) log2f(x)
#define __builtin_amdgcn_readfirstlane(x) emu::readfirstlane_u32(uint32_t(x))
#define __builtin_amdgcn_readlane(x, l) emu::xlane_u32(uint32_t(x), (l))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) emu::update_dpp((old), (src), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) emu::permlane32_swap((a), (b), (fi), (bc))
#define __builtin_amdgcn_permlane16_swap(a, b, fi, bc) emu::permlane16_swap((a), (b), (fi), (bc))
#define __builtin_amdgcn_mbcnt_lo(mask, x) (uint32_t(x) + uint32_t(__builtin_popcount(uint32_t(mask) & (emu::lane_id() >= 32 ? 0xffffffffu : ((1u << emu::lane_id()) - 1u)))))
#define __builtin_amdgcn_mbcnt_hi(mask, x) (uint32_t(x) + uint32_t(emu::lane_id() > 32 ? __builtin_popcount(uint32_t(mask) & ((1u << (emu::lane_id() - 32)) - 1u)) : 0))
// v_mfma_f32_16x16x4_f32: D[16][16] = A[16][4] B[4][16] + C. Lane l holds A[l % 16][l / 16] and B[l / 16][l % 16]; register r of the accumulator holds
// D[4 * (l / 16) + r][l % 16] (the layout csrc/bilateral_grid.hip's backward relies on). k summed in ascending order with fused multiply-adds.
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
static inline emu_f32x4 emu_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c) {
    uint64_t o[2][64]; emu::wave_exchange(__float_as_uint(a), __float_as_uint(b), o);
    const int l = int(emu::lane_id()), j = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(__uint_as_float(uint32_t(o[0][16 * k + i])), __uint_as_float(uint32_t(o[1][16 * k + j])), acc);
        c[r] = acc;
    }
    return c;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emu_mfma_f32_16x16x4f32((a), (b), (c))
