// Practical HBM ceilings of the box, hand-written (round 6; VERDICT.md round 5, "What's missing" 4 / "Next round" 2): read-only, write-only, copy and triad
// kernels with 16-byte accesses (global_load_dwordx4 / global_store_dwordx4), persistent grid-stride grids of 256 CUs x {2, 4, 8} workgroups of 256 threads
// (= 2 / 4 / 8 wavefronts per SIMD), plain and non-temporal, on buffers far beyond the 256 MiB Infinity Cache. These - not torch.sum / Tensor.copy_ - are what
// the streaming kernels of the step (sh_fwd, sh_bwd_adam, finish_adam, projection_ut) are priced against in DESIGN.md 4.
//   hipcc -O3 --offload-arch=gfx950 tools/hbm_stream.hip -o /tmp/hbm_stream && /tmp/hbm_stream > profiles/r06/hbm_stream_ceiling.json
// Also a 29-stream pass shaped like raster_finish_adam_kernel (reads 17 streams, read-modify-writes 12 of 4 / 12 / 16 bytes per element): what the stream COUNT costs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

template <bool NT> __device__ __forceinline__ float4 ld(const float4* p) {
    if (NT) return float4{__builtin_nontemporal_load(&p->x), __builtin_nontemporal_load(&p->y), __builtin_nontemporal_load(&p->z), __builtin_nontemporal_load(&p->w)};
    return *p;
}
template <bool NT> __device__ __forceinline__ void st(float4* p, float4 v) {
    if (NT) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w); }
    else *p = v;
}

// UNROLL independent 16-byte accesses per lane per trip: what keeps enough bytes in flight per wavefront
template <bool NT, int UNROLL>
__global__ void __launch_bounds__(256) k_read(const float4* __restrict__ a, float* __restrict__ out, size_t n4) {
    float s = 0.f;
    const size_t stride = size_t(gridDim.x) * 256 * UNROLL;
    for (size_t i = size_t(blockIdx.x) * 256 * UNROLL + threadIdx.x; i < n4; i += stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = (i + size_t(u) * 256 < n4) ? ld<NT>(a + i + size_t(u) * 256) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (s == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s; // never true: keeps the loads
}
template <bool NT, int UNROLL>
__global__ void __launch_bounds__(256) k_write(float4* __restrict__ a, size_t n4, float val) {
    const size_t stride = size_t(gridDim.x) * 256 * UNROLL;
    const float4 v{val, val, val, val};
    for (size_t i = size_t(blockIdx.x) * 256 * UNROLL + threadIdx.x; i < n4; i += stride) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) if (i + size_t(u) * 256 < n4) st<NT>(a + i + size_t(u) * 256, v);
    }
}
template <bool NT, int UNROLL>
__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
    const size_t stride = size_t(gridDim.x) * 256 * UNROLL;
    for (size_t i = size_t(blockIdx.x) * 256 * UNROLL + threadIdx.x; i < n4; i += stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = (i + size_t(u) * 256 < n4) ? ld<NT>(a + i + size_t(u) * 256) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) if (i + size_t(u) * 256 < n4) st<NT>(b + i + size_t(u) * 256, v[u]);
    }
}
template <bool NT, int UNROLL>
__global__ void __launch_bounds__(256) k_triad(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, size_t n4, float k) {
    const size_t stride = size_t(gridDim.x) * 256 * UNROLL;
    for (size_t i = size_t(blockIdx.x) * 256 * UNROLL + threadIdx.x; i < n4; i += stride) {
        float4 x[UNROLL], y[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const bool ok = i + size_t(u) * 256 < n4;
            x[u] = ok ? ld<NT>(a + i + size_t(u) * 256) : float4{0.f, 0.f, 0.f, 0.f};
            y[u] = ok ? ld<NT>(b + i + size_t(u) * 256) : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            if (i + size_t(u) * 256 < n4) st<NT>(c + i + size_t(u) * 256, float4{x[u].x + k * y[u].x, x[u].y + k * y[u].y, x[u].z + k * y[u].z, x[u].w + k * y[u].w});
    }
}
// Adam-shaped: p, m, v read + written, g read (7 accesses of 16 B per 4 elements: 28 B per element as K13)
template <int UNROLL>
__global__ void __launch_bounds__(256) k_adam(float4* __restrict__ p, float4* __restrict__ m, float4* __restrict__ v, const float4* __restrict__ g, size_t n4) {
    const size_t stride = size_t(gridDim.x) * 256 * UNROLL;
    for (size_t i = size_t(blockIdx.x) * 256 * UNROLL + threadIdx.x; i < n4; i += stride) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t j = i + size_t(u) * 256;
            if (j >= n4) continue;
            const float4 P = p[j], M = m[j], V = v[j], G = g[j];
            const float4 M2{0.9f * M.x + 0.1f * G.x, 0.9f * M.y + 0.1f * G.y, 0.9f * M.z + 0.1f * G.z, 0.9f * M.w + 0.1f * G.w};
            const float4 V2{0.999f * V.x + 0.001f * G.x * G.x, 0.999f * V.y + 0.001f * G.y * G.y, 0.999f * V.z + 0.001f * G.z * G.z, 0.999f * V.w + 0.001f * G.w * G.w};
            m[j] = M2; v[j] = V2;
            p[j] = float4{P.x - 1e-3f * M2.x * __builtin_amdgcn_rsqf(V2.x + 1e-15f), P.y - 1e-3f * M2.y * __builtin_amdgcn_rsqf(V2.y + 1e-15f),
                          P.z - 1e-3f * M2.z * __builtin_amdgcn_rsqf(V2.z + 1e-15f), P.w - 1e-3f * M2.w * __builtin_amdgcn_rsqf(V2.w + 1e-15f)};
        }
    }
}


// 12-byte (global_load_dwordx3 at a 12-byte lane stride: the [n,3] rows of means / scales / SH coefficients / their moments) and 4-byte accesses
struct alignas(4) V3 { float a[3]; };
template <int UNROLL, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_read3(const V3* __restrict__ a, float* __restrict__ out, size_t n3) {
    float s = 0.f;
    const size_t stride = size_t(gridDim.x) * BLOCK * UNROLL;
    for (size_t i = size_t(blockIdx.x) * BLOCK * UNROLL + threadIdx.x; i < n3; i += stride) {
        V3 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = (i + size_t(u) * BLOCK < n3) ? a[i + size_t(u) * BLOCK] : V3{{0.f, 0.f, 0.f}};
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) s += v[u].a[0] + v[u].a[1] + v[u].a[2];
    }
    if (s == 123.456f) out[blockIdx.x * BLOCK + threadIdx.x] = s;
}
template <int UNROLL, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_read1(const float* __restrict__ a, float* __restrict__ out, size_t n) {
    float s = 0.f;
    const size_t stride = size_t(gridDim.x) * BLOCK * UNROLL;
    for (size_t i = size_t(blockIdx.x) * BLOCK * UNROLL + threadIdx.x; i < n; i += stride) {
        float v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = (i + size_t(u) * BLOCK < n) ? a[i + size_t(u) * BLOCK] : 0.f;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) s += v[u];
    }
    if (s == 123.456f) out[blockIdx.x * BLOCK + threadIdx.x] = s;
}
template <int UNROLL, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_read4b(const float4* __restrict__ a, float* __restrict__ out, size_t n4) { // 16-byte loads, BLOCK threads per workgroup
    float s = 0.f;
    const size_t stride = size_t(gridDim.x) * BLOCK * UNROLL;
    for (size_t i = size_t(blockIdx.x) * BLOCK * UNROLL + threadIdx.x; i < n4; i += stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = (i + size_t(u) * BLOCK < n4) ? a[i + size_t(u) * BLOCK] : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (s == 123.456f) out[blockIdx.x * BLOCK + threadIdx.x] = s;
}
// Adam on [n,3] rows as 12-byte accesses (the form the step's kernels use for means / scales / SH rows)
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_adam3(V3* __restrict__ p, V3* __restrict__ m, V3* __restrict__ v, const V3* __restrict__ g, size_t n3) {
    const size_t i = size_t(blockIdx.x) * BLOCK + threadIdx.x;
    if (i >= n3) return;
    V3 P = p[i], M = m[i], V = v[i]; const V3 G = g[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        M.a[k] = 0.9f * M.a[k] + 0.1f * G.a[k]; V.a[k] = 0.999f * V.a[k] + 0.001f * G.a[k] * G.a[k];
        P.a[k] -= 1e-3f * M.a[k] * __builtin_amdgcn_rsqf(V.a[k] + 1e-15f);
    }
    m[i] = M; v[i] = V; p[i] = P;
}

// The coefficient stream of sh_fwd (csrc/sh.hip): per Gaussian 15 rows of 12 bytes (180 B), 64 Gaussians per wavefront = 11 520 contiguous bytes.
//   rows : the kernel's own mapping - lane = (Gaussian g = lane / 16 of 4 per trip, row k = lane % 16, k = 0 idle), 16 trips of one 12-byte load per lane
//   flat : the same 11 520 bytes as 720 float4, lane + 64 i: 12 trips of one 16-byte load per lane (the last one a quarter full)
// Both add everything up and write 12 bytes per Gaussian, so the traffic is sh_fwd's. WAVES wavefronts per workgroup.
template <int WAVES, bool FLAT>
__global__ void __launch_bounds__(64 * WAVES) k_shn(const float* __restrict__ shN, float* __restrict__ colors, size_t n_gauss) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t g0 = (size_t(blockIdx.x) * WAVES + wave) * 64;
    if (g0 >= n_gauss) return;
    const float* blk = shN + g0 * 45;
    float s = 0.f;
    if (FLAT) {
        float4 v[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) { const uint32_t idx = lane + 64 * i; v[i] = idx < 720 ? reinterpret_cast<const float4*>(blk)[idx] : float4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int i = 0; i < 12; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
    } else {
        V3 v[16];
        const uint32_t k = lane & 15, gl = lane >> 4;
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = k ? *reinterpret_cast<const V3*>(blk + (size_t(i * 4 + gl) * 15 + (k - 1)) * 3) : V3{{0.f, 0.f, 0.f}};
#pragma unroll
        for (int i = 0; i < 16; ++i) s += v[i].a[0] + v[i].a[1] + v[i].a[2];
    }
    V3 o; o.a[0] = s; o.a[1] = s + 1.f; o.a[2] = s + 2.f;
    reinterpret_cast<V3*>(colors)[g0 + lane] = o;
}

// NS equally long read streams walked in lock-step (thread i takes element i of every stream: the access pattern of a per-Gaussian pass over many [n, c] arrays
// such as raster_finish_adam_kernel's 29), each 16 bytes per thread. `skew` bytes are added to stream j's base (j x skew): 0 = every base 2 MiB-aligned, as the
// caching allocator hands large blocks out.
struct StreamPtrs { const float4* p[32]; };
template <int NS>
__global__ void __launch_bounds__(256) k_multi_read(const StreamPtrs sp, float* __restrict__ out, size_t n4) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 v[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) v[j] = sp.p[j][i];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NS; ++j) s += v[j].x + v[j].y + v[j].z + v[j].w;
    if (s == 123.456f) out[i & 1023] = s;
}
struct StreamPtrsW { float4* p[32]; };
template <int NS>   // read-modify-write of every stream (an optimizer-shaped pass)
__global__ void __launch_bounds__(256) k_multi_rmw(const StreamPtrsW sp, size_t n4) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 v[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) v[j] = sp.p[j][i];
#pragma unroll
    for (int j = 0; j < NS; ++j) { v[j].x += 1.f; v[j].y *= 0.5f; sp.p[j][i] = v[j]; }
}

// Does a one-pass kernel (load everything -> compute -> store everything, one element per thread) overlap its arithmetic with its memory traffic? The same 16-stream
// read-modify-write with WORK dependent-free FMA rounds of 16 instructions per thread in between (WORK = 50: ~800 VALU instructions per wavefront, what
// raster_finish_adam_kernel executes). If the time is max(memory, compute) the waves of a SIMD cover for each other; if it is the SUM they move in lock step.
template <int NS, int WORK, int WAVES_HINT>
__global__ void __launch_bounds__(256) k_multi_rmw_work(const StreamPtrsW sp, size_t n4, float c) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 v[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) v[j] = sp.p[j][i];
    float a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = v[k % NS].x + float(k);
    for (int w = 0; w < WORK; ++w) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = __builtin_fmaf(a[k], c, 0.5f);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += a[k];
#pragma unroll
    for (int j = 0; j < NS; ++j) { v[j].x += s; v[j].y *= 0.5f; sp.p[j][i] = v[j]; }
}

struct Timer {
    hipEvent_t e0, e1;
    Timer() { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); }
    template <class F> double ms(F&& f, int reps) {
        f(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < reps; ++r) f();
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float t = 0.f; (void)hipEventElapsedTime(&t, e0, e1);
        return t / reps;
    }
};

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const size_t bytes = size_t(argc > 1 ? atoll(argv[1]) : 2048) << 20; // per buffer, MiB
    const size_t n4 = bytes / 16;
    float4 *a, *b, *c, *d; float* out;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&c, bytes) != hipSuccess || hipMalloc(&d, bytes) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
    (void)hipMalloc(&out, sizeof(float) * 256 * 8192);
    (void)hipMemset(a, 0, bytes); (void)hipMemset(b, 0, bytes); (void)hipMemset(c, 0, bytes); (void)hipMemset(d, 0, bytes);
    Timer T;
    const int reps = 10;
    printf("{\"device\": \"%s\", \"cus\": %d, \"buffer_mib\": %zu, \"unit\": \"TB/s (bytes moved / time; copy and triad count reads + writes)\",\n", prop.name, cus, bytes >> 20);
    double best_read = 0, best_write = 0, best_copy = 0, best_triad = 0, best_adam = 0;
    std::string best_read_cfg, best_write_cfg, best_copy_cfg, best_triad_cfg;
    printf(" \"sweep\": [\n");
    bool first = true;
    auto emit = [&](const char* kind, bool nt, int wpc, int unroll, double ms, double moved, double& best, std::string* cfg) {
        const double tbs = moved / (ms * 1e-3) * 1e-12;
        printf("%s  {\"kernel\": \"%s\", \"nt\": %s, \"workgroups_per_cu\": %d, \"unroll\": %d, \"ms\": %.4f, \"tb_s\": %.3f}", first ? "" : ",\n", kind, nt ? "true" : "false", wpc, unroll, ms, tbs);
        first = false;
        if (tbs > best) { best = tbs; if (cfg) { char buf[96]; snprintf(buf, sizeof buf, "nt=%d wg/cu=%d unroll=%d", int(nt), wpc, unroll); *cfg = buf; } }
    };
    for (int wpc : {2, 4, 8}) {
        const int grid = cus * wpc;
#define SWEEP(NT, U) \
        emit("read", NT, wpc, U, T.ms([&] { hipLaunchKernelGGL((k_read<NT, U>), dim3(grid), dim3(256), 0, 0, a, out, n4); }, reps), double(bytes), best_read, &best_read_cfg); \
        emit("write", NT, wpc, U, T.ms([&] { hipLaunchKernelGGL((k_write<NT, U>), dim3(grid), dim3(256), 0, 0, b, n4, 1.f); }, reps), double(bytes), best_write, &best_write_cfg); \
        emit("copy", NT, wpc, U, T.ms([&] { hipLaunchKernelGGL((k_copy<NT, U>), dim3(grid), dim3(256), 0, 0, a, b, n4); }, reps), 2.0 * bytes, best_copy, &best_copy_cfg); \
        emit("triad", NT, wpc, U, T.ms([&] { hipLaunchKernelGGL((k_triad<NT, U>), dim3(grid), dim3(256), 0, 0, a, b, c, n4, 0.5f); }, reps), 3.0 * bytes, best_triad, &best_triad_cfg);
        SWEEP(false, 1) SWEEP(false, 4) SWEEP(false, 8) SWEEP(true, 4) SWEEP(true, 8)
#undef SWEEP
        emit("adam", false, wpc, 2, T.ms([&] { hipLaunchKernelGGL((k_adam<2>), dim3(grid), dim3(256), 0, 0, a, b, c, d, n4); }, reps), 7.0 * bytes, best_adam, nullptr);
    }
    // one workgroup per 256 x UNROLL elements (no grid-stride loop): the launch shape of most of the step's streaming kernels
    {
        const size_t per = 256 * 4;
        const int grid = int((n4 + per - 1) / per);
        double dummy = 0;
        emit("read_one_tile_per_workgroup", false, 0, 4, T.ms([&] { hipLaunchKernelGGL((k_read<false, 4>), dim3(grid), dim3(256), 0, 0, a, out, n4); }, reps), double(bytes), dummy, nullptr);
        emit("copy_one_tile_per_workgroup", false, 0, 4, T.ms([&] { hipLaunchKernelGGL((k_copy<false, 4>), dim3(grid), dim3(256), 0, 0, a, b, n4); }, reps), 2.0 * bytes, dummy, nullptr);
    }

    // access width and workgroup shape (one tile per workgroup unless a grid-stride count is given): what the [n,3]-row kernels of the step are made of
    {
        double dummy = 0;
        const size_t n3 = bytes / 12, n1 = bytes / 4;
#define ONE_TILE(KERN, BLOCK, U, N) int((N + size_t(BLOCK) * U - 1) / (size_t(BLOCK) * U))
        emit("read_16B_block64_one_tile", false, 0, 4, T.ms([&] { hipLaunchKernelGGL((k_read4b<4, 64>), dim3(ONE_TILE(0, 64, 4, n4)), dim3(64), 0, 0, a, out, n4); }, reps), double(bytes), dummy, nullptr);
        emit("read_16B_block64_one_tile", false, 0, 16, T.ms([&] { hipLaunchKernelGGL((k_read4b<16, 64>), dim3(ONE_TILE(0, 64, 16, n4)), dim3(64), 0, 0, a, out, n4); }, reps), double(bytes), dummy, nullptr);
        emit("read_12B_block256_one_tile", false, 0, 4, T.ms([&] { hipLaunchKernelGGL((k_read3<4, 256>), dim3(ONE_TILE(0, 256, 4, n3)), dim3(256), 0, 0, (const V3*)a, out, n3); }, reps), double(n3 * 12), dummy, nullptr);
        emit("read_12B_block64_one_tile", false, 0, 16, T.ms([&] { hipLaunchKernelGGL((k_read3<16, 64>), dim3(ONE_TILE(0, 64, 16, n3)), dim3(64), 0, 0, (const V3*)a, out, n3); }, reps), double(n3 * 12), dummy, nullptr);
        emit("read_12B_block256_one_tile", false, 0, 1, T.ms([&] { hipLaunchKernelGGL((k_read3<1, 256>), dim3(ONE_TILE(0, 256, 1, n3)), dim3(256), 0, 0, (const V3*)a, out, n3); }, reps), double(n3 * 12), dummy, nullptr);
        emit("read_12B_gridstride", false, 8, 4, T.ms([&] { hipLaunchKernelGGL((k_read3<4, 256>), dim3(cus * 8), dim3(256), 0, 0, (const V3*)a, out, n3); }, reps), double(n3 * 12), dummy, nullptr);
        emit("read_4B_block256_one_tile", false, 0, 4, T.ms([&] { hipLaunchKernelGGL((k_read1<4, 256>), dim3(ONE_TILE(0, 256, 4, n1)), dim3(256), 0, 0, (const float*)a, out, n1); }, reps), double(n1 * 4), dummy, nullptr);
        emit("read_4B_block256_one_tile", false, 0, 1, T.ms([&] { hipLaunchKernelGGL((k_read1<1, 256>), dim3(ONE_TILE(0, 256, 1, n1)), dim3(256), 0, 0, (const float*)a, out, n1); }, reps), double(n1 * 4), dummy, nullptr);
        emit("adam_12B_rows_block256_one_elem", false, 0, 1, T.ms([&] { hipLaunchKernelGGL((k_adam3<256>), dim3(ONE_TILE(0, 256, 1, n3)), dim3(256), 0, 0, (V3*)a, (V3*)b, (V3*)c, (const V3*)d, n3); }, reps), 7.0 * double(n3 * 12), dummy, nullptr);
        emit("adam_16B_one_elem", false, 0, 1, T.ms([&] { hipLaunchKernelGGL((k_adam<1>), dim3(ONE_TILE(0, 256, 1, n4)), dim3(256), 0, 0, a, b, c, d, n4); }, reps), 7.0 * bytes, dummy, nullptr);
#undef ONE_TILE
    }

    {   // sh_fwd's coefficient stream: 180 B read + 12 B written per Gaussian
        double dummy = 0;
        const size_t ng = (bytes / 180 / 256) * 256;
        const double moved = double(ng) * 192.0;
        emit("shN_rows_12B_1wave_per_workgroup", false, 0, 16, T.ms([&] { hipLaunchKernelGGL((k_shn<1, false>), dim3(ng / 64), dim3(64), 0, 0, (const float*)a, (float*)b, ng); }, reps), moved, dummy, nullptr);
        emit("shN_rows_12B_4waves_per_workgroup", false, 0, 16, T.ms([&] { hipLaunchKernelGGL((k_shn<4, false>), dim3(ng / 256), dim3(256), 0, 0, (const float*)a, (float*)b, ng); }, reps), moved, dummy, nullptr);
        emit("shN_flat_16B_1wave_per_workgroup", false, 0, 12, T.ms([&] { hipLaunchKernelGGL((k_shn<1, true>), dim3(ng / 64), dim3(64), 0, 0, (const float*)a, (float*)b, ng); }, reps), moved, dummy, nullptr);
        emit("shN_flat_16B_4waves_per_workgroup", false, 0, 12, T.ms([&] { hipLaunchKernelGGL((k_shn<4, true>), dim3(ng / 256), dim3(256), 0, 0, (const float*)a, (float*)b, ng); }, reps), moved, dummy, nullptr);
    }

    {   // many streams in lock-step: does the COUNT cost bandwidth, and does the relative alignment of the bases matter?
        double dummy = 0;
        const size_t per = size_t(64) << 20;                 // 64 MiB per stream
        const size_t m4 = per / 16;
        char* pool = reinterpret_cast<char*>(a);              // a, b, c, d are separate 2 GiB allocations: carve 8 streams out of each
        char* pools[4] = {reinterpret_cast<char*>(a), reinterpret_cast<char*>(b), reinterpret_cast<char*>(c), reinterpret_cast<char*>(d)};
        (void)pool;
        for (size_t skew : {size_t(0), size_t(4352), size_t(65536 + 256)}) {
            StreamPtrs sp; StreamPtrsW sw;
            for (int j = 0; j < 32; ++j) { char* base = pools[j & 3] + size_t(j >> 2) * (per + (size_t(4) << 20)) + size_t(j) * skew; sp.p[j] = reinterpret_cast<const float4*>(base); sw.p[j] = reinterpret_cast<float4*>(base); }
            const int grid = int((m4 + 255) / 256);
            char name[64];
#define MULTI(NS) \
            snprintf(name, sizeof name, "multi_read_%d_streams_skew%zu", NS, skew); \
            emit(name, false, 0, NS, T.ms([&] { hipLaunchKernelGGL((k_multi_read<NS>), dim3(grid), dim3(256), 0, 0, sp, out, m4); }, reps), double(NS) * per, dummy, nullptr); \
            snprintf(name, sizeof name, "multi_rmw_%d_streams_skew%zu", NS, skew); \
            emit(name, false, 0, NS, T.ms([&] { hipLaunchKernelGGL((k_multi_rmw<NS>), dim3(grid), dim3(256), 0, 0, sw, m4); }, reps), 2.0 * NS * per, dummy, nullptr);
            MULTI(2) MULTI(4) MULTI(8) MULTI(16) MULTI(32)
#undef MULTI
        }
    }

    {   // memory + arithmetic in a one-pass kernel: max or sum?
        double dummy = 0;
        const size_t per = size_t(64) << 20, m4 = per / 16;
        char* pools[4] = {reinterpret_cast<char*>(a), reinterpret_cast<char*>(b), reinterpret_cast<char*>(c), reinterpret_cast<char*>(d)};
        StreamPtrsW sw;
        for (int j = 0; j < 32; ++j) sw.p[j] = reinterpret_cast<float4*>(pools[j & 3] + size_t(j >> 2) * (per + (size_t(4) << 20)));
        const int grid = int((m4 + 255) / 256);
        emit("rmw16_work0", false, 0, 0, T.ms([&] { hipLaunchKernelGGL((k_multi_rmw_work<16, 0, 0>), dim3(grid), dim3(256), 0, 0, sw, m4, 0.999f); }, reps), 32.0 * per, dummy, nullptr);
        emit("rmw16_work25_400valu", false, 0, 25, T.ms([&] { hipLaunchKernelGGL((k_multi_rmw_work<16, 25, 0>), dim3(grid), dim3(256), 0, 0, sw, m4, 0.999f); }, reps), 32.0 * per, dummy, nullptr);
        emit("rmw16_work50_800valu", false, 0, 50, T.ms([&] { hipLaunchKernelGGL((k_multi_rmw_work<16, 50, 0>), dim3(grid), dim3(256), 0, 0, sw, m4, 0.999f); }, reps), 32.0 * per, dummy, nullptr);
        emit("rmw16_work100_1600valu", false, 0, 100, T.ms([&] { hipLaunchKernelGGL((k_multi_rmw_work<16, 100, 0>), dim3(grid), dim3(256), 0, 0, sw, m4, 0.999f); }, reps), 32.0 * per, dummy, nullptr);
        emit("rmw16_work200_3200valu", false, 0, 200, T.ms([&] { hipLaunchKernelGGL((k_multi_rmw_work<16, 200, 0>), dim3(grid), dim3(256), 0, 0, sw, m4, 0.999f); }, reps), 32.0 * per, dummy, nullptr);
    }
    // runtime paths for comparison
    {
        double dummy = 0;
        emit("hipMemcpyDtoD", false, 0, 0, T.ms([&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, reps), 2.0 * bytes, dummy, nullptr);
        emit("hipMemset", false, 0, 0, T.ms([&] { (void)hipMemsetAsync(b, 0, bytes, 0); }, reps), double(bytes), dummy, nullptr);
    }
    printf("\n ],\n \"best\": {\"read\": {\"tb_s\": %.3f, \"cfg\": \"%s\"}, \"write\": {\"tb_s\": %.3f, \"cfg\": \"%s\"}, \"copy\": {\"tb_s\": %.3f, \"cfg\": \"%s\"}, \"triad\": {\"tb_s\": %.3f, \"cfg\": \"%s\"}, \"adam_7_streams\": {\"tb_s\": %.3f}}}\n",
           best_read, best_read_cfg.c_str(), best_write, best_write_cfg.c_str(), best_copy, best_copy_cfg.c_str(), best_triad, best_triad_cfg.c_str(), best_adam);
    return 0;
}
