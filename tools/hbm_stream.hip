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
