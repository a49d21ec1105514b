// Microbenchmark (round 6): what would "entry-pair" evaluation buy the rasterizer's inner loop on gfx950?
//   hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 tools/pk_pair_rate.hip -o /tmp/pk_pair_rate && /tmp/pk_pair_rate
// The geometry of raster.hip's ray_eval (q = A d, t = (g.q)/|q|^2, w = g - t q, alpha = exp2(lo - |w|^2)) for TWO consecutive list entries against one ray per lane:
//   scalar   : two evaluations with plain v_fma_f32 on SGPR record operands (what raster_fwd does today; 64-byte records through s_load_dwordx16)
//   pk_sgpr  : one evaluation of a PAIR record (128 bytes, the two records' fields interleaved: dword 2k = e0[k], 2k + 1 = e1[k]) with v_pk_fma_f32 / v_pk_mul_f32 whose
//              record operand is an SGPR pair
//   pk_vgpr  : the same, the pair record read from LDS into VGPR pairs (broadcast ds_read_b128: every lane the same address) - the form a wavefront would use if
//              it interleaved plain 64-byte records through LDS itself
// Each kernel loops over `iters` record pairs from a small (scalar-cache / L2 resident) table, one wavefront per workgroup as raster_fwd, enough workgroups for 7 wavefronts
// per SIMD. Output: ns per (wavefront, entry) and the ratio to `scalar`. The kill criterion of the round-5 review is decided by this number first:
// the geometry is 27 of the forward's ~37 VALU instructions per evaluation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v2 __attribute__((ext_vector_type(2)));

struct Rec { float4 r0, r1, r2, r3; };       // as lfs::GaussRec: rows (A_r0, A_r1, A_r2, g_r), then (lo, r, g, b)
struct PairRec { v2 f[16]; };                // f[k] = (e0[k], e1[k])

__device__ __forceinline__ float fma3(float ax, float bx, float ay, float by, float az, float bz) {
    return __builtin_fmaf(az, bz, __builtin_fmaf(ay, by, ax * bx));
}
__device__ __forceinline__ float eval_scalar(const Rec& rec, float dx, float dy, float dz) {
    const float qx = fma3(rec.r0.x, dx, rec.r0.y, dy, rec.r0.z, dz);
    const float qy = fma3(rec.r1.x, dx, rec.r1.y, dy, rec.r1.z, dz);
    const float qz = fma3(rec.r2.x, dx, rec.r2.y, dy, rec.r2.z, dz);
    const float l = fma3(qx, qx, qy, qy, qz, qz);
    const float rl = fminf(__builtin_amdgcn_rcpf(l), 3.402823466e38f);
    const float t = fma3(rec.r0.w, qx, rec.r1.w, qy, rec.r2.w, qz) * rl;
    const float wx = __builtin_fmaf(-t, qx, rec.r0.w), wy = __builtin_fmaf(-t, qy, rec.r1.w), wz = __builtin_fmaf(-t, qz, rec.r2.w);
    return __builtin_amdgcn_exp2f(__builtin_fmaf(-wz, wz, __builtin_fmaf(-wy, wy, __builtin_fmaf(-wx, wx, rec.r3.x))));
}
__device__ __forceinline__ v2 pfma(v2 a, v2 b, v2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2 eval_pair(const v2 (&f)[16], v2 dx, v2 dy, v2 dz) {
    // field order of Rec: r0 = f[0..3] (A00 A01 A02 g0), r1 = f[4..7], r2 = f[8..11], r3 = f[12..15]
    const v2 qx = pfma(f[2], dz, pfma(f[1], dy, f[0] * dx));
    const v2 qy = pfma(f[6], dz, pfma(f[5], dy, f[4] * dx));
    const v2 qz = pfma(f[10], dz, pfma(f[9], dy, f[8] * dx));
    const v2 l = pfma(qz, qz, pfma(qy, qy, qx * qx));
    const v2 rl{fminf(__builtin_amdgcn_rcpf(l.x), 3.402823466e38f), fminf(__builtin_amdgcn_rcpf(l.y), 3.402823466e38f)};
    const v2 t = pfma(f[11], qz, pfma(f[7], qy, f[3] * qx)) * rl;
    const v2 wx = pfma(-t, qx, f[3]), wy = pfma(-t, qy, f[7]), wz = pfma(-t, qz, f[11]);
    const v2 ex = pfma(-wz, wz, pfma(-wy, wy, pfma(-wx, wx, f[12])));
    return v2{__builtin_amdgcn_exp2f(ex.x), __builtin_amdgcn_exp2f(ex.y)};
}

// the compositing tail of raster_fwd for one entry (kept identical in all three kernels: the part that stays serial)
__device__ __forceinline__ void composite(float araw, float cr, float cg, float cb, float& T, float& thr, float& p0, float& p1, float& p2) {
    const float alpha = fminf(0.999f, araw);
    const bool pass = !(alpha < thr);
    if (__ballot(pass) == 0ull) return;
    const float nT = T * (1.f - alpha);
    const bool fin = pass && nT <= 1e-4f;
    if (pass && !fin) {
        const float vis = alpha * T;
        p0 = __builtin_fmaf(cr, vis, p0); p1 = __builtin_fmaf(cg, vis, p1); p2 = __builtin_fmaf(cb, vis, p2);
        T = nT;
    }
    thr = fin ? __builtin_inff() : thr;
}

constexpr int TABLE = 1024; // records (64 KB) / pair records (128 KB)

template <bool TAIL>
__global__ void __launch_bounds__(64) k_scalar(const Rec* __restrict__ recs, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x;
    const float dx = (float(lane & 7) - 3.5f) * 0.01f, dy = (float(lane >> 3) - 3.5f) * 0.01f, dz = 1.f + float(iters) * 1e-12f;
    float T = 1.f, thr = 1.f / 255.f, p0 = 0.f, p1 = 0.f, p2 = 0.f, acc = 0.f;
    uint32_t k = blockIdx.x * 37u;
    for (int it = 0; it < iters; ++it) {
        const Rec a = recs[k & (TABLE - 1)], b = recs[(k + 1) & (TABLE - 1)];
        k += 2;
        const float a0 = eval_scalar(a, dx, dy, dz);
        if (TAIL) composite(a0, a.r3.y, a.r3.z, a.r3.w, T, thr, p0, p1, p2); else acc += a0;
        const float a1 = eval_scalar(b, dx, dy, dz);
        if (TAIL) composite(a1, b.r3.y, b.r3.z, b.r3.w, T, thr, p0, p1, p2); else acc += a1;
    }
    out[blockIdx.x * 64 + lane] = acc + p0 + p1 + p2 + T;
}

template <bool TAIL>
__global__ void __launch_bounds__(64) k_pk_sgpr(const PairRec* __restrict__ recs, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x;
    const float dxs = (float(lane & 7) - 3.5f) * 0.01f, dys = (float(lane >> 3) - 3.5f) * 0.01f;
    const float dzs = 1.f + float(iters) * 1e-12f;
    const v2 dx{dxs, dxs}, dy{dys, dys}, dz{dzs, dzs};
    float T = 1.f, thr = 1.f / 255.f, p0 = 0.f, p1 = 0.f, p2 = 0.f, acc = 0.f;
    uint32_t k = blockIdx.x * 37u;
    for (int it = 0; it < iters; ++it) {
        const PairRec& pr = recs[(k >> 1) & (TABLE / 2 - 1)];
        k += 2;
        v2 f[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) f[i] = pr.f[i];
        const v2 a = eval_pair(f, dx, dy, dz);
        if (TAIL) {
            composite(a.x, f[13].x, f[14].x, f[15].x, T, thr, p0, p1, p2);
            composite(a.y, f[13].y, f[14].y, f[15].y, T, thr, p0, p1, p2);
        } else acc += a.x + a.y;
    }
    out[blockIdx.x * 64 + lane] = acc + p0 + p1 + p2 + T;
}

template <bool TAIL>
__global__ void __launch_bounds__(64) k_pk_vgpr(const PairRec* __restrict__ recs, float* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) v2 s_tab[16 * 16]; // 16 pair records staged by the wavefront (the real kernel would refill this ring as it walks)
    const int lane = threadIdx.x;
    for (int i = lane; i < 16 * 16; i += 64) s_tab[i] = recs[(blockIdx.x * 16 + (i >> 4)) & (TABLE / 2 - 1)].f[i & 15];
    __builtin_amdgcn_wave_barrier();
    const float dxs = (float(lane & 7) - 3.5f) * 0.01f, dys = (float(lane >> 3) - 3.5f) * 0.01f;
    const float dzs = 1.f + float(iters) * 1e-12f;
    const v2 dx{dxs, dxs}, dy{dys, dys}, dz{dzs, dzs};
    float T = 1.f, thr = 1.f / 255.f, p0 = 0.f, p1 = 0.f, p2 = 0.f, acc = 0.f;
    uint32_t k = blockIdx.x * 37u;
    for (int it = 0; it < iters; ++it) {
        const float4* pr = reinterpret_cast<const float4*>(s_tab + ((k >> 1) & 15) * 16); // wave-uniform address: a broadcast read
        k += 2;
        v2 f[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float4 r = pr[i]; f[2 * i] = v2{r.x, r.y}; f[2 * i + 1] = v2{r.z, r.w}; }
        const v2 a = eval_pair(f, dx, dy, dz);
        if (TAIL) {
            composite(a.x, f[13].x, f[14].x, f[15].x, T, thr, p0, p1, p2);
            composite(a.y, f[13].y, f[14].y, f[15].y, T, thr, p0, p1, p2);
        } else acc += a.x + a.y;
    }
    out[blockIdx.x * 64 + lane] = acc + p0 + p1 + p2 + T;
}

template <class K, class P>
static double run(K kern, const P* recs, float* out, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, recs, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, recs, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.0;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * 4 * 7 * 4; // 7 wavefronts per SIMD resident, 4 rounds of them
    const int iters = 2048;
    std::vector<Rec> h(TABLE);
    srand(1);
    auto rnd = [] { return float(rand()) / float(RAND_MAX) * 2.f - 1.f; };
    for (auto& r : h) {
        r.r0 = make_float4(rnd() * 30.f, rnd() * 30.f, rnd() * 30.f, rnd() * 3.f);
        r.r1 = make_float4(rnd() * 30.f, rnd() * 30.f, rnd() * 30.f, rnd() * 3.f);
        r.r2 = make_float4(rnd() * 30.f, rnd() * 30.f, rnd() * 30.f, rnd() * 3.f - 40.f);
        r.r3 = make_float4(-0.5f + rnd(), rnd(), rnd(), rnd());
    }
    std::vector<PairRec> hp(TABLE / 2);
    for (int i = 0; i < TABLE / 2; ++i) {
        const float* a = reinterpret_cast<const float*>(&h[2 * i]);
        const float* b = reinterpret_cast<const float*>(&h[2 * i + 1]);
        for (int k = 0; k < 16; ++k) hp[i].f[k] = v2{a[k], b[k]};
    }
    Rec* d_r; PairRec* d_p; float* out;
    hipMalloc(&d_r, sizeof(Rec) * TABLE); hipMalloc(&d_p, sizeof(PairRec) * TABLE / 2); hipMalloc(&out, sizeof(float) * size_t(blocks) * 64);
    hipMemcpy(d_r, h.data(), sizeof(Rec) * TABLE, hipMemcpyHostToDevice);
    hipMemcpy(d_p, hp.data(), sizeof(PairRec) * TABLE / 2, hipMemcpyHostToDevice);
    const double entries = double(blocks) * iters * 2.0;
    const double simds = cus * 4.0;
    auto report = [&](const char* name, double ms, double base) {
        // SIMD-time per (wavefront, entry): ms * simds / entries
        const double ns = ms * 1e6 * simds / entries;
        printf("  \"%s\": {\"ms\": %.4f, \"simd_ns_per_wave_entry\": %.2f, \"vs_scalar\": %.3f},\n", name, ms, ns, ms / base);
    };
    printf("{\"device\": \"%s\", \"cus\": %d, \"blocks\": %d, \"iters\": %d,\n", p.name, cus, blocks, iters);
    for (int rep = 0; rep < 2; ++rep) {
        const double s0 = run(k_scalar<false>, d_r, out, blocks, iters), s1 = run(k_scalar<true>, d_r, out, blocks, iters);
        printf(" \"geometry_only_rep%d\": {\n", rep);
        report("scalar", s0, s0);
        report("pk_sgpr", run(k_pk_sgpr<false>, d_p, out, blocks, iters), s0);
        report("pk_vgpr_lds", run(k_pk_vgpr<false>, d_p, out, blocks, iters), s0);
        printf("  \"_\": 0},\n \"with_compositing_tail_rep%d\": {\n", rep);
        report("scalar", s1, s1);
        report("pk_sgpr", run(k_pk_sgpr<true>, d_p, out, blocks, iters), s1);
        report("pk_vgpr_lds", run(k_pk_vgpr<true>, d_p, out, blocks, iters), s1);
        printf("  \"_\": 0},\n");
    }
    printf(" \"_\": 0}\n");
    return 0;
}
