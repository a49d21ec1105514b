// Microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 on gfx950 (does packed f32 double the f32 FMA rate of a SIMD?).
//   hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 tools/pk_rate.hip -o /tmp/pk_rate && /tmp/pk_rate
// Every thread runs ITERS iterations over 16 independent accumulator chains (scalar: 16 v_fma_f32; packed: 16 v_pk_fma_f32 = 32 FMAs).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096, CH = 16;

__global__ void __launch_bounds__(256) k_scalar(float* out, float a, float b) {
    float acc[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) acc[i] = float(threadIdx.x + i);
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CH; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_packed(float* out, float a, float b) {
    v2 acc[CH];
    const v2 av{a, a}, bv{b, b};
#pragma unroll
    for (int i = 0; i < CH; ++i) acc[i] = v2{float(threadIdx.x + i), float(threadIdx.x - i)};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CH; ++i) acc[i] = __builtin_elementwise_fma(acc[i], av, bv);
    }
    v2 s{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < CH; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

template <class K>
static double run(K kern, float* out, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 0.999f, 0.001f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 0.999f, 0.001f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.0;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * 8; // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    float* out;
    hipMalloc(&out, sizeof(float) * size_t(blocks) * 256);
    const double clk = p.clockRate * 1e3; // Hz
    const double ms_s = run(k_scalar, out, blocks), ms_p = run(k_packed, out, blocks);
    const double waves = double(blocks) * 4.0, simds = cus * 4.0;
    const double instr = waves * ITERS * CH; // wave-instructions of the timed loop body
    const double cyc_s = ms_s * 1e-3 * clk * simds / instr, cyc_p = ms_p * 1e-3 * clk * simds / instr;
    const double tf_s = instr * 64 * 2 / (ms_s * 1e-3) * 1e-12, tf_p = instr * 64 * 4 / (ms_p * 1e-3) * 1e-12;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %.0f, \"v_fma_f32\": {\"ms\": %.4f, \"cycles_per_wave_instr\": %.2f, \"tflops\": %.1f}, "
           "\"v_pk_fma_f32\": {\"ms\": %.4f, \"cycles_per_wave_instr\": %.2f, \"tflops\": %.1f}}\n",
           p.name, cus, clk * 1e-6, ms_s, cyc_s, tf_s, ms_p, cyc_p, tf_p);
    hipFree(out);
    return 0;
}
