// Test infrastructure (tests/test_gpu_raster.py compiles and runs it on the GPU box): two properties of gfx950 instructions that the rasterizer's backward relies on
// since round 6 and that no product-level test can see directly.
//   1. v_rcp_f32(1.0) == 1.0 exactly: an invalid lane of a backward evaluation carries alpha = 0 and multiplies its transmittance by 1 / (1 - 0) (LFS_BWD_ALPHA0).
//      One ulp off would drift T by 6e-8 per evaluation - invisible to every parity bar, wrong all the same.
//   2. v_mul_legacy_f32: 0 * inf = 0, 0 * NaN = 0 (mul_zero, lfs_raster_common.cuh: the reciprocal of a zero length meets a zero numerator without a clamp).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
__global__ void probe(const float* in, float* out) {
    const int i = threadIdx.x;
    out[i] = __builtin_amdgcn_rcpf(in[i]);
    float r;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(in[64 + i]), "v"(in[128 + i]));
    out[64 + i] = r;
}
int main() {
    float h_in[192], h_out[128];
    for (int i = 0; i < 64; ++i) h_in[i] = ldexpf(1.f, i - 32);           // powers of two 2^-32 .. 2^31: the reciprocal must be exact
    const float inf = INFINITY, nan = NAN;
    const float a[8] = {0.f, 0.f, inf, -0.f, 0.f, 3.f, 0.f, 2.5f}, b[8] = {inf, nan, 0.f, inf, 7.f, 0.f, -inf, 4.f};
    for (int i = 0; i < 64; ++i) { h_in[64 + i] = a[i & 7]; h_in[128 + i] = b[i & 7]; }
    float *d_in, *d_out;
    if (hipMalloc(&d_in, sizeof(h_in)) != hipSuccess || hipMalloc(&d_out, sizeof(h_out)) != hipSuccess) { printf("{\"error\": \"hipMalloc\"}\n"); return 1; }
    hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_in, d_out);
    if (hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost) != hipSuccess) { printf("{\"error\": \"kernel\"}\n"); return 1; }
    int rcp_exact = 1;
    for (int i = 0; i < 64; ++i) rcp_exact &= (h_out[i] == ldexpf(1.f, 32 - i));
    uint32_t one_bits; memcpy(&one_bits, &h_out[32], 4);
    int legacy_ok = 1;
    for (int i = 0; i < 7; ++i) legacy_ok &= (h_out[64 + i] == 0.f);
    legacy_ok &= (h_out[64 + 7] == 10.f);
    printf("{\"rcp_of_one_bits\": \"0x%08x\", \"rcp_exact_on_powers_of_two\": %d, \"mul_legacy_zero_times_anything_is_zero\": %d}\n", one_bits, rcp_exact, legacy_ok);
    return 0;
}
