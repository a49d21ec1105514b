// The per-element Adam update of fast_gs::optimizer::adam_step (fastgs/optimizer/include/adam_kernels.cuh:13-36), shared by
// adam.hip, by the SH backward that applies it to shN / sh0 in place (sh.hip) and by raster_finish_adam_kernel (raster.hip). The arithmetic is
// pinned to un-fused IEEE operations inside the function (the first two files are compiled with -ffp-contract=off anyway; raster.hip is not),
// so every caller produces the bits of fast_gs::optimizer::adam_step.
#pragma once
#include <hip/hip_runtime.h>

namespace lfs {

struct AdamScalars { float lr, beta1, beta2, eps, bc1_rcp, bc2_sqrt_rcp; };

__device__ __forceinline__ void adam_elem(float& p, float& m, float& v, const float g, const AdamScalars& s) {
#pragma clang fp contract(off)
    const float m1 = s.beta1 * m + (1.0f - s.beta1) * g;
    const float m2 = s.beta2 * v + (1.0f - s.beta2) * g * g;
    const float denom = sqrtf(m2) * s.bc2_sqrt_rcp + s.eps;
    const float step = s.lr * s.bc1_rcp;
    p -= step * m1 / denom;
    m = m1; v = m2;
}

} // namespace lfs
