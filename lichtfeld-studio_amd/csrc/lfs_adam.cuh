// The per-element Adam update of fast_gs::optimizer::adam_step (fastgs/optimizer/include/adam_kernels.cuh:13-36), shared by
// adam.hip and by the SH backward that applies it to shN in place (sh.hip). Both are compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

namespace lfs {

struct AdamScalars { float lr, beta1, beta2, eps, bc1_rcp, bc2_sqrt_rcp; };

__device__ __forceinline__ void adam_elem(float& p, float& m, float& v, const float g, const AdamScalars& s) {
    const float m1 = s.beta1 * m + (1.0f - s.beta1) * g;
    const float m2 = s.beta2 * v + (1.0f - s.beta2) * g * g;
    const float denom = sqrtf(m2) * s.bc2_sqrt_rcp + s.eps;
    const float step = s.lr * s.bc1_rcp;
    p -= step * m1 / denom;
    m = m1; v = m2;
}

} // namespace lfs
