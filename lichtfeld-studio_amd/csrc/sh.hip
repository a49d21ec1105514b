// K2 / K9 — spherical harmonics colour and its vjp (replace
// gsplat::spherical_harmonics_fwd / _bwd; reference: gsplat/SphericalHarmonicsCUDA.cu
// :20-110 forward polynomial, :112-371 vjp, :373-481 kernels; host
// SphericalHarmonics.cpp:15-76).
//
// HBM-bound: 12*K B of coefficients per Gaussian dominate. The reference maps one
// thread to one (Gaussian, channel) so coefficient loads are 12-B strided across
// lanes; here lane k of a group of LPG lanes owns basis k of a Gaussian, so a wave's
// coefficient access is one fully coalesced 64 x 12 B block (a 4-lanes x 48-B layout was
// measured 2.5x slower: partial-line stores), and the basis polynomial is evaluated once
// per Gaussian in a lane-per-Gaussian phase and handed over through LDS (see the kernels).
#include "lfs_math.cuh"
#include "lfs_sh.cuh"
#include <algorithm>
#include <cstdlib>
#include "lfs_prof.h"
#include "lfs_adam.cuh"
#include "lfs_step_internal.h"
#include "../../include/lfs_gsplat.h"

namespace lfs {

// Where the operands of one call live. MODEL = false: the op of Ops.h (dirs, coeffs [n,K,3], bool masks).
// MODEL = true (fused L2 extension, not in Ops.h): what rasterizer.cpp:256-263 builds around the op with libtorch -
// dirs = means - campos, masks = all(radii > 0), coeffs = cat(sh0, shN), colors = clamp_min(sh + 0.5, 0) - folded
// in, so none of those tensors is materialised (the cat alone is a 192 MB copy per step at 1M Gaussians, and
// again in the backward).
struct ShArgs {
    uint32_t n, K; int degree;
    const float* dirs; const float* coeffs; const uint8_t* masks;                      // op
    const float* means; const float* viewmat; const float* sh0; const float* shN; const int32_t* radii; // model
    const float* colors;                                                                // model bwd: clamped forward output
    // strided / alternative operands of the model variant (the fastgs rasterizer keeps colours inside its 64-B records):
    const float* campos;      // camera centre [3] instead of a view matrix
    const uint32_t* mask_u32; // visibility = mask_u32[g] != 0 instead of radii
    uint32_t cs, vs;          // element stride of colors / v_colors rows (0 = 3)
    uint32_t ds;              // model bwd: element stride of the v_dirs rows (0 = 3)
    bool dirs_store;          // model bwd: v_dirs rows are WRITTEN (0 for invisible Gaussians) instead of added to
    const int32_t* abort_flag; // (nullable) speculative training step: != 0 on the device -> the kernel must not touch the parameters (lfs_step_internal.h)
    float* rec_rgb; uint32_t rec_stride; // model fwd (pipelined training step): the colour is ALSO written to rec_rgb + rec_stride * g (the rgb slots of the rasterizer's 64-byte records)
};
template <bool MODEL> LFS_DI bool sh_on(const ShArgs& a, uint32_t g) {
    if (MODEL) return a.mask_u32 ? a.mask_u32[g] != 0u : a.radii == nullptr ? true : (a.radii[2 * g] > 0 && a.radii[2 * g + 1] > 0); // (no visibility given: every Gaussian)
    return a.masks == nullptr || a.masks[g] != 0;
}
template <bool MODEL> LFS_DI f3 sh_dir(const ShArgs& a, uint32_t g) {
    if (MODEL) { const f3 cp = a.campos ? f3{a.campos[0], a.campos[1], a.campos[2]} : campos_of(a.viewmat); const f3 m = ld3(a.means, g); return {m.x - cp.x, m.y - cp.y, m.z - cp.z}; }
    return ld3(a.dirs, g);
}
template <bool MODEL, class T> LFS_DI T* sh_coef(T* coeffs, T* sh0, T* shN, uint32_t K, uint32_t g, int k) {
    if (MODEL) return k == 0 ? sh0 + size_t(g) * 3 : shN + (size_t(g) * (K - 1) + (k - 1)) * 3;
    return coeffs + (size_t(g) * K + k) * 3;
}

// One wavefront per workgroup handles 64 Gaussians in three phases (no cross-wave sync needed):
//   1. lane = Gaussian : evaluate the basis polynomial ONCE per Gaussian, park b[0..LPG) in LDS (row stride LPG+1:
//                        conflict-free for both access directions);
//   2. lane = (Gaussian, basis) for 64/LPG Gaussians per iteration: coefficient rows are read / written as fully
//                        coalesced 64 x 12 B blocks; fwd: 16-lane butterfly sum; bwd: v_coeffs = b_k v, and
//                        s_k = coeff_k . v overwrites b_k in LDS;
//   3. (bwd) lane = Gaussian : dL/d(dir) = sum_k s_k grad b_k, evaluated once per Gaussian.
// The previous layout evaluated the polynomial in every one of the LPG lanes of a Gaussian and was VALU-bound
// (rocprof: 8.1e7 VALU instructions = 0.13 ms of the 0.20 ms backward at 1M Gaussians, K = 16).
// LFS_SH_FWD_UNCOND (round 6): the coefficient rows are requested for EVERY Gaussian of the wavefront, together with the visibility word and the direction - one
// memory round trip per wavefront instead of two in series (visibility -> ballot -> rows); the rows of an invisible Gaussian are read and dropped. Pays when most
// Gaussians are visible (SYN-B: 95 %); a view that sees a small part of a scene reads (1 - visible fraction) x 180 B per Gaussian more than it needs.
#ifndef LFS_SH_FWD_UNCOND
#define LFS_SH_FWD_UNCOND 0
#endif
#ifndef LFS_SH_FWD_SPLIT
#define LFS_SH_FWD_SPLIT 1
#endif
template <int LPG, bool MODEL>
LFS_DI void sh_fwd_block(const ShArgs& a, float* __restrict__ colors, const uint32_t g0 /* first of the workgroup's 64 Gaussians */) {
    __shared__ float lds[64 * (LPG + 1)];
    __shared__ float lds_dc[MODEL ? 64 * 3 : 1];
    const uint32_t lane = threadIdx.x;
    const int degree = a.degree;
    const int Kd = (degree + 1) * (degree + 1);
    // Memory round trips of a wavefront, in series: (1) the visibility word and the direction of its 64 Gaussians, issued together; (2) the coefficient rows of
    // the visible ones - issued as soon as the visibility BALLOT is known, before the basis polynomial is evaluated, so the direction's latency and the
    // polynomial hide under them. (Until round 3 the chain was radii.x -> radii.y -> means -> basis -> LDS -> coefficients: four round trips, 0.073 ms for
    // 224 MB at 1M Gaussians = 3.1 TB/s.)
    const uint32_t gmine = g0 + lane;
    bool on = false;
    f3 d{0.f, 0.f, 1.f};
    if (gmine < a.n) {
        d = sh_dir<MODEL>(a, gmine);   // (read for masked-out rows too: 12 B, no dependent wait)
        if (MODEL && a.mask_u32 == nullptr && a.radii != nullptr) { const int2 r = *reinterpret_cast<const int2*>(a.radii + 2 * size_t(gmine)); on = r.x > 0 && r.y > 0; }
        else on = sh_on<MODEL>(a, gmine);
    }
    const unsigned long long vis = __ballot(on);
    constexpr int GPI = 64 / LPG; // Gaussians per iteration
    const int k = lane % LPG;
    // Coefficient loads: masked-out Gaussians get colour 0 and their coefficients are not fetched. LFS_SH_FWD_SPLIT (round 5): the LPG row loads of a lane are issued
    // in TWO halves - the first before the basis polynomial is evaluated (it hides the direction's latency and the polynomial), the second once the basis sits in LDS and
    // its ~30 temporaries are dead - so that at most LPG/2 x 3 prefetch registers are live next to the polynomial: 100 -> <= 64 VGPRs = 8 instead of 5 wavefronts per SIMD
    // (round 4 measured the streaming kernels that run 8 at 5.1 - 5.8 TB/s and this one at 3.6). 0 = all LPG loads up front (rounds 3 - 4).
    constexpr int HALF = (LFS_SH_FWD_SPLIT && LPG >= 8) ? LPG / 2 : LPG;
    float c0[LPG], c1[LPG], c2[LPG];
    // Addresses: one wave-uniform 64-bit row base per iteration (SGPRs) + ONE 32-bit per-lane element offset shared by all iterations (global_load with an SGPR base),
    // instead of a 64-bit VGPR address pair per iteration (32 VGPRs of the 98 the model form needed). The model form's k == 0 lanes take their row from sh0 - another
    // allocation, so another base: the lane = Gaussian phase loads sh0 (one coalesced 768-byte block per wavefront) and parks it in LDS next to the basis.
    const uint32_t KK = MODEL ? a.K - 1 : a.K;                         // rows of the coefficient tensor the lanes k >= (MODEL ? 1 : 0) walk
    const uint32_t lane_el = ((lane / LPG) * KK + (MODEL ? uint32_t(k) - 1u : uint32_t(k))) * 3u;
    const float* const walk = MODEL ? a.shN : a.coeffs;
    auto fetch = [&](const int it) {
        const uint32_t gl = it * GPI + lane / LPG;
        c0[it] = c1[it] = c2[it] = 0.f;
        if (k < Kd && (!MODEL || k >= 1) && (LFS_SH_FWD_UNCOND || ((vis >> gl) & 1ull))) {
            const float* row = walk + size_t(g0 + uint32_t(it) * GPI) * KK * 3u;   // (uniform)
            const V3f t3 = *reinterpret_cast<const V3f*>(row + lane_el);   // ONE 12-byte load (global_load_dwordx3): written element by element the compiler emits three
            c0[it] = t3.a[0]; c1[it] = t3.a[1]; c2[it] = t3.a[2];          // dword loads at a 12-byte lane stride - 48 instead of 16 trips through the address unit per wavefront
        }
    };
    f3 dc{0.f, 0.f, 0.f};
    if (MODEL && on) dc = ld3(a.sh0, gmine);
#pragma unroll
    for (int it = 0; it < HALF; ++it) fetch(it);
    {   // phase 1
        float b[25];
#pragma unroll
        for (int k = 0; k < 25; ++k) b[k] = 0.f;
        if (on) {
            if (degree >= 1) { const float inorm = 1.f / sqrtf(d.x * d.x + d.y * d.y + d.z * d.z); d = d * inorm; }
            sh_basis<false>(degree, d.x, d.y, d.z, b, nullptr, nullptr, nullptr);
        } // masked-out rows: b = 0 -> colour 0
#pragma unroll
        for (int k = 0; k < LPG; ++k) lds[lane * (LPG + 1) + k] = (k < 25) ? b[k] : 0.f;
        if (MODEL) { lds_dc[lane * 3] = dc.x; lds_dc[lane * 3 + 1] = dc.y; lds_dc[lane * 3 + 2] = dc.z; }
    }
    __syncthreads();
#pragma unroll
    for (int it = HALF; it < LPG; ++it) fetch(it);
#pragma unroll
    for (int it = 0; it < LPG; ++it) {
        const uint32_t gl = it * GPI + lane / LPG;
        const uint32_t g = g0 + gl;
        const float bk = lds[gl * (LPG + 1) + k];
        if (MODEL && k == 0) { c0[it] = lds_dc[gl * 3]; c1[it] = lds_dc[gl * 3 + 1]; c2[it] = lds_dc[gl * 3 + 2]; }   // (0 for masked-out Gaussians: dc was never loaded)
        float r0 = bk * c0[it], r1 = bk * c1[it], r2 = bk * c2[it];
        r0 = group_sum<LPG>(r0); r1 = group_sum<LPG>(r1); r2 = group_sum<LPG>(r2);
        if (k == 0 && g < a.n) {
            if (LFS_SH_FWD_UNCOND && !((vis >> gl) & 1ull)) { r0 = 0.f; r1 = 0.f; r2 = 0.f; } // (0 x an un-masked non-finite coefficient would be NaN)
            if (MODEL) { r0 = fmaxf(r0 + 0.5f, 0.f); r1 = fmaxf(r1 + 0.5f, 0.f); r2 = fmaxf(r2 + 0.5f, 0.f); }
            const size_t cs = (MODEL && a.cs) ? a.cs : 3;
            colors[cs * g] = r0; colors[cs * g + 1] = r1; colors[cs * g + 2] = r2;
            if (MODEL && a.rec_rgb != nullptr && ((vis >> gl) & 1ull)) { // (uniform pointer test) pipelined step: the projection kernel has written the record of every visible Gaussian already
                float* rr = a.rec_rgb + size_t(a.rec_stride) * g;
                rr[0] = r0; rr[1] = r1; rr[2] = r2;
            }
        }
    }
}
template <int LPG, bool MODEL>
__global__ void __launch_bounds__(64) sh_fwd_kernel(const ShArgs a, float* __restrict__ colors) { sh_fwd_block<LPG, MODEL>(a, colors, blockIdx.x * 64u); }
// The same blocks walked by a FIXED number of wavefronts (the pipelined step's side stream, gut_step.hip): a grid of one-wavefront workgroups as large as the problem
// fills every wave slot of the chip for its whole run time, and a kernel of the main stream whose workgroups need four free slots on one CU at once then starts when this
// one ENDS (rocprofv3 trace, profiles/r06/pipeline_slot_starvation.txt: the 56-us projection kernel took 308 us beside a 300-us single-wave grid). A few wavefronts per
// CU, each with a whole block's coefficient rows in flight, leave the other slots free and still move several TB/s.
template <int LPG>
__global__ void __launch_bounds__(64) sh_fwd_persistent_kernel(const ShArgs a, float* __restrict__ colors) {
    for (uint32_t g0 = blockIdx.x * 64u; g0 < a.n; g0 += gridDim.x * 64u) {
        sh_fwd_block<LPG, true>(a, colors, g0);
        __syncthreads();   // (the next block's basis overwrites the LDS rows)
    }
}


// op   : v_coeffs [n,K,3] fully written, v_dirs [n,3] (or NULL) fully written.
// model: v_colors = dL/d(clamped colors), the clamp passes where the stored colour is > 0; v_sh0 / v_shN written
//        (ACCUM = false) or added to (ACCUM = true: second and later views of a step); v_means += dL/d(dirs).
// ADAM (model form, single view): the higher-degree gradient rows are not stored at all - basis * dL/dcolour is consumed by the
//        Adam update of shN on the spot (v_shN unused; adam.m / adam.v = its moments). Saves the 180 B / Gaussian write here and the
//        180 B / Gaussian read in the optimizer kernel; element-wise identical to sh_bwd followed by adam_step on shN.
// m0 / v0 / s0 (optional): the same for the degree-0 coefficients sh0 (then v_sh0 is not written either) - the all-inline training step.
struct ShAdam { float* m; float* v; AdamScalars s; float* m0 = nullptr; float* v0 = nullptr; AdamScalars s0 = AdamScalars{}; };

template <int LPG, bool MODEL, bool ACCUM, bool ADAM = false>
__global__ void __launch_bounds__(64) sh_bwd_kernel(const ShArgs a, const float* __restrict__ v_colors,
                                                    float* __restrict__ v_coeffs, float* __restrict__ v_sh0, float* __restrict__ v_shN,
                                                    float* __restrict__ v_dirs, const ShAdam adam = ShAdam{}) {
    __shared__ float lds[64 * (LPG + 1)];
    __shared__ float ldv[64 * 3];
    if (ADAM && a.abort_flag != nullptr && *a.abort_flag != 0) return; // (uniform) the step is being re-run with larger buffers: no update from this attempt
    const uint32_t lane = threadIdx.x;
    const uint32_t g0 = blockIdx.x * 64u;
    const int degree = a.degree;
    const int Kd = (degree + 1) * (degree + 1);
    const bool want_dirs = (v_dirs != nullptr) && degree >= 1;
    // phase 1 (lane = Gaussian)
    const uint32_t gmine = g0 + lane;
    // every operand of phase 1 is requested at once - visibility, direction, dL/dcolour, the stored colour - and masked afterwards: one memory round trip
    // instead of three in series (radii.x -> radii.y -> the rest); the rows of a masked-out Gaussian are read and ignored
    bool on = false;
    f3 d{0.f, 0.f, 0.f};
    float inorm = 1.f;
    {
        float b[25];
#pragma unroll
        for (int k = 0; k < 25; ++k) b[k] = 0.f;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (gmine < a.n) {
            f3 dr = sh_dir<MODEL>(a, gmine);
            const size_t vs = (MODEL && a.vs) ? a.vs : 3, cs = (MODEL && a.cs) ? a.cs : 3;
            float w0, w1, w2, k0 = 1.f, k1 = 1.f, k2 = 1.f;
            if (MODEL && a.vs == 16 && ((reinterpret_cast<uintptr_t>(v_colors) - 4) & 15) == 0) { // rows of a rasterizer accumulator (3DGUT: slots 13..15, fastgs: 5..7): ONE aligned 16-byte load instead of three strided dwords
                const float4 r = *reinterpret_cast<const float4*>(v_colors + 16 * size_t(gmine) - 1);
                w0 = r.y; w1 = r.z; w2 = r.w;
            } else { w0 = v_colors[vs * gmine]; w1 = v_colors[vs * gmine + 1]; w2 = v_colors[vs * gmine + 2]; }
            if (MODEL) { k0 = a.colors[cs * gmine]; k1 = a.colors[cs * gmine + 1]; k2 = a.colors[cs * gmine + 2]; }
            if (MODEL && a.mask_u32 == nullptr && a.radii != nullptr) { const int2 r = *reinterpret_cast<const int2*>(a.radii + 2 * size_t(gmine)); on = r.x > 0 && r.y > 0; }
            else on = sh_on<MODEL>(a, gmine);
            if (on) {
                d = dr;
                if (degree >= 1) { inorm = 1.f / sqrtf(d.x * d.x + d.y * d.y + d.z * d.z); d = d * inorm; }
                sh_basis<false>(degree, d.x, d.y, d.z, b, nullptr, nullptr, nullptr);
                v0 = (k0 > 0.f) ? w0 : 0.f; v1 = (k1 > 0.f) ? w1 : 0.f; v2 = (k2 > 0.f) ? w2 : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < LPG; ++k) lds[lane * (LPG + 1) + k] = (k < 25) ? b[k] : 0.f;
        ldv[lane * 3] = v0; ldv[lane * 3 + 1] = v1; ldv[lane * 3 + 2] = v2;
    }
    __syncthreads();
    // phase 2 (lane = (Gaussian, basis))
    constexpr int GPI = 64 / LPG;
    const int k = lane % LPG;
    for (int it = 0; it < LPG; ++it) {
        const uint32_t gl = it * GPI + lane / LPG;
        const uint32_t g = g0 + gl;
        if (g >= a.n || uint32_t(k) >= a.K) continue;
        const float bk = lds[gl * (LPG + 1) + k]; // 0 for masked-out Gaussians and for k >= Kd
        const float v0 = ldv[gl * 3], v1 = ldv[gl * 3 + 1], v2 = ldv[gl * 3 + 2];
        const float o0 = bk * v0, o1 = bk * v1, o2 = bk * v2;
        float sk = 0.f;
        if (ADAM && k == 0 && adam.m0 != nullptr) { // sh0 row: basis 0 is constant (no direction gradient); update in place, nothing stored
            const size_t e = size_t(g) * 3;
            float* pp = const_cast<float*>(a.sh0) + e;
            float p0 = pp[0], p1 = pp[1], p2 = pp[2];
            float m0 = adam.m0[e], m1 = adam.m0[e + 1], m2 = adam.m0[e + 2], q0 = adam.v0[e], q1 = adam.v0[e + 1], q2 = adam.v0[e + 2];
            adam_elem(p0, m0, q0, o0, adam.s0); adam_elem(p1, m1, q1, o1, adam.s0); adam_elem(p2, m2, q2, o2, adam.s0);
            pp[0] = p0; pp[1] = p1; pp[2] = p2;
            adam.m0[e] = m0; adam.m0[e + 1] = m1; adam.m0[e + 2] = m2; adam.v0[e] = q0; adam.v0[e + 1] = q1; adam.v0[e + 2] = q2;
        } else if (ADAM && k >= 1) {
            // shN[g][k-1][:] : read once (the direction gradient needs the pre-update value), update, write back.
            // (Loading four rows ahead per lane was measured slower: 0.31 vs 0.23 ms.)
            const size_t e = (size_t(g) * (a.K - 1) + (k - 1)) * 3;
            float* pp = const_cast<float*>(a.shN) + e;
            float p0 = pp[0], p1 = pp[1], p2 = pp[2];
            if (want_dirs && k < Kd) sk = p0 * v0 + p1 * v1 + p2 * v2;
            float m0 = adam.m[e], m1 = adam.m[e + 1], m2 = adam.m[e + 2], q0 = adam.v[e], q1 = adam.v[e + 1], q2 = adam.v[e + 2];
            adam_elem(p0, m0, q0, o0, adam.s); adam_elem(p1, m1, q1, o1, adam.s); adam_elem(p2, m2, q2, o2, adam.s);
            pp[0] = p0; pp[1] = p1; pp[2] = p2;
            adam.m[e] = m0; adam.m[e + 1] = m1; adam.m[e + 2] = m2; adam.v[e] = q0; adam.v[e + 1] = q1; adam.v[e + 2] = q2;
        } else {
            float* vc = sh_coef<MODEL>(v_coeffs, v_sh0, v_shN, a.K, g, k);
            if (ACCUM) { if (bk != 0.f) { vc[0] += o0; vc[1] += o1; vc[2] += o2; } }
            else { vc[0] = o0; vc[1] = o1; vc[2] = o2; }
            if (want_dirs && k >= 1 && k < Kd && (v0 != 0.f || v1 != 0.f || v2 != 0.f)) { // (basis 0 is constant: no direction gradient, sh0 is not read)
                const float* cf = sh_coef<MODEL>(a.coeffs, a.sh0, a.shN, a.K, g, k);
                sk = cf[0] * v0 + cf[1] * v1 + cf[2] * v2;
            }
        }
        if (want_dirs) lds[gl * (LPG + 1) + k] = sk;
    }
    if (v_dirs == nullptr) return;
    __syncthreads();
    // phase 3 (lane = Gaussian)
    float ox = 0.f, oy = 0.f, oz = 0.f;
    if (on && want_dirs) {
        float b[25], bx[25], by[25], bz[25];
        sh_basis<true>(degree, d.x, d.y, d.z, b, bx, by, bz);
        float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
        for (int kk = 1; kk < LPG && kk < 25; ++kk) {
            const float sk = lds[lane * (LPG + 1) + kk];
            gx += bx[kk] * sk; gy += by[kk] * sk; gz += bz[kk] * sk;
        }
        // tangent-plane projection + chain through the normalisation
        const float dd = gx * d.x + gy * d.y + gz * d.z;
        ox = (gx - dd * d.x) * inorm; oy = (gy - dd * d.y) * inorm; oz = (gz - dd * d.z) * inorm;
    }
    if (gmine < a.n) {
        if (MODEL && a.dirs_store) {
            const size_t ds = a.ds ? a.ds : 3;
            if (ds == 3) st3(v_dirs, gmine, f3{ox, oy, oz});
            else { v_dirs[ds * gmine] = ox; v_dirs[ds * gmine + 1] = oy; v_dirs[ds * gmine + 2] = oz; } // (0 when off)
        } else if (MODEL) { if (on && want_dirs) { const f3 o = ld3(v_dirs, gmine); st3(v_dirs, gmine, f3{o.x + ox, o.y + oy, o.z + oz}); } }
        else st3(v_dirs, gmine, f3{ox, oy, oz});
    }
}

// ---- the SH backward of the PIPELINED training step (csrc/gut_step.hip), in two kernels on two streams ----
// sh_bwd_kernel<.., ADAM> does three things per Gaussian: dL/d(dirs) (needs every coefficient row as it was BEFORE the update), and the Adam updates of sh0 and shN - 1.1 GB
// of read-modify-write at 1 M Gaussians, a quarter of a millisecond on the HBM, while the next thing the step needs is dL/d(dirs) alone (the means update, then the next
// projection). The pipelined step splits it:
//   sh_pipe_dirs_kernel (main stream) : phases 1 - 3 of sh_bwd_kernel without any coefficient write: reads the rows (only those of Gaussians with a non-zero dL/dcolour),
//                                       writes dL/d(dirs) [n,3] and, per Gaussian, the 32-byte hand-over row {unit direction (3), visible, masked dL/dcolour (3), 0} -
//                                       everything the update kernel needs from buffers the main stream is about to overwrite (means, radii, colours, accumulator rows)
//   sh_pipe_adam_kernel (side stream) : phase 1 from the hand-over row, phase 2 = the ADAM branch of sh_bwd_kernel. Runs UNDER the finish pass, the next step's projection,
//                                       tile binning, sort and culling - kernels that leave the HBM idle.
// Same arithmetic in the same order as sh_bwd_kernel<LPG, true, false, true> (the direction is handed over AFTER its normalisation, the colour gradient after the clamp
// mask): bit-identical parameters, moments and dL/d(dirs).
template <int LPG>
__global__ void __launch_bounds__(64) sh_pipe_dirs_kernel(const ShArgs a, const float* __restrict__ v_colors /* accumulator rows + 13, stride 16 */,
                                                          float* __restrict__ v_dirs, float4* __restrict__ handover /* [n][2] */,
                                                          const int32_t* __restrict__ abort_flag, int32_t* __restrict__ abort_snapshot) {
    __shared__ float lds[64 * (LPG + 1)];
    __shared__ float ldv[64 * 3];
    const uint32_t lane = threadIdx.x;
    const uint32_t g0 = blockIdx.x * 64u;
    if (blockIdx.x == 0 && lane == 0) *abort_snapshot = *abort_flag; // the side stream must not look at the live flag: the next step's scan rewrites it
    const int degree = a.degree;
    const int Kd = (degree + 1) * (degree + 1);
    const uint32_t gmine = g0 + lane;
    bool on = false;
    f3 d{0.f, 0.f, 0.f};
    float inorm = 1.f;
    {
        float b[25];
#pragma unroll
        for (int k = 0; k < 25; ++k) b[k] = 0.f;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (gmine < a.n) {
            const f3 dr = sh_dir<true>(a, gmine);
            const float4 r = *reinterpret_cast<const float4*>(v_colors + 16 * size_t(gmine) - 1); // slots 12..15 of the row: one aligned 16-byte load
            const float k0 = a.colors[3 * size_t(gmine)], k1 = a.colors[3 * size_t(gmine) + 1], k2 = a.colors[3 * size_t(gmine) + 2];
            const int2 rr = *reinterpret_cast<const int2*>(a.radii + 2 * size_t(gmine));
            on = rr.x > 0 && rr.y > 0;
            if (on) {
                d = dr;
                if (degree >= 1) { inorm = 1.f / sqrtf(d.x * d.x + d.y * d.y + d.z * d.z); d = d * inorm; }
                sh_basis<false, (LPG > 16 ? 4 : 3)>(degree, d.x, d.y, d.z, b, nullptr, nullptr, nullptr);
                v0 = (k0 > 0.f) ? r.y : 0.f; v1 = (k1 > 0.f) ? r.z : 0.f; v2 = (k2 > 0.f) ? r.w : 0.f;
            }
            handover[2 * size_t(gmine)] = make_float4(d.x, d.y, d.z, on ? 1.f : 0.f);
            handover[2 * size_t(gmine) + 1] = make_float4(v0, v1, v2, 0.f);
        }
#pragma unroll
        for (int k = 0; k < LPG; ++k) lds[lane * (LPG + 1) + k] = (k < 25) ? b[k] : 0.f;
        ldv[lane * 3] = v0; ldv[lane * 3 + 1] = v1; ldv[lane * 3 + 2] = v2;
    }
    __syncthreads();
    constexpr int GPI = 64 / LPG;
    const int k = lane % LPG;
#pragma unroll 4
    for (int it = 0; it < LPG; ++it) {
        const uint32_t gl = it * GPI + lane / LPG;
        const uint32_t g = g0 + gl;
        const float v0 = ldv[gl * 3], v1 = ldv[gl * 3 + 1], v2 = ldv[gl * 3 + 2];
        float sk = 0.f;
        // s_k = coeff_k . dL/dcolour; a zero gradient gives 0 x (finite) = 0 whatever the row holds: its 12 bytes are not fetched
        if (g < a.n && k >= 1 && k < Kd && uint32_t(k) < a.K && (v0 != 0.f || v1 != 0.f || v2 != 0.f)) {
            const V3f p = *reinterpret_cast<const V3f*>(a.shN + (size_t(g) * (a.K - 1) + (k - 1)) * 3);
            sk = p.a[0] * v0 + p.a[1] * v1 + p.a[2] * v2;
        }
        lds[gl * (LPG + 1) + k] = sk;
    }
    __syncthreads();
    float ox = 0.f, oy = 0.f, oz = 0.f;
    if (on && degree >= 1) {
        float b[25], bx[25], by[25], bz[25];
        sh_basis<true, (LPG > 16 ? 4 : 3)>(degree, d.x, d.y, d.z, b, bx, by, bz);
        float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
        for (int kk = 1; kk < LPG && kk < 25; ++kk) {
            const float sk = lds[lane * (LPG + 1) + kk];
            gx += bx[kk] * sk; gy += by[kk] * sk; gz += bz[kk] * sk;
        }
        const float dd = gx * d.x + gy * d.y + gz * d.z;
        ox = (gx - dd * d.x) * inorm; oy = (gy - dd * d.y) * inorm; oz = (gz - dd * d.z) * inorm;
    }
    if (gmine < a.n) st3(v_dirs, gmine, f3{ox, oy, oz});
}

// A FIXED grid of one-wavefront workgroups walks the blocks of 64 Gaussians (see sh_fwd_persistent_kernel: a problem-sized single-wave grid would take every wave slot of
// the chip and starve the main stream's kernels). With only a few wavefronts per CU the bandwidth has to come from each wavefront: the rows of a block are updated in
// batches of SH_PIPE_U, and the loads of the NEXT batch (parameter, two moments: 3 x 12 bytes per lane and row) are in flight while one batch goes through Adam and is stored.
// SH_PIPE_DEPTH rows (parameter + two moments: 3 x 12 bytes per lane and row) are in flight per lane: row `it + SH_PIPE_DEPTH` is requested into the registers row `it`
// has just left. The sh0 rows are updated in the lane = Gaussian phase (one coalesced 768-byte block per tensor and wavefront; basis 0 is the constant), so that the
// lane = (Gaussian, basis) phase walks ONE tensor triple: wave-uniform row bases in SGPRs + one 32-bit per-lane offset. 57 VGPRs: two of these wavefronts per SIMD leave
// the main stream's kernels three quarters of the register file (the first form, double-buffered batches behind 64-bit per-lane addresses, took 131 each, and the
// projection kernel - 94 VGPRs - ran at 2 instead of 5 wavefronts per SIMD beside it: 150 instead of 56 us, profiles/r06/pipeline/lease11_timelines.txt).
#ifndef LFS_SH_PIPE_DEPTH
#define LFS_SH_PIPE_DEPTH 4
#endif
constexpr int SH_PIPE_DEPTH = LFS_SH_PIPE_DEPTH;
#if defined(LFS_PIPE_ADAM_WAVES) && !defined(LFS_EMULATE)
#define LFS_PIPE_ADAM_ATTR __attribute__((amdgpu_waves_per_eu(LFS_PIPE_ADAM_WAVES)))   // (A/B hook: a register budget for the kernel - 8 = 64 VGPRs, with spills)
#else
#define LFS_PIPE_ADAM_ATTR
#endif
template <int LPG>
__global__ void __launch_bounds__(64) LFS_PIPE_ADAM_ATTR sh_pipe_adam_kernel(const uint32_t n, const uint32_t K, const int degree, float* __restrict__ sh0, float* __restrict__ shN,
                                                          const float4* __restrict__ handover, const ShAdam adam, const int32_t* __restrict__ abort_snapshot) {
    __shared__ float lds[64 * (LPG + 1)];
    __shared__ float ldv[64 * 3];
    if (*abort_snapshot != 0) return; // (uniform) the attempt did not fit its buffers: no update, the host runs the step again
    const uint32_t lane = threadIdx.x;
    constexpr int GPI = 64 / LPG;
    constexpr int D = (LPG < SH_PIPE_DEPTH) ? LPG : SH_PIPE_DEPTH;
    const int k = lane % LPG;
    const uint32_t KK = K - 1;
    const bool row_k = k >= 1 && uint32_t(k) < K;
    const uint32_t lane_el = ((lane / LPG) * KK + uint32_t(k - 1)) * 3u;   // (k == 0 lanes: never used)
    float* const mN = adam.m; float* const vN = adam.v;
    for (uint32_t g0 = blockIdx.x * 64u; g0 < n; g0 += gridDim.x * 64u) {
        const uint32_t gmine = g0 + lane;
        auto row_ok = [&](const int it) { return row_k && (g0 + uint32_t(it) * GPI + lane / LPG) < n; };
        V3f P[D], M[D], Q[D];
        auto load = [&](const int it, const int slot) {
            if (row_ok(it)) {
                const size_t base = size_t(g0 + uint32_t(it) * GPI) * KK * 3u;   // (uniform)
                P[slot] = *reinterpret_cast<const V3f*>(shN + base + lane_el); M[slot] = *reinterpret_cast<const V3f*>(mN + base + lane_el);
                Q[slot] = *reinterpret_cast<const V3f*>(vN + base + lane_el);
            }
        };
        {
            float b[25];
#pragma unroll
            for (int kk = 0; kk < 25; ++kk) b[kk] = 0.f;
            float v0 = 0.f, v1 = 0.f, v2 = 0.f;
            bool on = false;
            if (gmine < n) {
                const float4 h0 = handover[2 * size_t(gmine)], h1 = handover[2 * size_t(gmine) + 1];
                on = h0.w != 0.f;
                if (on) sh_basis<false, (LPG > 16 ? 4 : 3)>(degree, h0.x, h0.y, h0.z, b, nullptr, nullptr, nullptr);
                v0 = h1.x; v1 = h1.y; v2 = h1.z;
            }
#pragma unroll
            for (int kk = 0; kk < LPG; ++kk) lds[lane * (LPG + 1) + kk] = (kk < 25) ? b[kk] : 0.f;
            ldv[lane * 3] = v0; ldv[lane * 3 + 1] = v1; ldv[lane * 3 + 2] = v2;
            if (gmine < n) {   // the sh0 row (sh_bwd_kernel's k == 0 lane): gradient = basis 0 (the constant) x dL/dcolour - after the polynomial's registers are free
                const float b0 = on ? 0.2820947917738781f : 0.f;
                V3f p = *reinterpret_cast<const V3f*>(sh0 + 3 * size_t(gmine)), m = *reinterpret_cast<const V3f*>(adam.m0 + 3 * size_t(gmine)),
                    q = *reinterpret_cast<const V3f*>(adam.v0 + 3 * size_t(gmine));
                adam_elem(p.a[0], m.a[0], q.a[0], b0 * v0, adam.s0); adam_elem(p.a[1], m.a[1], q.a[1], b0 * v1, adam.s0); adam_elem(p.a[2], m.a[2], q.a[2], b0 * v2, adam.s0);
                *reinterpret_cast<V3f*>(sh0 + 3 * size_t(gmine)) = p; *reinterpret_cast<V3f*>(adam.m0 + 3 * size_t(gmine)) = m; *reinterpret_cast<V3f*>(adam.v0 + 3 * size_t(gmine)) = q;
            }
        }
#pragma unroll
        for (int it = 0; it < D; ++it) load(it, it);   // (requested once the polynomial's temporaries are dead: see the register note above)
        __syncthreads();
#pragma unroll 1
        for (int it0 = 0; it0 < LPG; it0 += D) {   // (a real loop: fully unrolled the compiler hoists the loads of every later row and the allocation doubles)
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int it = it0 + u;
                if (row_ok(it)) {
                    const uint32_t gl = it * GPI + lane / LPG;
                    const float bk = lds[gl * (LPG + 1) + k];
                    const float o0 = bk * ldv[gl * 3], o1 = bk * ldv[gl * 3 + 1], o2 = bk * ldv[gl * 3 + 2];
                    V3f p = P[u], m = M[u], q = Q[u];
                    adam_elem(p.a[0], m.a[0], q.a[0], o0, adam.s); adam_elem(p.a[1], m.a[1], q.a[1], o1, adam.s); adam_elem(p.a[2], m.a[2], q.a[2], o2, adam.s);
                    const size_t base = size_t(g0 + uint32_t(it) * GPI) * KK * 3u;
                    *reinterpret_cast<V3f*>(shN + base + lane_el) = p; *reinterpret_cast<V3f*>(mN + base + lane_el) = m; *reinterpret_cast<V3f*>(vN + base + lane_el) = q;
                }
                if (it + D < LPG) load(it + D, u);
            }
        }
        __syncthreads();   // (the next block's lane = Gaussian phase overwrites the LDS rows)
    }
}

// wavefronts the side-stream kernels of the pipelined step run with: LFS_PIPE_WAVES per CU (default 8 = two of a SIMD's eight slots) x 256 CUs
static inline uint32_t sh_pipe_side_grid() {
    static uint32_t g = 0;
    if (g == 0) {
        uint32_t per_cu = 8;
#ifndef LFS_EMULATE
        if (const char* e = getenv("LFS_PIPE_WAVES")) { const int v = atoi(e); if (v >= 1 && v <= 32) per_cu = uint32_t(v); }
#endif
        g = 256u * per_cu;
    }
    return g;
}
static inline int lanes_for(uint32_t k) { return k <= 1 ? 1 : k <= 4 ? 4 : k <= 16 ? 16 : 32; }

template <bool MODEL>
static int sh_launch_fwd(const ShArgs& a, uint32_t Kcover, float* colors, hipStream_t s, uint32_t fixed_grid = 0) {
    const dim3 grid((a.n + 63) / 64), block(64);
    lfs::ProfScope prof("sh_fwd", s);
    if (MODEL && fixed_grid != 0 && fixed_grid < grid.x) { // the side stream of the pipelined step: a fixed number of wavefronts walk the blocks
        const dim3 pg(fixed_grid);
        switch (lanes_for(Kcover)) {
        case 1: hipLaunchKernelGGL((sh_fwd_persistent_kernel<1>), pg, block, 0, s, a, colors); break;
        case 4: hipLaunchKernelGGL((sh_fwd_persistent_kernel<4>), pg, block, 0, s, a, colors); break;
        case 16: hipLaunchKernelGGL((sh_fwd_persistent_kernel<16>), pg, block, 0, s, a, colors); break;
        default: hipLaunchKernelGGL((sh_fwd_persistent_kernel<32>), pg, block, 0, s, a, colors); break;
        }
        return (int)hipGetLastError();
    }
    switch (lanes_for(Kcover)) {
    case 1: hipLaunchKernelGGL((sh_fwd_kernel<1, MODEL>), grid, block, 0, s, a, colors); break;
    case 4: hipLaunchKernelGGL((sh_fwd_kernel<4, MODEL>), grid, block, 0, s, a, colors); break;
    case 16: hipLaunchKernelGGL((sh_fwd_kernel<16, MODEL>), grid, block, 0, s, a, colors); break;
    default: hipLaunchKernelGGL((sh_fwd_kernel<32, MODEL>), grid, block, 0, s, a, colors); break;
    }
    return (int)hipGetLastError();
}

template <bool MODEL, bool ACCUM>
static int sh_launch_bwd(const ShArgs& a, const float* v_colors, float* v_coeffs, float* v_sh0, float* v_shN, float* v_dirs, hipStream_t s) {
    const dim3 grid((a.n + 63) / 64), block(64);
    lfs::ProfScope prof("sh_bwd", s);
    switch (lanes_for(a.K)) { // every one of the K rows of v_coeffs is written
    case 1: hipLaunchKernelGGL((sh_bwd_kernel<1, MODEL, ACCUM>), grid, block, 0, s, a, v_colors, v_coeffs, v_sh0, v_shN, v_dirs); break;
    case 4: hipLaunchKernelGGL((sh_bwd_kernel<4, MODEL, ACCUM>), grid, block, 0, s, a, v_colors, v_coeffs, v_sh0, v_shN, v_dirs); break;
    case 16: hipLaunchKernelGGL((sh_bwd_kernel<16, MODEL, ACCUM>), grid, block, 0, s, a, v_colors, v_coeffs, v_sh0, v_shN, v_dirs); break;
    default: hipLaunchKernelGGL((sh_bwd_kernel<32, MODEL, ACCUM>), grid, block, 0, s, a, v_colors, v_coeffs, v_sh0, v_shN, v_dirs); break;
    }
    return (int)hipGetLastError();
}

static int sh_launch_bwd_adam(const ShArgs& a, const float* v_colors, float* v_sh0, float* v_dirs, const ShAdam& adam, hipStream_t s) {
    const dim3 grid((a.n + 63) / 64), block(64);
    lfs::ProfScope prof("sh_bwd_adam", s);
    switch (lanes_for(a.K)) {
    case 1: return LFS_E_INVALID; // K == 1: there is no shN
    case 4: hipLaunchKernelGGL((sh_bwd_kernel<4, true, false, true>), grid, block, 0, s, a, v_colors, nullptr, v_sh0, nullptr, v_dirs, adam); break;
    case 16: hipLaunchKernelGGL((sh_bwd_kernel<16, true, false, true>), grid, block, 0, s, a, v_colors, nullptr, v_sh0, nullptr, v_dirs, adam); break;
    default: hipLaunchKernelGGL((sh_bwd_kernel<32, true, false, true>), grid, block, 0, s, a, v_colors, nullptr, v_sh0, nullptr, v_dirs, adam); break;
    }
    return (int)hipGetLastError();
}

// ---- multi-view forms (SH-sharded data parallelism, dist.ShExchange): the owner of a block of Gaussians evaluates SH for the views of
// ALL ranks in one launch. Coefficient rows are fetched once and kept in registers across the views (the per-view launches
// re-read 192 B per Gaussian and view); the backward accumulates the coefficient gradient over the views in registers and either
// stores it once or, ADAM, hands it straight to the Adam update of shN (as sh_bwd_kernel<.., ADAM>). Per-view operands
// (radii, colors, v_colors) are [V, view_stride, ...] with the first n rows of each view used.
struct ShViews {
    uint32_t n, K; int degree; uint32_t V, view_stride;
    const float* means; const float* viewmats; const float* sh0; const float* shN; const int32_t* radii; const float* colors;
};

template <int LPG>
__global__ void __launch_bounds__(64) sh_views_fwd_kernel(const ShViews a, float* __restrict__ colors) {
    __shared__ float lds[64 * (LPG + 1)];
    const uint32_t lane = threadIdx.x, g0 = blockIdx.x * 64u, gmine = g0 + lane;
    const int degree = a.degree, Kd = (degree + 1) * (degree + 1);
    constexpr int GPI = 64 / LPG;
    const int k = lane % LPG;
    float c0[LPG], c1[LPG], c2[LPG];
#pragma unroll
    for (int it = 0; it < LPG; ++it) {
        const uint32_t g = g0 + it * GPI + lane / LPG;
        c0[it] = c1[it] = c2[it] = 0.f;
        if (g < a.n && k < Kd) {
            const float* cf = sh_coef<true>((const float*)nullptr, a.sh0, a.shN, a.K, g, k);
            c0[it] = cf[0]; c1[it] = cf[1]; c2[it] = cf[2];
        }
    }
    f3 m{0.f, 0.f, 0.f};
    if (gmine < a.n) m = ld3(a.means, gmine);
    for (uint32_t v = 0; v < a.V; ++v) {
        {   // phase 1 (lane = Gaussian)
            float b[25];
#pragma unroll
            for (int kk = 0; kk < 25; ++kk) b[kk] = 0.f;
            const int32_t* rr = a.radii + (size_t(v) * a.view_stride + gmine) * 2;
            if (gmine < a.n && rr[0] > 0 && rr[1] > 0) {
                const f3 cp = campos_of(a.viewmats + 16 * v);
                f3 d{m.x - cp.x, m.y - cp.y, m.z - cp.z};
                if (degree >= 1) { const float inorm = 1.f / sqrtf(d.x * d.x + d.y * d.y + d.z * d.z); d = d * inorm; }
                sh_basis<false>(degree, d.x, d.y, d.z, b, nullptr, nullptr, nullptr);
            }
#pragma unroll
            for (int kk = 0; kk < LPG; ++kk) lds[lane * (LPG + 1) + kk] = (kk < 25) ? b[kk] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < LPG; ++it) {
            const uint32_t gl = it * GPI + lane / LPG, g = g0 + gl;
            const float bk = lds[gl * (LPG + 1) + k];
            const float r0 = group_sum<LPG>(bk * c0[it]), r1 = group_sum<LPG>(bk * c1[it]), r2 = group_sum<LPG>(bk * c2[it]);
            if (k == 0 && g < a.n) {
                float* o = colors + (size_t(v) * a.view_stride + g) * 3;
                o[0] = fmaxf(r0 + 0.5f, 0.f); o[1] = fmaxf(r1 + 0.5f, 0.f); o[2] = fmaxf(r2 + 0.5f, 0.f);
            }
        }
        __syncthreads();
    }
}

template <int LPG, bool ADAM>
__global__ void __launch_bounds__(64) sh_views_bwd_kernel(const ShViews a, const float* __restrict__ v_colors, const int accumulate,
                                                          float* __restrict__ v_sh0, float* __restrict__ v_shN, float* __restrict__ v_means,
                                                          const ShAdam adam) {
    __shared__ float lds[64 * (LPG + 1)];
    __shared__ float ldv[64 * 3];
    const uint32_t lane = threadIdx.x, g0 = blockIdx.x * 64u, gmine = g0 + lane;
    const int degree = a.degree, Kd = (degree + 1) * (degree + 1);
    const bool want_dirs = degree >= 1;
    constexpr int GPI = 64 / LPG;
    const int k = lane % LPG;
    // gradient accumulators over the views (the coefficient rows the dL/d(dirs) term needs sit in LDS: below)
    float a0[LPG], a1[LPG], a2[LPG];
#pragma unroll
    for (int it = 0; it < LPG; ++it) a0[it] = a1[it] = a2[it] = 0.f;
    f3 m{0.f, 0.f, 0.f};
    if (gmine < a.n) m = ld3(a.means, gmine);
    // Round 5: the higher-degree coefficient rows of the wavefront's 64 Gaussians are fetched ONCE (coalesced 12-byte loads) into LDS and read from there by every
    // view's dL/d(dirs) term and by the Adam epilogue - they used to be re-read from global memory per view (48 dword loads per wavefront and view: L2 hits, but
    // 8 views = 1.4 GB through the address unit at 1 M Gaussians, the multi-view pass of an 8-GPU rank took 0.66 ms against 0.23 for one view). Row stride K - 1
    // floats x 3 (45 at degree 3: odd, the lanes of a read spread over the banks). Same values, same arithmetic: results are bit-identical.
    LFS_DYN_LDS(float, s_cof);   // [64][(K - 1) * 3]
    const uint32_t cof_stride = (a.K - 1u) * 3u;
    if (want_dirs || ADAM) {
#pragma unroll
        for (int it = 0; it < LPG; ++it) {
            const uint32_t gl = it * GPI + lane / LPG, g = g0 + gl;
            if (k >= 1 && uint32_t(k) < a.K && g < a.n) {
                const V3f t3 = *reinterpret_cast<const V3f*>(a.shN + (size_t(g) * (a.K - 1) + (k - 1)) * 3);
                float* dst = s_cof + gl * cof_stride + (k - 1) * 3;
                dst[0] = t3.a[0]; dst[1] = t3.a[1]; dst[2] = t3.a[2];
            }
        }
    }
    float ox = 0.f, oy = 0.f, oz = 0.f;
    for (uint32_t v = 0; v < a.V; ++v) {
        f3 d{0.f, 0.f, 0.f};
        float inorm = 1.f;
        bool on = false;
        {   // phase 1 (lane = Gaussian)
            float b[25];
#pragma unroll
            for (int kk = 0; kk < 25; ++kk) b[kk] = 0.f;
            float v0 = 0.f, v1 = 0.f, v2 = 0.f;
            const size_t row = size_t(v) * a.view_stride + gmine;
            if (gmine < a.n) {
                if (a.radii != nullptr) { const int32_t* rr = a.radii + row * 2; on = rr[0] > 0 && rr[1] > 0; }
                // radii == NULL (the factored gradient exchange, dist.ColorGradExchange): the rows arrive already masked by visibility and by the clamp of the
                // rank that rendered the view - a Gaussian takes part in a view exactly when its row is not zero
                else on = v_colors[row * 3] != 0.f || v_colors[row * 3 + 1] != 0.f || v_colors[row * 3 + 2] != 0.f;
            }
            if (on) {
                const f3 cp = campos_of(a.viewmats + 16 * v);
                d = {m.x - cp.x, m.y - cp.y, m.z - cp.z};
                if (degree >= 1) { inorm = 1.f / sqrtf(d.x * d.x + d.y * d.y + d.z * d.z); d = d * inorm; }
                sh_basis<false, (LPG <= 4 ? 1 : LPG <= 16 ? 3 : 4)>(degree, d.x, d.y, d.z, b, nullptr, nullptr, nullptr);
                v0 = v_colors[row * 3]; v1 = v_colors[row * 3 + 1]; v2 = v_colors[row * 3 + 2];
                if (a.colors != nullptr) {                        // clamp_min backward
                    if (!(a.colors[row * 3] > 0.f)) v0 = 0.f;
                    if (!(a.colors[row * 3 + 1] > 0.f)) v1 = 0.f;
                    if (!(a.colors[row * 3 + 2] > 0.f)) v2 = 0.f;
                }
            }
#pragma unroll
            for (int kk = 0; kk < LPG; ++kk) lds[lane * (LPG + 1) + kk] = (kk < 25) ? b[kk] : 0.f;
            ldv[lane * 3] = v0; ldv[lane * 3 + 1] = v1; ldv[lane * 3 + 2] = v2;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < LPG; ++it) { // phase 2 (lane = (Gaussian, basis))
            const uint32_t gl = it * GPI + lane / LPG;
            const float bk = lds[gl * (LPG + 1) + k];
            const float v0 = ldv[gl * 3], v1 = ldv[gl * 3 + 1], v2 = ldv[gl * 3 + 2];
            a0[it] += bk * v0; a1[it] += bk * v1; a2[it] += bk * v2;
            if (want_dirs) {
                float sk = 0.f;
                if (k >= 1 && k < Kd && g0 + gl < a.n && (v0 != 0.f || v1 != 0.f || v2 != 0.f)) { // (basis 0 is constant: sh0 is not needed)
                    const float* cf = s_cof + gl * cof_stride + (k - 1) * 3;   // (written before the first barrier of the view loop)
                    sk = cf[0] * v0 + cf[1] * v1 + cf[2] * v2;
                }
                lds[gl * (LPG + 1) + k] = sk;
            }
        }
        __syncthreads();
        if (on && want_dirs) { // phase 3 (lane = Gaussian)
            float b[25], bx[25], by[25], bz[25];
            sh_basis<true, (LPG <= 4 ? 1 : LPG <= 16 ? 3 : 4)>(degree, d.x, d.y, d.z, b, bx, by, bz);
            float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
            for (int kk = 1; kk < LPG && kk < 25; ++kk) {
                const float sk = lds[lane * (LPG + 1) + kk];
                gx += bx[kk] * sk; gy += by[kk] * sk; gz += bz[kk] * sk;
            }
            const float dd = gx * d.x + gy * d.y + gz * d.z;
            ox += (gx - dd * d.x) * inorm; oy += (gy - dd * d.y) * inorm; oz += (gz - dd * d.z) * inorm;
        }
        __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < LPG; ++it) {
        const uint32_t g = g0 + it * GPI + lane / LPG;
        if (g >= a.n || uint32_t(k) >= a.K) continue;
        if (k == 0) {
            float* o = v_sh0 + size_t(g) * 3;
            if (accumulate) { o[0] += a0[it]; o[1] += a1[it]; o[2] += a2[it]; }
            else { o[0] = a0[it]; o[1] = a1[it]; o[2] = a2[it]; }
            continue;
        }
        const size_t e = (size_t(g) * (a.K - 1) + (k - 1)) * 3;
        if (ADAM) {
            float* pp = const_cast<float*>(a.shN) + e;
            const float* cf = s_cof + (it * GPI + lane / LPG) * cof_stride + (k - 1) * 3;   // the row as it was fetched at the top: nothing has written shN since
            float q0 = cf[0], q1 = cf[1], q2 = cf[2];
            float m0 = adam.m[e], m1 = adam.m[e + 1], m2 = adam.m[e + 2], s0 = adam.v[e], s1 = adam.v[e + 1], s2 = adam.v[e + 2];
            adam_elem(q0, m0, s0, a0[it], adam.s); adam_elem(q1, m1, s1, a1[it], adam.s); adam_elem(q2, m2, s2, a2[it], adam.s);
            pp[0] = q0; pp[1] = q1; pp[2] = q2;
            adam.m[e] = m0; adam.m[e + 1] = m1; adam.m[e + 2] = m2; adam.v[e] = s0; adam.v[e + 1] = s1; adam.v[e + 2] = s2;
        } else if (v_shN == nullptr) { // (nobody reads the higher-degree gradient: iteration <= 1000, fused_adam.cpp:68-70)
        } else if (accumulate) { v_shN[e] += a0[it]; v_shN[e + 1] += a1[it]; v_shN[e + 2] += a2[it]; }
        else { v_shN[e] = a0[it]; v_shN[e + 1] = a1[it]; v_shN[e + 2] = a2[it]; }
    }
    if (gmine < a.n && want_dirs) { const f3 o = ld3(v_means, gmine); st3(v_means, gmine, f3{o.x + ox, o.y + oy, o.z + oz}); }
}

// used by fastgs_{prep,blend}.hip: SH colour of visible primitives written straight into the blend records (stride in floats),
// and its backward reading dL/dcolour from the blend accumulator rows
int sh_records_fwd(uint32_t n, uint32_t K, uint32_t degree, const float* means, const float* campos, const float* sh0, const float* shN,
                   const uint32_t* mask_u32, float* colors, uint32_t colors_stride, hipStream_t s) {
    ShArgs a{};
    a.n = n; a.K = K; a.degree = int(degree); a.means = means; a.campos = campos; a.sh0 = sh0; a.shN = shN; a.mask_u32 = mask_u32; a.cs = colors_stride;
    return sh_launch_fwd<true>(a, (degree + 1) * (degree + 1), colors, s);
}
struct ShAdamArgs { float* exp_avg; float* exp_avg_sq; float lr, beta1, beta2, eps, bc1_rcp, bc2_sqrt_rcp; }; // as declared in lfs_fastgs.cuh
int sh_records_bwd(uint32_t n, uint32_t K, uint32_t degree, const float* means, const float* campos, const float* sh0, const float* shN,
                   const uint32_t* mask_u32, const float* colors, uint32_t colors_stride, const float* v_colors, uint32_t v_stride,
                   float* v_sh0, float* v_shN, float* v_means, hipStream_t s, const ShAdamArgs* adam) {
    ShArgs a{};
    a.n = n; a.K = K; a.degree = int(degree); a.means = means; a.campos = campos; a.sh0 = sh0; a.shN = shN; a.mask_u32 = mask_u32;
    a.colors = colors; a.cs = colors_stride; a.vs = v_stride;
    if (adam && K > 1) {
        const ShAdam ad{adam->exp_avg, adam->exp_avg_sq, AdamScalars{adam->lr, adam->beta1, adam->beta2, adam->eps, adam->bc1_rcp, adam->bc2_sqrt_rcp}};
        return sh_launch_bwd_adam(a, v_colors, v_sh0, v_means, ad, s);
    }
    return sh_launch_bwd<true, false>(a, v_colors, nullptr, v_sh0, v_shN, v_means, s);
}

} // namespace lfs

extern "C" int lfs_spherical_harmonics_fwd(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* dirs, const float* coeffs, const uint8_t* masks,
    float* colors, lfs_stream_t stream) {
    const uint32_t Kd = (degrees_to_use + 1) * (degrees_to_use + 1);
    if (degrees_to_use > 4 || Kd > K) return LFS_E_INVALID;
    if (n == 0) return LFS_OK;
    if (!dirs || !coeffs || !colors) return LFS_E_INVALID;
    lfs::ShArgs a{};
    a.n = n; a.K = K; a.degree = int(degrees_to_use); a.dirs = dirs; a.coeffs = coeffs; a.masks = masks;
    return lfs::sh_launch_fwd<false>(a, Kd, colors, (hipStream_t)stream);
}

extern "C" int lfs_spherical_harmonics_bwd(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* dirs, const float* coeffs, const uint8_t* masks,
    const float* v_colors, float* v_coeffs, float* v_dirs, lfs_stream_t stream) {
    const uint32_t Kd = (degrees_to_use + 1) * (degrees_to_use + 1);
    if (degrees_to_use > 4 || Kd > K || K > 32) return LFS_E_INVALID;
    if (n == 0) return LFS_OK;
    if (!dirs || !coeffs || !v_colors || !v_coeffs) return LFS_E_INVALID;
    lfs::ShArgs a{};
    a.n = n; a.K = K; a.degree = int(degrees_to_use); a.dirs = dirs; a.coeffs = coeffs; a.masks = masks;
    return lfs::sh_launch_bwd<false, false>(a, v_colors, v_coeffs, nullptr, nullptr, v_dirs, (hipStream_t)stream);
}

// radii == NULL (the training step: the colours are evaluated BEFORE the projection, which writes them into the rasterizer's records): every Gaussian
int lfs::sh_model_fwd_impl(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* means, const float* viewmat, const float* sh0, const float* shN,
    const int32_t* radii, float* colors, hipStream_t stream) {
    const uint32_t Kd = (degrees_to_use + 1) * (degrees_to_use + 1);
    if (degrees_to_use > 4 || Kd > K || K > 32) return LFS_E_INVALID;
    if (n == 0) return LFS_OK;
    if (!means || !viewmat || !sh0 || (K > 1 && !shN) || !colors) return LFS_E_INVALID;
    lfs::ShArgs a{};
    a.n = n; a.K = K; a.degree = int(degrees_to_use); a.means = means; a.viewmat = viewmat; a.sh0 = sh0; a.shN = shN; a.radii = radii;
    return lfs::sh_launch_fwd<true>(a, Kd, colors, stream);
}

extern "C" int lfs_sh_model_fwd(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* means, const float* viewmat, const float* sh0, const float* shN,
    const int32_t* radii, float* colors, lfs_stream_t stream) {
    if (!radii) return LFS_E_INVALID;
    return lfs::sh_model_fwd_impl(n, K, degrees_to_use, means, viewmat, sh0, shN, radii, colors, (hipStream_t)stream);
}

extern "C" int lfs_sh_model_bwd(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* means, const float* viewmat, const float* sh0, const float* shN,
    const int32_t* radii, const float* colors, const float* v_colors, int accumulate,
    float* v_sh0, float* v_shN, float* v_means, lfs_stream_t stream) {
    const uint32_t Kd = (degrees_to_use + 1) * (degrees_to_use + 1);
    if (degrees_to_use > 4 || Kd > K || K > 32) return LFS_E_INVALID;
    if (n == 0) return LFS_OK;
    if (!means || !viewmat || !sh0 || (K > 1 && (!shN || !v_shN)) || !radii || !colors || !v_colors || !v_sh0 || !v_means) return LFS_E_INVALID;
    lfs::ShArgs a{};
    a.n = n; a.K = K; a.degree = int(degrees_to_use); a.means = means; a.viewmat = viewmat; a.sh0 = sh0; a.shN = shN; a.radii = radii; a.colors = colors;
    if (accumulate) return lfs::sh_launch_bwd<true, true>(a, v_colors, nullptr, v_sh0, v_shN, v_means, (hipStream_t)stream);
    return lfs::sh_launch_bwd<true, false>(a, v_colors, nullptr, v_sh0, v_shN, v_means, (hipStream_t)stream);
}

// lfs_sh_model_bwd with dL/dcolour read from the rasterizer's accumulator rows (slots 13..15 of 16 floats) and dL/d(dirs) WRITTEN to its own [n,3] array:
// the data-parallel step runs the SH backward BEFORE the finish pass, so that the shN gradient - 45 of a Gaussian's 59 floats - is final (and its
// all-reduce on the wire) while lfs_gut_finish_grads still runs (csrc/gut_step.hip).
int lfs::sh_model_bwd_rows_impl(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* means, const float* viewmat, const float* sh0, const float* shN,
    const int32_t* radii, const float* colors, const float* acc_rows, int accumulate, float* v_sh0, float* v_shN, float* v_dirs, hipStream_t stream,
    float* shN_exp_avg, float* shN_exp_avg_sq, const float* shN_scalars) {
    const uint32_t Kd = (degrees_to_use + 1) * (degrees_to_use + 1);
    if (degrees_to_use > 4 || Kd > K || K > 32) return LFS_E_INVALID;
    if (n == 0) return LFS_OK;
    const bool inline_adam = shN_exp_avg != nullptr;   // shN updated in place by this launch (one view per step): no shN gradient is written
    if (inline_adam && (accumulate || K < 2 || !shN_exp_avg_sq || !shN_scalars)) return LFS_E_INVALID;
    if (!means || !viewmat || !sh0 || (K > 1 && (!shN || (!v_shN && !inline_adam))) || !radii || !colors || !acc_rows || !v_sh0 || !v_dirs) return LFS_E_INVALID;
    lfs::ShArgs a{};
    a.n = n; a.K = K; a.degree = int(degrees_to_use); a.means = means; a.viewmat = viewmat; a.sh0 = sh0; a.shN = shN; a.radii = radii; a.colors = colors;
    a.vs = 16; a.dirs_store = true;
    if (inline_adam) {
        const float* t = shN_scalars;
        const lfs::ShAdam adam{shN_exp_avg, shN_exp_avg_sq, lfs::AdamScalars{t[0], t[1], t[2], t[3], t[4], t[5]}};
        return lfs::sh_launch_bwd_adam(a, acc_rows + 13, v_sh0, v_dirs, adam, stream);
    }
    if (accumulate) return lfs::sh_launch_bwd<true, true>(a, acc_rows + 13, nullptr, v_sh0, v_shN, v_dirs, stream);
    return lfs::sh_launch_bwd<true, false>(a, acc_rows + 13, nullptr, v_sh0, v_shN, v_dirs, stream);
}

extern "C" int lfs_sh_model_bwd_adam(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* means, const float* viewmat, const float* sh0, float* shN,
    const int32_t* radii, const float* colors, const float* v_colors, float* v_sh0, float* v_means,
    float* shN_exp_avg, float* shN_exp_avg_sq, float lr, float beta1, float beta2, float eps, float bias_correction1_rcp,
    float bias_correction2_sqrt_rcp, lfs_stream_t stream) {
    const uint32_t Kd = (degrees_to_use + 1) * (degrees_to_use + 1);
    if (degrees_to_use > 4 || Kd > K || K > 32 || K < 2) return LFS_E_INVALID;
    if (n == 0) return LFS_OK;
    if (!means || !viewmat || !sh0 || !shN || !radii || !colors || !v_colors || !v_sh0 || !v_means || !shN_exp_avg || !shN_exp_avg_sq) return LFS_E_INVALID;
    lfs::ShArgs a{};
    a.n = n; a.K = K; a.degree = int(degrees_to_use); a.means = means; a.viewmat = viewmat; a.sh0 = sh0; a.shN = shN; a.radii = radii; a.colors = colors;
    const lfs::ShAdam adam{shN_exp_avg, shN_exp_avg_sq, lfs::AdamScalars{lr, beta1, beta2, eps, bias_correction1_rcp, bias_correction2_sqrt_rcp}};
    return lfs::sh_launch_bwd_adam(a, v_colors, v_sh0, v_means, adam, (hipStream_t)stream);
}

// The all-inline training step (one view, one rank): dL/dcolour is read from the rasterizer's accumulator rows (acc_rows + 13, stride 16 floats),
// dL/d(dirs) is WRITTEN to v_dirs [n,3] (raster_finish_adam_kernel adds it to the means gradient), and BOTH coefficient tensors
// are updated in place by their Adam steps: no gradient tensor of the spherical harmonics exists.
int lfs::sh_model_bwd_adam_all_impl(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* means, const float* viewmat, float* sh0, float* shN,
    const int32_t* radii, const float* colors, const float* acc_rows, float* v_dirs,
    float* sh0_exp_avg, float* sh0_exp_avg_sq, const float* sh0_scalars /* lr, beta1, beta2, eps, bc1_rcp, bc2_sqrt_rcp */,
    float* shN_exp_avg, float* shN_exp_avg_sq, const float* shN_scalars, hipStream_t stream, const int32_t* abort_flag) {
    const uint32_t Kd = (degrees_to_use + 1) * (degrees_to_use + 1);
    if (degrees_to_use > 4 || Kd > K || K > 32 || K < 2) return LFS_E_INVALID;
    if (n == 0) return LFS_OK;
    if (!means || !viewmat || !sh0 || !shN || !radii || !colors || !acc_rows || !v_dirs || !sh0_exp_avg || !sh0_exp_avg_sq || !sh0_scalars || !shN_exp_avg ||
        !shN_exp_avg_sq || !shN_scalars) return LFS_E_INVALID;
    lfs::ShArgs a{};
    a.n = n; a.K = K; a.degree = int(degrees_to_use); a.means = means; a.viewmat = viewmat; a.sh0 = sh0; a.shN = shN; a.radii = radii; a.colors = colors;
    a.abort_flag = abort_flag;
    a.vs = 16; a.dirs_store = true; // dL/dcolour: slots 13..15 of the 16-float rows; dL/d(dirs): its own contiguous [n,3] (partial-row writes into the rows cost more than they save)
    lfs::ShAdam adam{shN_exp_avg, shN_exp_avg_sq, lfs::AdamScalars{shN_scalars[0], shN_scalars[1], shN_scalars[2], shN_scalars[3], shN_scalars[4], shN_scalars[5]}};
    adam.m0 = sh0_exp_avg; adam.v0 = sh0_exp_avg_sq;
    adam.s0 = lfs::AdamScalars{sh0_scalars[0], sh0_scalars[1], sh0_scalars[2], sh0_scalars[3], sh0_scalars[4], sh0_scalars[5]};
    return lfs::sh_launch_bwd_adam(a, acc_rows + 13, nullptr, v_dirs, adam, stream);
}

extern "C" int lfs_sh_model_bwd_adam_all(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* means, const float* viewmat, float* sh0, float* shN,
    const int32_t* radii, const float* colors, const float* acc_rows, float* v_dirs,
    float* sh0_exp_avg, float* sh0_exp_avg_sq, const float* sh0_scalars /* lr, beta1, beta2, eps, bc1_rcp, bc2_sqrt_rcp */,
    float* shN_exp_avg, float* shN_exp_avg_sq, const float* shN_scalars, lfs_stream_t stream) {
    return lfs::sh_model_bwd_adam_all_impl(n, K, degrees_to_use, means, viewmat, sh0, shN, radii, colors, acc_rows, v_dirs, sh0_exp_avg, sh0_exp_avg_sq, sh0_scalars,
                                           shN_exp_avg, shN_exp_avg_sq, shN_scalars, (hipStream_t)stream, nullptr);
}

// The two halves of lfs_sh_model_bwd_adam_all for the pipelined training step (kernels above). pipe_dirs: main stream, before the finish pass; pipe_adam: side stream.
int lfs::sh_pipe_dirs_impl(uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* means, const float* viewmat, const float* shN, const int32_t* radii,
                           const float* colors, const float* acc_rows, float* v_dirs, void* handover, const int32_t* abort_flag, int32_t* abort_snapshot, hipStream_t s) {
    const uint32_t Kd = (degrees_to_use + 1) * (degrees_to_use + 1);
    if (degrees_to_use > 4 || Kd > K || K > 32 || K < 2) return LFS_E_INVALID;
    if (n == 0) return LFS_OK;
    if (!means || !viewmat || !shN || !radii || !colors || !acc_rows || !v_dirs || !handover || !abort_flag || !abort_snapshot) return LFS_E_INVALID;
    lfs::ShArgs a{};
    a.n = n; a.K = K; a.degree = int(degrees_to_use); a.means = means; a.viewmat = viewmat; a.shN = shN; a.radii = radii; a.colors = colors;
    const dim3 grid((n + 63) / 64), block(64);
    lfs::ProfScope prof("sh_bwd_dirs", s);
    float4* h = static_cast<float4*>(handover);
    switch (lfs::lanes_for(K)) {
    case 4: hipLaunchKernelGGL((lfs::sh_pipe_dirs_kernel<4>), grid, block, 0, s, a, acc_rows + 13, v_dirs, h, abort_flag, abort_snapshot); break;
    case 16: hipLaunchKernelGGL((lfs::sh_pipe_dirs_kernel<16>), grid, block, 0, s, a, acc_rows + 13, v_dirs, h, abort_flag, abort_snapshot); break;
    default: hipLaunchKernelGGL((lfs::sh_pipe_dirs_kernel<32>), grid, block, 0, s, a, acc_rows + 13, v_dirs, h, abort_flag, abort_snapshot); break;
    }
    return (int)hipGetLastError();
}

int lfs::sh_pipe_adam_impl(uint32_t n, uint32_t K, uint32_t degrees_to_use, float* sh0, float* shN, const void* handover, float* sh0_exp_avg, float* sh0_exp_avg_sq,
                           const float* sh0_scalars, float* shN_exp_avg, float* shN_exp_avg_sq, const float* shN_scalars, const int32_t* abort_snapshot, hipStream_t s) {
    const uint32_t Kd = (degrees_to_use + 1) * (degrees_to_use + 1);
    if (degrees_to_use > 4 || Kd > K || K > 32 || K < 2) return LFS_E_INVALID;
    if (n == 0) return LFS_OK;
    if (!sh0 || !shN || !handover || !sh0_exp_avg || !sh0_exp_avg_sq || !sh0_scalars || !shN_exp_avg || !shN_exp_avg_sq || !shN_scalars || !abort_snapshot) return LFS_E_INVALID;
    lfs::ShAdam adam{shN_exp_avg, shN_exp_avg_sq, lfs::AdamScalars{shN_scalars[0], shN_scalars[1], shN_scalars[2], shN_scalars[3], shN_scalars[4], shN_scalars[5]}};
    adam.m0 = sh0_exp_avg; adam.v0 = sh0_exp_avg_sq;
    adam.s0 = lfs::AdamScalars{sh0_scalars[0], sh0_scalars[1], sh0_scalars[2], sh0_scalars[3], sh0_scalars[4], sh0_scalars[5]};
    // a fixed number of wavefronts (lfs::sh_pipe_side_waves() per CU x 256 CUs), not one per block: see sh_fwd_persistent_kernel
    const dim3 grid(std::min<uint32_t>((n + 63) / 64, lfs::sh_pipe_side_grid())), block(64);
    lfs::ProfScope prof("sh_bwd_adam", s);
    const float4* h = static_cast<const float4*>(handover);
    switch (lfs::lanes_for(K)) {
    case 4: hipLaunchKernelGGL((lfs::sh_pipe_adam_kernel<4>), grid, block, 0, s, n, K, int(degrees_to_use), sh0, shN, h, adam, abort_snapshot); break;
    case 16: hipLaunchKernelGGL((lfs::sh_pipe_adam_kernel<16>), grid, block, 0, s, n, K, int(degrees_to_use), sh0, shN, h, adam, abort_snapshot); break;
    default: hipLaunchKernelGGL((lfs::sh_pipe_adam_kernel<32>), grid, block, 0, s, n, K, int(degrees_to_use), sh0, shN, h, adam, abort_snapshot); break;
    }
    return (int)hipGetLastError();
}

// lfs_sh_model_fwd for the pipelined step: visible Gaussians only (radii), colours to `colors` [n,3] AND into the rgb slots of the rasterizer's records
int lfs::sh_model_fwd_records_impl(uint32_t n, uint32_t K, uint32_t degrees_to_use, const float* means, const float* viewmat, const float* sh0, const float* shN,
                                   const int32_t* radii, float* colors, float* rec_rgb, uint32_t rec_stride, hipStream_t stream) {
    const uint32_t Kd = (degrees_to_use + 1) * (degrees_to_use + 1);
    if (degrees_to_use > 4 || Kd > K || K > 32) return LFS_E_INVALID;
    if (n == 0) return LFS_OK;
    if (!means || !viewmat || !sh0 || (K > 1 && !shN) || !colors || !radii || !rec_rgb) return LFS_E_INVALID;
    lfs::ShArgs a{};
    a.n = n; a.K = K; a.degree = int(degrees_to_use); a.means = means; a.viewmat = viewmat; a.sh0 = sh0; a.shN = shN; a.radii = radii;
    a.rec_rgb = rec_rgb; a.rec_stride = rec_stride;
    return lfs::sh_launch_fwd<true>(a, Kd, colors, stream, lfs::sh_pipe_side_grid());
}

static bool sh_views_ok(uint32_t n, uint32_t K, uint32_t degree, uint32_t V, uint32_t stride) {
    const uint32_t Kd = (degree + 1) * (degree + 1);
    return degree <= 4 && Kd <= K && K <= 32 && V >= 1 && stride >= n;
}

extern "C" int lfs_sh_model_fwd_views(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, uint32_t n_views, uint32_t view_stride, const float* means, const float* viewmats,
    const float* sh0, const float* shN, const int32_t* radii, float* colors, lfs_stream_t stream) {
    if (n == 0 || n_views == 0) return LFS_OK;
    if (!sh_views_ok(n, K, degrees_to_use, n_views, view_stride)) return LFS_E_INVALID;
    if (!means || !viewmats || !sh0 || (K > 1 && !shN) || !radii || !colors) return LFS_E_INVALID;
    const lfs::ShViews a{n, K, int(degrees_to_use), n_views, view_stride, means, viewmats, sh0, shN, radii, nullptr};
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((n + 63) / 64), block(64);
    lfs::ProfScope prof("sh_fwd", s);
    switch (lfs::lanes_for((degrees_to_use + 1) * (degrees_to_use + 1))) {
    case 1: hipLaunchKernelGGL((lfs::sh_views_fwd_kernel<1>), grid, block, 0, s, a, colors); break;
    case 4: hipLaunchKernelGGL((lfs::sh_views_fwd_kernel<4>), grid, block, 0, s, a, colors); break;
    case 16: hipLaunchKernelGGL((lfs::sh_views_fwd_kernel<16>), grid, block, 0, s, a, colors); break;
    default: hipLaunchKernelGGL((lfs::sh_views_fwd_kernel<32>), grid, block, 0, s, a, colors); break;
    }
    return (int)hipGetLastError();
}

extern "C" int lfs_sh_model_bwd_views(
    uint32_t n, uint32_t K, uint32_t degrees_to_use, uint32_t n_views, uint32_t view_stride, const float* means, const float* viewmats,
    const float* sh0, float* shN, const int32_t* radii, const float* colors, const float* v_colors, int accumulate,
    float* v_sh0, float* v_shN, float* v_means, float* shN_exp_avg, float* shN_exp_avg_sq, float lr, float beta1, float beta2, float eps,
    float bias_correction1_rcp, float bias_correction2_sqrt_rcp, lfs_stream_t stream) {
    if (n == 0 || n_views == 0) return LFS_OK;
    if (!sh_views_ok(n, K, degrees_to_use, n_views, view_stride)) return LFS_E_INVALID;
    const bool inline_adam = shN_exp_avg != nullptr;
    // radii / colors may BOTH be NULL: v_colors rows pre-masked by their producer (see the kernel); v_shN may be NULL without inline Adam: gradient not wanted
    if (!means || !viewmats || !sh0 || (K > 1 && !shN) || ((radii == nullptr) != (colors == nullptr)) || !v_colors || !v_sh0 || !v_means) return LFS_E_INVALID;
    if (inline_adam && (accumulate || !shN_exp_avg_sq || K < 2)) return LFS_E_INVALID;
    const lfs::ShViews a{n, K, int(degrees_to_use), n_views, view_stride, means, viewmats, sh0, shN, radii, colors};
    const lfs::ShAdam adam{shN_exp_avg, shN_exp_avg_sq, lfs::AdamScalars{lr, beta1, beta2, eps, bias_correction1_rcp, bias_correction2_sqrt_rcp}};
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((n + 63) / 64), block(64);
    const size_t cof_bytes = size_t(64) * (K > 1 ? (K - 1) * 3 : 1) * sizeof(float);   // the wavefront's coefficient rows in LDS (11.5 KB at degree 3)
    lfs::ProfScope prof(inline_adam ? "sh_bwd_adam" : "sh_bwd", s);
#define LFS_SH_VIEWS_BWD(L)                                                                                                                  \
    if (inline_adam) hipLaunchKernelGGL((lfs::sh_views_bwd_kernel<L, true>), grid, block, cof_bytes, s, a, v_colors, accumulate, v_sh0, v_shN, v_means, adam); \
    else hipLaunchKernelGGL((lfs::sh_views_bwd_kernel<L, false>), grid, block, cof_bytes, s, a, v_colors, accumulate, v_sh0, v_shN, v_means, adam)
    switch (lfs::lanes_for(K)) {
    case 1: LFS_SH_VIEWS_BWD(1); break;
    case 4: LFS_SH_VIEWS_BWD(4); break;
    case 16: LFS_SH_VIEWS_BWD(16); break;
    default: LFS_SH_VIEWS_BWD(32); break;
    }
#undef LFS_SH_VIEWS_BWD
    return (int)hipGetLastError();
}
