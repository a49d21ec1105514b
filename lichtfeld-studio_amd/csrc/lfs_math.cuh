// Device-side small-vector / quaternion arithmetic for the gfx950 kernels.
// Matrices are row-major with math indexing m[r][c]. The formulas restate what
// the reference takes from glm (quat_cast / rotate / slerp / mat3_cast) and
// gsplat/Utils.cuh (quat_to_rotmat and its vjp, safe_normalize) so that results
// agree with the CUDA path to fp32 rounding; the code itself is written for
// wave64 CDNA4 (no glm, no cooperative groups).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lfs {

struct f3 { float x, y, z; };
struct f2 { float x, y; };
// row i of a packed [n,3] float array as ONE 12-byte access (global_load / store_dwordx3): written element by element the compiler emits three
// single-dword accesses at a 12-byte lane stride
struct alignas(4) V3f { float a[3]; };
struct quat { float w, x, y, z; };   // reference order (w, x, y, z)
struct m3 { float m[3][3]; };

#define LFS_DI __device__ __forceinline__
// the workgroup's dynamic LDS block as an array `name` of `type` (the host emulator of tests/emul hands out a heap block instead)
// LFS_WAVE_LOCKSTEP(): a point where the code relies on the lanes of a wavefront executing in lock-step with an in-order LDS pipeline (nothing to
// emit on the GPU; the emulator's lanes are independent fibers and meet here)
#ifdef LFS_EMULATE
#define LFS_DYN_LDS(type, name) type* name = reinterpret_cast<type*>(emu::dyn_lds())
#define LFS_WAVE_LOCKSTEP() ((void)emu::ballot(true))
#define LFS_SYSTEM_FENCE() ((void)0)
#else
#define LFS_SYSTEM_FENCE() __threadfence_system()
#define LFS_DYN_LDS(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
#define LFS_WAVE_LOCKSTEP() ((void)0)
#endif

LFS_DI f3 ld3(const float* __restrict__ base, size_t i) { const V3f t = reinterpret_cast<const V3f*>(base)[i]; return {t.a[0], t.a[1], t.a[2]}; }
LFS_DI void st3(float* __restrict__ base, size_t i, f3 v) { V3f t; t.a[0] = v.x; t.a[1] = v.y; t.a[2] = v.z; reinterpret_cast<V3f*>(base)[i] = t; }
LFS_DI f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
LFS_DI f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
LFS_DI f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
LFS_DI f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
LFS_DI f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
LFS_DI float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
LFS_DI f3 cross(f3 a, f3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
LFS_DI f3 mul(const m3& A, f3 v) {
    return {A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z,
            A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
            A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z};
}
LFS_DI f3 mul_t(const m3& A, f3 v) { // A^T v
    return {A.m[0][0] * v.x + A.m[1][0] * v.y + A.m[2][0] * v.z,
            A.m[0][1] * v.x + A.m[1][1] * v.y + A.m[2][1] * v.z,
            A.m[0][2] * v.x + A.m[1][2] * v.y + A.m[2][2] * v.z};
}

LFS_DI float qdot(quat a, quat b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
LFS_DI quat qnormalize(quat q) {
    float len = sqrtf(qdot(q, q));
    if (len <= 0.f) return {1.f, 0.f, 0.f, 0.f};
    float inv = 1.f / len;
    return {q.w * inv, q.x * inv, q.y * inv, q.z * inv};
}
LFS_DI quat qinverse(quat q) {
    float d = qdot(q, q);
    return {q.w / d, -q.x / d, -q.y / d, -q.z / d};
}
LFS_DI f3 qrotate(quat q, f3 v) {
    f3 qv{q.x, q.y, q.z};
    f3 uv = cross(qv, v);
    f3 uuv = cross(qv, uv);
    return v + ((uv * q.w) + uuv) * 2.f;
}
LFS_DI m3 qmat3(quat q) {
    float qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z;
    float qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z;
    float qwx = q.w * q.x, qwy = q.w * q.y, qwz = q.w * q.z;
    m3 R;
    R.m[0][0] = 1.f - 2.f * (qyy + qzz); R.m[1][0] = 2.f * (qxy + qwz); R.m[2][0] = 2.f * (qxz - qwy);
    R.m[0][1] = 2.f * (qxy - qwz); R.m[1][1] = 1.f - 2.f * (qxx + qzz); R.m[2][1] = 2.f * (qyz + qwx);
    R.m[0][2] = 2.f * (qxz + qwy); R.m[1][2] = 2.f * (qyz - qwx); R.m[2][2] = 1.f - 2.f * (qxx + qyy);
    return R;
}
// rotation block of a row-major [4,4] world->camera matrix -> quaternion
LFS_DI quat qcast_viewmat(const float* __restrict__ vm) {
    const float r00 = vm[0], r01 = vm[1], r02 = vm[2];
    const float r10 = vm[4], r11 = vm[5], r12 = vm[6];
    const float r20 = vm[8], r21 = vm[9], r22 = vm[10];
    float fx = r00 - r11 - r22, fy = r11 - r00 - r22, fz = r22 - r00 - r11, fw = r00 + r11 + r22;
    int big = 0; float fb = fw;
    if (fx > fb) { fb = fx; big = 1; }
    if (fy > fb) { fb = fy; big = 2; }
    if (fz > fb) { fb = fz; big = 3; }
    float bv = sqrtf(fb + 1.f) * 0.5f;
    float mult = 0.25f / bv;
    if (big == 0) return {bv, (r21 - r12) * mult, (r02 - r20) * mult, (r10 - r01) * mult};
    if (big == 1) return {(r21 - r12) * mult, bv, (r10 + r01) * mult, (r02 + r20) * mult};
    if (big == 2) return {(r02 - r20) * mult, (r10 + r01) * mult, bv, (r21 + r12) * mult};
    return {(r10 - r01) * mult, (r02 + r20) * mult, (r21 + r12) * mult, bv};
}
LFS_DI quat qslerp(quat x, quat y, float a) {
    quat z = y;
    float c = qdot(x, y);
    if (c < 0.f) { z = {-y.w, -y.x, -y.y, -y.z}; c = -c; }
    if (c > 1.f - 1.1920928955078125e-07f) {
        float b = 1.f - a;
        return {x.w * b + z.w * a, x.x * b + z.x * a, x.y * b + z.y * a, x.z * b + z.z * a};
    }
    float ang = acosf(c);
    float s0 = sinf((1.f - a) * ang), s1 = sinf(a * ang), sd = sinf(ang);
    return {(s0 * x.w + s1 * z.w) / sd, (s0 * x.x + s1 * z.x) / sd, (s0 * x.y + s1 * z.y) / sd, (s0 * x.z + s1 * z.z) / sd};
}

// normalising quat(w,x,y,z) -> R. cap > 0 clamps the inverse norm (add_noise).
LFS_DI m3 quat_to_rotmat(float w, float x, float y, float z, float cap = 0.f) {
    float inv = 1.f / sqrtf(x * x + y * y + z * z + w * w);
    if (cap > 0.f && !(inv < cap)) inv = cap;
    x *= inv; y *= inv; z *= inv; w *= inv;
    float x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    m3 R;
    R.m[0][0] = 1.f - 2.f * (y2 + z2); R.m[1][0] = 2.f * (xy + wz); R.m[2][0] = 2.f * (xz - wy);
    R.m[0][1] = 2.f * (xy - wz); R.m[1][1] = 1.f - 2.f * (x2 + z2); R.m[2][1] = 2.f * (yz + wx);
    R.m[0][2] = 2.f * (xz + wy); R.m[1][2] = 2.f * (yz - wx); R.m[2][2] = 1.f - 2.f * (x2 + y2);
    return R;
}

// dL/dquat (un-normalised) from G = dL/dR, accumulating into vq[4] = (w,x,y,z)
LFS_DI void quat_to_rotmat_vjp(float w, float x, float y, float z, const m3& G, float* vq) {
    float inv = 1.f / sqrtf(x * x + y * y + z * z + w * w);
    x *= inv; y *= inv; z *= inv; w *= inv;
    const auto& g = G.m;
    float vw = 2.f * (x * (g[2][1] - g[1][2]) + y * (g[0][2] - g[2][0]) + z * (g[1][0] - g[0][1]));
    float vx = 2.f * (-2.f * x * (g[1][1] + g[2][2]) + y * (g[1][0] + g[0][1]) + z * (g[2][0] + g[0][2]) + w * (g[2][1] - g[1][2]));
    float vy = 2.f * (x * (g[1][0] + g[0][1]) - 2.f * y * (g[0][0] + g[2][2]) + z * (g[2][1] + g[1][2]) + w * (g[0][2] - g[2][0]));
    float vz = 2.f * (x * (g[2][0] + g[0][2]) + y * (g[2][1] + g[1][2]) - 2.f * z * (g[0][0] + g[1][1]) + w * (g[1][0] - g[0][1]));
    float d = vw * w + vx * x + vy * y + vz * z;
    vq[0] += (vw - d * w) * inv;
    vq[1] += (vx - d * x) * inv;
    vq[2] += (vy - d * y) * inv;
    vq[3] += (vz - d * z) * inv;
}

LFS_DI f3 safe_normalize(f3 v) {
    float l = v.x * v.x + v.y * v.y + v.z * v.z;
    return l > 0.f ? v * (1.f / sqrtf(l)) : v;
}
LFS_DI f3 safe_normalize_bw(f3 v, f3 d_out) {
    float l = v.x * v.x + v.y * v.y + v.z * v.z;
    if (l > 0.f) {
        float il = 1.f / sqrtf(l);
        float il3 = il * il * il;
        return il * d_out - (il3 * dot(d_out, v)) * v;
    }
    return d_out;
}

} // namespace lfs
