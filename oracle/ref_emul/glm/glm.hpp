// ORACLE / TEST INFRASTRUCTURE ONLY - never part of the product.
//
// A small subset of GLM (g-truc/glm; the reference pins it through vcpkg, `vcpkg.json` builtin-baseline 4334d8b4..., GLM 1.0.1), restated from
// the published implementation so that the reference's OWN kernels (/root/reference/gsplat/*.cu, *.cuh, compiled in place as host code by
// oracle/Makefile `refk`) find the types and functions they use: vec<2|3|4>, mat<C,R> (column-major), qua (w,x,y,z constructor order), and
// dot / cross / length / normalize / transpose / inverse / outerProduct / quat_cast / mat3_cast / rotate / slerp / make_vec* / make_mat*.
// GLM itself is a third-party dependency that is not vendored under /root/reference, so it cannot be "compiled in place"; every function
// below follows GLM's formula AND operation order (glm/detail/func_geometric.inl, type_mat*.inl, type_quat.inl, gtc/quaternion.inl,
// ext/quaternion_common.inl), because fp32 results depend on it.
#pragma once
#include <cmath>
#include <cstddef>
#include <limits>

namespace glm {

typedef int length_t;
enum qualifier { defaultp };

template <length_t L, typename T, qualifier Q = defaultp> struct vec;
template <length_t C, length_t R, typename T, qualifier Q = defaultp> struct mat;
template <typename T, qualifier Q = defaultp> struct qua;

// ---------------------------------------------------------------------------------------------------------------- vec
template <typename T, qualifier Q> struct vec<2, T, Q> {
    T x, y;
    vec() = default;
    constexpr vec(T s) : x(s), y(s) {}
    template <typename A, typename B> constexpr vec(A a, B b) : x(T(a)), y(T(b)) {}
    template <typename U> constexpr vec(const vec<2, U, Q>& v) : x(T(v.x)), y(T(v.y)) {}
    T& operator[](length_t i) { return (&x)[i]; }
    constexpr const T& operator[](length_t i) const { return (&x)[i]; }
    vec& operator+=(const vec& o) { x += o.x; y += o.y; return *this; }
    vec& operator-=(const vec& o) { x -= o.x; y -= o.y; return *this; }
    vec& operator*=(T s) { x *= s; y *= s; return *this; }
    vec& operator*=(const vec& o) { x *= o.x; y *= o.y; return *this; }
    vec& operator/=(T s) { x /= s; y /= s; return *this; }
};
template <typename T, qualifier Q> struct vec<3, T, Q> {
    T x, y, z;
    vec() = default;
    constexpr vec(T s) : x(s), y(s), z(s) {}
    template <typename A, typename B, typename C> constexpr vec(A a, B b, C c) : x(T(a)), y(T(b)), z(T(c)) {}
    template <typename U> constexpr vec(const vec<3, U, Q>& v) : x(T(v.x)), y(T(v.y)), z(T(v.z)) {}
    template <typename C> constexpr vec(const vec<2, T, Q>& v, C c) : x(v.x), y(v.y), z(T(c)) {}
    constexpr vec(const vec<4, T, Q>& v);
    T& operator[](length_t i) { return (&x)[i]; }
    constexpr const T& operator[](length_t i) const { return (&x)[i]; }
    vec& operator+=(const vec& o) { x += o.x; y += o.y; z += o.z; return *this; }
    vec& operator-=(const vec& o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
    vec& operator*=(T s) { x *= s; y *= s; z *= s; return *this; }
    vec& operator*=(const vec& o) { x *= o.x; y *= o.y; z *= o.z; return *this; }
    vec& operator/=(T s) { x /= s; y /= s; z /= s; return *this; }
};
template <typename T, qualifier Q> struct vec<4, T, Q> {
    T x, y, z, w;
    vec() = default;
    constexpr vec(T s) : x(s), y(s), z(s), w(s) {}
    template <typename A, typename B, typename C, typename D> constexpr vec(A a, B b, C c, D d) : x(T(a)), y(T(b)), z(T(c)), w(T(d)) {}
    template <typename D> constexpr vec(const vec<3, T, Q>& v, D d) : x(v.x), y(v.y), z(v.z), w(T(d)) {}
    template <typename U> constexpr vec(const vec<4, U, Q>& v) : x(T(v.x)), y(T(v.y)), z(T(v.z)), w(T(v.w)) {}
    T& operator[](length_t i) { return (&x)[i]; }
    constexpr const T& operator[](length_t i) const { return (&x)[i]; }
    vec& operator+=(const vec& o) { x += o.x; y += o.y; z += o.z; w += o.w; return *this; }
    vec& operator-=(const vec& o) { x -= o.x; y -= o.y; z -= o.z; w -= o.w; return *this; }
    vec& operator*=(T s) { x *= s; y *= s; z *= s; w *= s; return *this; }
    vec& operator/=(T s) { x /= s; y /= s; z /= s; w /= s; return *this; }
};
template <typename T, qualifier Q> constexpr vec<3, T, Q>::vec(const vec<4, T, Q>& v) : x(v.x), y(v.y), z(v.z) {}

#define GLM_SUBSET_VEC_OPS(L, ...)                                                                                                        \
    template <typename T, qualifier Q> constexpr vec<L, T, Q> operator+(const vec<L, T, Q>& a, const vec<L, T, Q>& b) { return __VA_ARGS__(+); } \
    template <typename T, qualifier Q> constexpr vec<L, T, Q> operator-(const vec<L, T, Q>& a, const vec<L, T, Q>& b) { return __VA_ARGS__(-); } \
    template <typename T, qualifier Q> constexpr vec<L, T, Q> operator*(const vec<L, T, Q>& a, const vec<L, T, Q>& b) { return __VA_ARGS__(*); } \
    template <typename T, qualifier Q> constexpr vec<L, T, Q> operator/(const vec<L, T, Q>& a, const vec<L, T, Q>& b) { return __VA_ARGS__(/); }
#define GLM_V2(op) vec<2, T, Q>(a.x op b.x, a.y op b.y)
#define GLM_V3(op) vec<3, T, Q>(a.x op b.x, a.y op b.y, a.z op b.z)
#define GLM_V4(op) vec<4, T, Q>(a.x op b.x, a.y op b.y, a.z op b.z, a.w op b.w)
GLM_SUBSET_VEC_OPS(2, GLM_V2)
GLM_SUBSET_VEC_OPS(3, GLM_V3)
GLM_SUBSET_VEC_OPS(4, GLM_V4)
#undef GLM_V2
#undef GLM_V3
#undef GLM_V4
#undef GLM_SUBSET_VEC_OPS
// vec (op) scalar, scalar (op) vec, unary minus - component-wise, as in glm/detail/type_vec*.inl
template <length_t L, typename T, qualifier Q> constexpr vec<L, T, Q> operator*(const vec<L, T, Q>& a, T s) { return a * vec<L, T, Q>(s); }
template <length_t L, typename T, qualifier Q> constexpr vec<L, T, Q> operator*(T s, const vec<L, T, Q>& a) { return vec<L, T, Q>(s) * a; }
template <length_t L, typename T, qualifier Q> constexpr vec<L, T, Q> operator/(const vec<L, T, Q>& a, T s) { return a / vec<L, T, Q>(s); }
template <length_t L, typename T, qualifier Q> constexpr vec<L, T, Q> operator/(T s, const vec<L, T, Q>& a) { return vec<L, T, Q>(s) / a; }
template <length_t L, typename T, qualifier Q> constexpr vec<L, T, Q> operator+(const vec<L, T, Q>& a, T s) { return a + vec<L, T, Q>(s); }
template <length_t L, typename T, qualifier Q> constexpr vec<L, T, Q> operator+(T s, const vec<L, T, Q>& a) { return vec<L, T, Q>(s) + a; }
template <length_t L, typename T, qualifier Q> constexpr vec<L, T, Q> operator-(const vec<L, T, Q>& a, T s) { return a - vec<L, T, Q>(s); }
template <length_t L, typename T, qualifier Q> constexpr vec<L, T, Q> operator-(T s, const vec<L, T, Q>& a) { return vec<L, T, Q>(s) - a; }
template <typename T, qualifier Q> constexpr vec<2, T, Q> operator-(const vec<2, T, Q>& a) { return vec<2, T, Q>(-a.x, -a.y); }
template <typename T, qualifier Q> constexpr vec<3, T, Q> operator-(const vec<3, T, Q>& a) { return vec<3, T, Q>(-a.x, -a.y, -a.z); }
template <typename T, qualifier Q> constexpr vec<4, T, Q> operator-(const vec<4, T, Q>& a) { return vec<4, T, Q>(-a.x, -a.y, -a.z, -a.w); }
template <length_t L, typename T, qualifier Q> constexpr bool operator==(const vec<L, T, Q>& a, const vec<L, T, Q>& b) {
    for (length_t i = 0; i < L; ++i) if (a[i] != b[i]) return false;
    return true;
}

// geometric (glm/detail/func_geometric.inl: compute_dot / compute_cross / length = sqrt(dot) / normalize = v * inversesqrt(dot))
template <typename T, qualifier Q> constexpr T dot(const vec<2, T, Q>& a, const vec<2, T, Q>& b) { return a.x * b.x + a.y * b.y; }
template <typename T, qualifier Q> constexpr T dot(const vec<3, T, Q>& a, const vec<3, T, Q>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T, qualifier Q> constexpr T dot(const vec<4, T, Q>& a, const vec<4, T, Q>& b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }
template <typename T, qualifier Q> constexpr vec<3, T, Q> cross(const vec<3, T, Q>& x, const vec<3, T, Q>& y) {
    return vec<3, T, Q>(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
template <length_t L, typename T, qualifier Q> T length(const vec<L, T, Q>& v) { return std::sqrt(dot(v, v)); }
template <length_t L, typename T, qualifier Q> vec<L, T, Q> normalize(const vec<L, T, Q>& v) { return v * (T(1) / std::sqrt(dot(v, v))); }

// ---------------------------------------------------------------------------------------------------------------- mat (column-major: m[col][row])
template <length_t C, length_t R, typename T, qualifier Q> struct mat {
    typedef vec<R, T, Q> col_type;
    typedef vec<C, T, Q> row_type;
    col_type value[C];
    mat() = default;
    explicit constexpr mat(T s) : value{} { for (length_t i = 0; i < C; ++i) for (length_t j = 0; j < R; ++j) value[i][j] = (i == j) ? s : T(0); }
    // C == R == 2
    template <typename A0, typename A1, typename A2, typename A3>
    constexpr mat(A0 x0, A1 y0, A2 x1, A3 y1) : value{col_type(x0, y0), col_type(x1, y1)} { static_assert(C == 2 && R == 2, "mat2"); }
    // 3x3 (9 scalars, column by column)
    template <typename A0, typename A1, typename A2, typename A3, typename A4, typename A5, typename A6, typename A7, typename A8>
    constexpr mat(A0 x0, A1 y0, A2 z0, A3 x1, A4 y1, A5 z1, A6 x2, A7 y2, A8 z2) : value{col_type(x0, y0, z0), col_type(x1, y1, z1), col_type(x2, y2, z2)} {
        static_assert(C == 3 && R == 3, "mat3");
    }
    // 3x2 (6 scalars: three columns of two rows)
    template <typename A0, typename A1, typename A2, typename A3, typename A4, typename A5>
    constexpr mat(A0 x0, A1 y0, A2 x1, A3 y1, A4 x2, A5 y2) : value{col_type(x0, y0), col_type(x1, y1), col_type(x2, y2)} { static_assert(C == 3 && R == 2, "mat3x2"); }
    constexpr mat(const col_type& c0, const col_type& c1) : value{c0, c1} { static_assert(C == 2, "2 columns"); }
    constexpr mat(const col_type& c0, const col_type& c1, const col_type& c2) : value{c0, c1, c2} { static_assert(C == 3, "3 columns"); }
    constexpr mat(const col_type& c0, const col_type& c1, const col_type& c2, const col_type& c3) : value{c0, c1, c2, c3} { static_assert(C == 4, "4 columns"); }
    col_type& operator[](length_t i) { return value[i]; }
    constexpr const col_type& operator[](length_t i) const { return value[i]; }
    mat& operator+=(const mat& o) { for (length_t i = 0; i < C; ++i) value[i] += o.value[i]; return *this; }
    mat& operator-=(const mat& o) { for (length_t i = 0; i < C; ++i) value[i] -= o.value[i]; return *this; }
    mat& operator*=(T s) { for (length_t i = 0; i < C; ++i) value[i] *= s; return *this; }
};
template <length_t C, length_t R, typename T, qualifier Q> mat<C, R, T, Q> operator+(const mat<C, R, T, Q>& a, const mat<C, R, T, Q>& b) { mat<C, R, T, Q> r; for (length_t i = 0; i < C; ++i) r[i] = a[i] + b[i]; return r; }
template <length_t C, length_t R, typename T, qualifier Q> mat<C, R, T, Q> operator-(const mat<C, R, T, Q>& a, const mat<C, R, T, Q>& b) { mat<C, R, T, Q> r; for (length_t i = 0; i < C; ++i) r[i] = a[i] - b[i]; return r; }
template <length_t C, length_t R, typename T, qualifier Q> mat<C, R, T, Q> operator-(const mat<C, R, T, Q>& a) { mat<C, R, T, Q> r; for (length_t i = 0; i < C; ++i) r[i] = -a[i]; return r; }
template <length_t C, length_t R, typename T, qualifier Q> mat<C, R, T, Q> operator*(const mat<C, R, T, Q>& a, T s) { mat<C, R, T, Q> r; for (length_t i = 0; i < C; ++i) r[i] = a[i] * s; return r; }
template <length_t C, length_t R, typename T, qualifier Q> mat<C, R, T, Q> operator*(T s, const mat<C, R, T, Q>& a) { mat<C, R, T, Q> r; for (length_t i = 0; i < C; ++i) r[i] = a[i] * s; return r; }
template <length_t C, length_t R, typename T, qualifier Q> mat<C, R, T, Q> operator/(const mat<C, R, T, Q>& a, T s) { mat<C, R, T, Q> r; for (length_t i = 0; i < C; ++i) r[i] = a[i] / s; return r; }
// mat * column vector: sum over the columns in index order (type_mat3x3.inl: m[0][i]*v.x + m[1][i]*v.y + m[2][i]*v.z)
// (the vector is a non-deduced parameter, as GLM's `typename mat::row_type const&` is: a vec of another scalar type converts)
template <length_t C, length_t R, typename T, qualifier Q> vec<R, T, Q> operator*(const mat<C, R, T, Q>& m, const typename mat<C, R, T, Q>::row_type& v) {
    vec<R, T, Q> r;
    for (length_t i = 0; i < R; ++i) { T s = m[0][i] * v[0]; for (length_t k = 1; k < C; ++k) s = s + m[k][i] * v[k]; r[i] = s; }
    return r;
}
// row vector * mat: r[c] = dot(m[c], v)  (type_mat3x3.inl operator*(row_type, mat): m[c][0]*v.x + m[c][1]*v.y + m[c][2]*v.z)
template <length_t C, length_t R, typename T, qualifier Q> vec<C, T, Q> operator*(const vec<R, T, Q>& v, const mat<C, R, T, Q>& m) {
    vec<C, T, Q> r;
    for (length_t c = 0; c < C; ++c) { T s = m[c][0] * v[0]; for (length_t k = 1; k < R; ++k) s = s + m[c][k] * v[k]; r[c] = s; }
    return r;
}
// mat * mat: Result[j][i] = A[0][i]*B[j][0] + A[1][i]*B[j][1] + ... in index order (type_mat*.inl)
template <length_t K, length_t R, length_t C2, typename T, qualifier Q>
mat<C2, R, T, Q> operator*(const mat<K, R, T, Q>& a, const mat<C2, K, T, Q>& b) {
    mat<C2, R, T, Q> r;
    for (length_t j = 0; j < C2; ++j)
        for (length_t i = 0; i < R; ++i) { T s = a[0][i] * b[j][0]; for (length_t k = 1; k < K; ++k) s = s + a[k][i] * b[j][k]; r[j][i] = s; }
    return r;
}
template <length_t C, length_t R, typename T, qualifier Q> mat<R, C, T, Q> transpose(const mat<C, R, T, Q>& m) {
    mat<R, C, T, Q> r;
    for (length_t i = 0; i < C; ++i) for (length_t j = 0; j < R; ++j) r[j][i] = m[i][j];
    return r;
}
template <length_t C, length_t R, typename T, qualifier Q> mat<C, R, T, Q> outerProduct(const vec<R, T, Q>& c, const vec<C, T, Q>& r) {
    mat<C, R, T, Q> m;
    for (length_t i = 0; i < C; ++i) m[i] = c * r[i];
    return m;
}
template <typename T, qualifier Q> T determinant(const mat<2, 2, T, Q>& m) { return m[0][0] * m[1][1] - m[1][0] * m[0][1]; }
template <typename T, qualifier Q> T determinant(const mat<3, 3, T, Q>& m) {
    return +m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2]) - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2]) + m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]);
}
// func_matrix.inl compute_inverse<2,2> / <3,3>
template <typename T, qualifier Q> mat<2, 2, T, Q> inverse(const mat<2, 2, T, Q>& m) {
    const T OneOverDeterminant = T(1) / (+m[0][0] * m[1][1] - m[1][0] * m[0][1]);
    return mat<2, 2, T, Q>(+m[1][1] * OneOverDeterminant, -m[0][1] * OneOverDeterminant, -m[1][0] * OneOverDeterminant, +m[0][0] * OneOverDeterminant);
}
template <typename T, qualifier Q> mat<3, 3, T, Q> inverse(const mat<3, 3, T, Q>& m) {
    const T OneOverDeterminant = T(1) / determinant(m);
    mat<3, 3, T, Q> I;
    I[0][0] = +(m[1][1] * m[2][2] - m[2][1] * m[1][2]) * OneOverDeterminant;
    I[1][0] = -(m[1][0] * m[2][2] - m[2][0] * m[1][2]) * OneOverDeterminant;
    I[2][0] = +(m[1][0] * m[2][1] - m[2][0] * m[1][1]) * OneOverDeterminant;
    I[0][1] = -(m[0][1] * m[2][2] - m[2][1] * m[0][2]) * OneOverDeterminant;
    I[1][1] = +(m[0][0] * m[2][2] - m[2][0] * m[0][2]) * OneOverDeterminant;
    I[2][1] = -(m[0][0] * m[2][1] - m[2][0] * m[0][1]) * OneOverDeterminant;
    I[0][2] = +(m[0][1] * m[1][2] - m[1][1] * m[0][2]) * OneOverDeterminant;
    I[1][2] = -(m[0][0] * m[1][2] - m[1][0] * m[0][2]) * OneOverDeterminant;
    I[2][2] = +(m[0][0] * m[1][1] - m[1][0] * m[0][1]) * OneOverDeterminant;
    return I;
}

// ---------------------------------------------------------------------------------------------------------------- qua
// GLM 1.0 default layout: storage x, y, z, w; constructor argument order (w, x, y, z) (GLM_FORCE_QUAT_DATA_WXYZ is not set by the reference)
template <typename T, qualifier Q> struct qua {
    T x, y, z, w;
    qua() = default;
    constexpr qua(T w_, T x_, T y_, T z_) : x(x_), y(y_), z(z_), w(w_) {}
};
template <typename T, qualifier Q> constexpr qua<T, Q> operator-(const qua<T, Q>& q) { return qua<T, Q>(-q.w, -q.x, -q.y, -q.z); }
template <typename T, qualifier Q> constexpr qua<T, Q> operator+(const qua<T, Q>& a, const qua<T, Q>& b) { return qua<T, Q>(a.w + b.w, a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename T, qualifier Q> constexpr qua<T, Q> operator*(const qua<T, Q>& q, T s) { return qua<T, Q>(q.w * s, q.x * s, q.y * s, q.z * s); }
template <typename T, qualifier Q> constexpr qua<T, Q> operator*(T s, const qua<T, Q>& q) { return q * s; }
template <typename T, qualifier Q> constexpr qua<T, Q> operator/(const qua<T, Q>& q, T s) { return qua<T, Q>(q.w / s, q.x / s, q.y / s, q.z / s); }
// compute_dot<qua>: vec4 tmp(a.w*b.w, a.x*b.x, a.y*b.y, a.z*b.z); (tmp.x + tmp.y) + (tmp.z + tmp.w)
template <typename T, qualifier Q> constexpr T dot(const qua<T, Q>& a, const qua<T, Q>& b) { return (a.w * b.w + a.x * b.x) + (a.y * b.y + a.z * b.z); }
template <typename T, qualifier Q> T length(const qua<T, Q>& q) { return std::sqrt(dot(q, q)); }
template <typename T, qualifier Q> qua<T, Q> normalize(const qua<T, Q>& q) { // ext/quaternion_geometric.inl
    const T len = length(q);
    if (len <= T(0)) return qua<T, Q>(T(1), T(0), T(0), T(0));
    const T oneOverLen = T(1) / len;
    return qua<T, Q>(q.w * oneOverLen, q.x * oneOverLen, q.y * oneOverLen, q.z * oneOverLen);
}
template <typename T, qualifier Q> constexpr qua<T, Q> conjugate(const qua<T, Q>& q) { return qua<T, Q>(q.w, -q.x, -q.y, -q.z); }
template <typename T, qualifier Q> constexpr qua<T, Q> inverse(const qua<T, Q>& q) { return conjugate(q) / dot(q, q); }
// type_quat.inl operator*(qua, vec3): uv = cross(QuatVector, v); uuv = cross(QuatVector, uv); v + ((uv * q.w) + uuv) * 2
template <typename T, qualifier Q> vec<3, T, Q> operator*(const qua<T, Q>& q, const vec<3, T, Q>& v) {
    const vec<3, T, Q> QuatVector(q.x, q.y, q.z);
    const vec<3, T, Q> uv(cross(QuatVector, v));
    const vec<3, T, Q> uuv(cross(QuatVector, uv));
    return v + ((uv * q.w) + uuv) * static_cast<T>(2);
}
template <typename T, qualifier Q> vec<3, T, Q> rotate(const qua<T, Q>& q, const vec<3, T, Q>& v) { return q * v; } // gtx/quaternion.inl
// gtc/quaternion.inl mat3_cast
template <typename T, qualifier Q> mat<3, 3, T, Q> mat3_cast(const qua<T, Q>& q) {
    mat<3, 3, T, Q> Result(T(1));
    const T qxx(q.x * q.x), qyy(q.y * q.y), qzz(q.z * q.z), qxz(q.x * q.z), qxy(q.x * q.y), qyz(q.y * q.z), qwx(q.w * q.x), qwy(q.w * q.y), qwz(q.w * q.z);
    Result[0][0] = T(1) - T(2) * (qyy + qzz);
    Result[0][1] = T(2) * (qxy + qwz);
    Result[0][2] = T(2) * (qxz - qwy);
    Result[1][0] = T(2) * (qxy - qwz);
    Result[1][1] = T(1) - T(2) * (qxx + qzz);
    Result[1][2] = T(2) * (qyz + qwx);
    Result[2][0] = T(2) * (qxz + qwy);
    Result[2][1] = T(2) * (qyz - qwx);
    Result[2][2] = T(1) - T(2) * (qxx + qyy);
    return Result;
}
// gtc/quaternion.inl quat_cast(mat3)
template <typename T, qualifier Q> qua<T, Q> quat_cast(const mat<3, 3, T, Q>& m) {
    const T fourXSquaredMinus1 = m[0][0] - m[1][1] - m[2][2];
    const T fourYSquaredMinus1 = m[1][1] - m[0][0] - m[2][2];
    const T fourZSquaredMinus1 = m[2][2] - m[0][0] - m[1][1];
    const T fourWSquaredMinus1 = m[0][0] + m[1][1] + m[2][2];
    int biggestIndex = 0;
    T fourBiggestSquaredMinus1 = fourWSquaredMinus1;
    if (fourXSquaredMinus1 > fourBiggestSquaredMinus1) { fourBiggestSquaredMinus1 = fourXSquaredMinus1; biggestIndex = 1; }
    if (fourYSquaredMinus1 > fourBiggestSquaredMinus1) { fourBiggestSquaredMinus1 = fourYSquaredMinus1; biggestIndex = 2; }
    if (fourZSquaredMinus1 > fourBiggestSquaredMinus1) { fourBiggestSquaredMinus1 = fourZSquaredMinus1; biggestIndex = 3; }
    const T biggestVal = std::sqrt(fourBiggestSquaredMinus1 + T(1)) * T(0.5);
    const T mult = T(0.25) / biggestVal;
    switch (biggestIndex) {
    case 0: return qua<T, Q>(biggestVal, (m[1][2] - m[2][1]) * mult, (m[2][0] - m[0][2]) * mult, (m[0][1] - m[1][0]) * mult);
    case 1: return qua<T, Q>((m[1][2] - m[2][1]) * mult, biggestVal, (m[0][1] + m[1][0]) * mult, (m[2][0] + m[0][2]) * mult);
    case 2: return qua<T, Q>((m[2][0] - m[0][2]) * mult, (m[0][1] + m[1][0]) * mult, biggestVal, (m[1][2] + m[2][1]) * mult);
    default: return qua<T, Q>((m[0][1] - m[1][0]) * mult, (m[2][0] + m[0][2]) * mult, (m[1][2] + m[2][1]) * mult, biggestVal);
    }
}
template <typename T> constexpr T mix(T x, T y, T a) { return x * (T(1) - a) + y * a; } // func_common.inl compute_mix
// ext/quaternion_common.inl slerp
template <typename T, qualifier Q> qua<T, Q> slerp(const qua<T, Q>& x, const qua<T, Q>& y, T a) {
    qua<T, Q> z = y;
    T cosTheta = dot(x, y);
    if (cosTheta < T(0)) { z = -y; cosTheta = -cosTheta; }
    if (cosTheta > T(1) - std::numeric_limits<T>::epsilon())
        return qua<T, Q>(mix(x.w, z.w, a), mix(x.x, z.x, a), mix(x.y, z.y, a), mix(x.z, z.z, a));
    const T angle = std::acos(cosTheta);
    return (std::sin((T(1) - a) * angle) * x + std::sin(a * angle) * z) / std::sin(angle);
}

// gtc/type_ptr.inl
template <typename T> vec<2, T, defaultp> make_vec2(const T* p) { return vec<2, T, defaultp>(p[0], p[1]); }
template <typename T> vec<3, T, defaultp> make_vec3(const T* p) { return vec<3, T, defaultp>(p[0], p[1], p[2]); }
template <typename T> vec<4, T, defaultp> make_vec4(const T* p) { return vec<4, T, defaultp>(p[0], p[1], p[2], p[3]); }
template <typename T> mat<3, 3, T, defaultp> make_mat3(const T* p) { return mat<3, 3, T, defaultp>(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]); }
template <length_t L, typename T, qualifier Q> const T* value_ptr(const vec<L, T, Q>& v) { return &v.x; }
template <length_t L, typename T, qualifier Q> T* value_ptr(vec<L, T, Q>& v) { return &v.x; }

typedef vec<2, float> vec2; typedef vec<3, float> vec3; typedef vec<4, float> vec4;
typedef vec<2, float> fvec2; typedef vec<3, float> fvec3; typedef vec<4, float> fvec4;
typedef mat<2, 2, float> mat2; typedef mat<3, 3, float> mat3; typedef mat<4, 4, float> mat4;
typedef mat<2, 2, float> fmat2; typedef mat<3, 3, float> fmat3; typedef mat<4, 4, float> fmat4;
typedef mat<3, 2, float> mat3x2;
typedef qua<float> quat; typedef qua<float> fquat;

} // namespace glm
