// ORACLE/_ref - TEST INFRASTRUCTURE ONLY. gsplat/Common.h includes <glm/gtc/type_ptr.hpp> for its typedefs; the loader sources need only the enum below them.
#pragma once
namespace glm {
    template <int N, class T> struct vec {};
    template <int C, int R, class T> struct mat {};
} // namespace glm
