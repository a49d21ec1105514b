// ORACLE/_ref - TEST INFRASTRUCTURE ONLY. include/core/splat_data.hpp names glm::mat4 in one declaration (SplatData::transform, not compiled here).
#pragma once
namespace glm { struct mat4 {}; struct vec3 {}; }
