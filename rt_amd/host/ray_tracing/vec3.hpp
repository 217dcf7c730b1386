// vec3.hpp -- the three-float vector the host API is written in.  The reference spells it glm::vec3
// (an external dependency this tree does not vendor); only the layout -- three packed floats, which the
// reference also relies on when it hands a ColorBuffer to the GPU as float* (rt_cuda/demo_frame_cuda.cpp:39)
// -- and the few operations the scene builders use are needed here.
#pragma once

#include <cmath>

namespace ray_tracing
{
struct vec3
{
  union { float x; float r; };
  union { float y; float g; };
  union { float z; float b; };

  constexpr vec3() : x(0.0f), y(0.0f), z(0.0f) {}
  constexpr vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
  constexpr explicit vec3(float s) : x(s), y(s), z(s) {}
};
static_assert(sizeof(vec3) == 3 * sizeof(float), "vec3 must be three packed floats");

inline vec3 operator+(vec3 const & a, vec3 const & b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 const & a, vec3 const & b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator-(vec3 const & a) { return {-a.x, -a.y, -a.z}; }
inline vec3 operator*(vec3 const & a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 const & a) { return a * s; }

inline float dot(vec3 const & a, vec3 const & b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// v * inversesqrt(dot(v, v)), the GLSL definition glm::normalize follows.
inline vec3 normalize(vec3 const & v) { return v * (1.0f / std::sqrt(dot(v, v))); }
// x * (1 - a) + y * a, the GLSL definition glm::mix follows.
inline float mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }
}  // namespace ray_tracing
