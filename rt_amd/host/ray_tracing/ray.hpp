// ray.hpp -- origin + direction pair (reference: ray_tracing/ray.hpp:7-24).
#pragma once

#include "vec3.hpp"

namespace ray_tracing
{
class Ray
{
public:
  Ray() = default;
  Ray(vec3 const & origin, vec3 const & direction) : m_origin(origin), m_direction(direction) {}

  vec3 const & Origin() const { return m_origin; }
  vec3 const & Direction() const { return m_direction; }
  vec3 PointAt(float t) const { return m_origin + m_direction * t; }

private:
  vec3 m_origin;
  vec3 m_direction;
};
}  // namespace ray_tracing
