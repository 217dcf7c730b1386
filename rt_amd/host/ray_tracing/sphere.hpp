// sphere.hpp -- the one hitable primitive (reference: ray_tracing/sphere.hpp:12-33).
#pragma once

#include "hitable_object.hpp"

namespace ray_tracing
{
class Sphere : public HitableObject
{
public:
  Sphere(vec3 const & center, float radius, std::shared_ptr<Material> material)
    : HitableObject(std::move(material)), m_center(center), m_radius(radius)
  {}

  uint8_t GetType() const override { return kHitableObjectSphereType; }

  vec3 const GetCenter() const { return m_center; }
  float GetRadius() const { return m_radius; }

private:
  vec3 m_center;
  float m_radius = 1.0f;
};
}  // namespace ray_tracing
