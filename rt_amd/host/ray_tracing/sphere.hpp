// sphere.hpp -- the one hitable primitive (reference: ray_tracing/sphere.hpp:12-33, sphere.cpp:11-81).
#pragma once

#include "hitable_object.hpp"

#include <cmath>

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

  // Both roots of |O + tD - C|^2 = r^2 that fall into [tmin, tmax], smaller first; a discriminant within kEps of
  // zero counts as one tangent hit at -b / 2a (reference: sphere.cpp:11-61).
  std::vector<Hit> Trace(Ray const & ray, float tmin, float tmax) const override
  {
    Quadratic const q = Solve(ray);
    if (q.discriminant < 0.0f)
      return {};
    float t[2];
    int count = 0;
    if (std::fabs(q.discriminant) < kEps)
    {
      float const tangent = -q.b / (2.0f * q.a);
      if (!(tangent < tmin || tangent > tmax))
        t[count++] = tangent;
    }
    else
    {
      float const root = std::sqrt(q.discriminant);
      float const t1 = (-q.b - root) / (2.0f * q.a);
      float const t2 = (-q.b + root) / (2.0f * q.a);
      bool const in1 = t1 >= tmin && t1 <= tmax;
      bool const in2 = t2 >= tmin && t2 <= tmax;
      if (in1 && in2)
      {
        t[count++] = std::min(t1, t2);
        t[count++] = std::max(t1, t2);
      }
      else if (in1)
        t[count++] = t1;
      else if (in2)
        t[count++] = t2;
    }
    std::vector<Hit> hits;
    for (int i = 0; i < count; ++i)
      hits.push_back(HitAt(ray, t[i]));
    return hits;
  }

  // The smaller root only: a ray that starts on or inside the sphere (smaller root below tmin) misses it, the
  // larger root is never tried (reference: sphere.cpp:63-81).
  std::optional<Hit> TraceNearest(Ray const & ray, float tmin, float tmax) const override
  {
    Quadratic const q = Solve(ray);
    if (q.discriminant < 0.0f)
      return {};
    float const root = std::sqrt(q.discriminant);
    float const t = std::min((-q.b - root) / (2.0f * q.a), (-q.b + root) / (2.0f * q.a));
    if (t < tmin || t > tmax)
      return {};
    return HitAt(ray, t);
  }

private:
  struct Quadratic
  {
    float a, b, discriminant;
  };

  Quadratic Solve(Ray const & ray) const
  {
    vec3 const d = ray.Origin() - m_center;
    Quadratic q;
    q.a = dot(ray.Direction(), ray.Direction());
    q.b = 2.0f * dot(d, ray.Direction());
    float const c = dot(d, d) - m_radius * m_radius;
    q.discriminant = q.b * q.b - 4 * q.a * c;
    return q;
  }

  Hit HitAt(Ray const & ray, float t) const
  {
    vec3 const p = ray.PointAt(t);
    return Hit(t, p, normalize(p - m_center), m_material);
  }

  vec3 m_center;
  float m_radius = 1.0f;
};
}  // namespace ray_tracing
