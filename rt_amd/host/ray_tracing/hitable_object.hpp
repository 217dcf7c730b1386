// hitable_object.hpp -- base of the scene objects handed to Frame::AddObject, the Hit record and the two
// collection queries (reference: ray_tracing/hitable_object.hpp:13-87).
//
// The ray queries here are plain host geometry on the object model -- the reference declares them on
// HitableObject and so does this mirror -- for callers that want to ask a scene a question (picking, tests).
// They are not a renderer: frames are traced by the HIP kernel, which restates the same arithmetic on the device.
#pragma once

#include "material.hpp"
#include "ray.hpp"

#include <algorithm>
#include <cstdint>
#include <memory>
#include <optional>
#include <vector>

namespace ray_tracing
{
// One intersection of a ray with an object (reference: hitable_object.hpp:15-30).
struct Hit
{
  float m_parameterT = 0.0f;
  vec3 m_position = {};
  vec3 m_normal = {};
  std::shared_ptr<Material> m_material;

  Hit() = default;
  Hit(float t, vec3 const & position, vec3 const & normal, std::shared_ptr<Material> material)
    : m_parameterT(t), m_position(position), m_normal(normal), m_material(std::move(material))
  {}
};

class HitableObject
{
public:
  explicit HitableObject(std::shared_ptr<Material> material) : m_material(std::move(material)) {}
  virtual ~HitableObject() = default;

  virtual uint8_t GetType() const = 0;

  std::shared_ptr<Material> GetMaterial() const { return m_material; }

  // Every intersection with parameter in [tmin, tmax], nearest first (reference: hitable_object.hpp:46).
  virtual std::vector<Hit> Trace(Ray const & ray, float tmin, float tmax) const = 0;
  // The intersection a path tracer continues from (reference: hitable_object.hpp:47).
  virtual std::optional<Hit> TraceNearest(Ray const & ray, float tmin, float tmax) const = 0;

protected:
  std::shared_ptr<Material> m_material;
};

// All hits of all objects of a collection of (smart) pointers, by increasing parameter
// (reference: hitable_object.hpp:53-69).
template <typename T>
std::vector<Hit> TraceInHitableCollection(T const & collection, Ray const & ray, float tmin, float tmax)
{
  std::vector<Hit> all;
  all.reserve(collection.size() * 2);
  for (auto const & object : collection)
    for (Hit & h : object->Trace(ray, tmin, tmax))
      all.push_back(std::move(h));
  std::sort(all.begin(), all.end(), [](Hit const & x, Hit const & y) { return x.m_parameterT < y.m_parameterT; });
  return all;
}

// The nearest of the objects' nearest hits; among equal parameters the object that comes first in the collection
// wins (reference: hitable_object.hpp:71-87).
template <typename T>
std::optional<Hit> TraceNearestInHitableCollection(T const & collection, Ray const & ray, float tmin, float tmax)
{
  std::optional<Hit> best;
  for (auto const & object : collection)
  {
    std::optional<Hit> h = object->TraceNearest(ray, tmin, tmax);
    if (h && (!best || h->m_parameterT < best->m_parameterT))
      best = std::move(h);
  }
  return best;
}
}  // namespace ray_tracing
