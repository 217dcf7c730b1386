// hitable_object.hpp -- base of the scene objects handed to Frame::AddObject
// (reference: ray_tracing/hitable_object.hpp:32-51).  Carries the material; intersection itself runs
// in the HIP kernel, so there are no host-side Trace virtuals.
#pragma once

#include "material.hpp"

#include <cstdint>
#include <memory>

namespace ray_tracing
{
class HitableObject
{
public:
  explicit HitableObject(std::shared_ptr<Material> material) : m_material(std::move(material)) {}
  virtual ~HitableObject() = default;

  virtual uint8_t GetType() const = 0;

  std::shared_ptr<Material> GetMaterial() const { return m_material; }

protected:
  std::shared_ptr<Material> m_material;
};
}  // namespace ray_tracing
