// scene_flattener.hpp -- turns the object model handed to Frame::AddObject / AddLightSource into the
// plain arrays the rt_hip C ABI takes.
//
// Same job as DemoFrameCUDA::AddObject / FindMaterial / AddLightSource in the reference
// (rt_cuda/demo_frame_cuda.cpp:69-128): spheres and directional lights are kept, anything else is
// dropped silently, and materials are shared between spheres when their parameters agree within kEps.
// Unlike the reference, two materials are only shared when their type tags agree as well.
#pragma once

#include "ray_tracing/default_materials.hpp"
#include "ray_tracing/directional_light.hpp"
#include "ray_tracing/sphere.hpp"
#include "ray_tracing/types.hpp"

#include "rt_hip.h"

#include <cmath>
#include <cstdint>
#include <memory>
#include <vector>

namespace ray_tracing_hip
{
class SceneFlattener
{
public:
  void AddObject(ray_tracing::HitableObject const * object)
  {
    auto const * sphere = dynamic_cast<ray_tracing::Sphere const *>(object);
    if (sphere == nullptr)
      return;
    auto const c = sphere->GetCenter();
    m_centersRadii.push_back(c.x);
    m_centersRadii.push_back(c.y);
    m_centersRadii.push_back(c.z);
    m_centersRadii.push_back(sphere->GetRadius());
    m_materialIndex.push_back(FindMaterial(*sphere->GetMaterial()));
  }

  void AddLightSource(ray_tracing::Light const * light)
  {
    auto const * directional = dynamic_cast<ray_tracing::DirectionalLight const *>(light);
    if (directional == nullptr)
      return;
    rt_hip_light l;
    l.type = directional->GetType();
    auto const & d = directional->GetDirection();
    auto const & c = directional->GetColor();
    l.direction[0] = d.x; l.direction[1] = d.y; l.direction[2] = d.z;
    l.color[0] = c.x; l.color[1] = c.y; l.color[2] = c.z;
    m_lights.push_back(l);
  }

  void Clear()
  {
    m_centersRadii.clear();
    m_materialIndex.clear();
    m_materials.clear();
    m_lights.clear();
  }

  uint32_t SphereCount() const { return static_cast<uint32_t>(m_materialIndex.size()); }
  std::vector<float> const & CentersRadii() const { return m_centersRadii; }
  std::vector<uint32_t> const & MaterialIndex() const { return m_materialIndex; }
  std::vector<rt_hip_material> const & Materials() const { return m_materials; }
  std::vector<rt_hip_light> const & Lights() const { return m_lights; }

private:
  uint32_t FindMaterial(ray_tracing::Material const & mat)
  {
    auto const albedo = mat.GetAlbedo();
    float const roughness = mat.GetRoughness();
    float const refraction = mat.GetRefraction();
    uint32_t const type = mat.GetType();
    auto const near = [](float a, float b) { return std::fabs(a - b) < ray_tracing::kEps; };
    for (size_t i = 0; i < m_materials.size(); ++i)
    {
      auto const & m = m_materials[i];
      if (m.type == type && near(m.albedo[0], albedo.x) && near(m.albedo[1], albedo.y) &&
          near(m.albedo[2], albedo.z) && near(m.roughness, roughness) && near(m.refraction, refraction))
        return static_cast<uint32_t>(i);
    }
    rt_hip_material m;
    m.type = type;
    m.albedo[0] = albedo.x; m.albedo[1] = albedo.y; m.albedo[2] = albedo.z;
    m.roughness = roughness;
    m.refraction = refraction;
    m_materials.push_back(m);
    return static_cast<uint32_t>(m_materials.size() - 1);
  }

  std::vector<float> m_centersRadii;       // 4 per sphere
  std::vector<uint32_t> m_materialIndex;   // 1 per sphere
  std::vector<rt_hip_material> m_materials;
  std::vector<rt_hip_light> m_lights;
};
}  // namespace ray_tracing_hip
