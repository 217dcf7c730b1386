// default_materials.hpp -- Matte, Metal and Glass parameter holders
// (reference: ray_tracing/default_materials.hpp:12-70; same constructors and getters).
#pragma once

#include "material.hpp"

namespace ray_tracing
{
namespace material
{
// Lambert-like diffuse surface; keeps Material's default roughness (0.7) and refraction (0).
class Matte : public Material
{
public:
  explicit Matte(vec3 const & albedo) : m_albedo(albedo) {}

  uint8_t GetType() const override { return kMaterialMatteType; }
  vec3 GetAlbedo() const override { return m_albedo; }

private:
  vec3 m_albedo;
};

// Rough mirror.
class Metal : public Material
{
public:
  Metal(vec3 const & albedo, float roughness, float refraction)
    : m_albedo(albedo), m_roughness(roughness), m_refraction(refraction)
  {}

  uint8_t GetType() const override { return kMaterialMetalType; }
  vec3 GetAlbedo() const override { return m_albedo; }
  float GetRoughness() const override { return m_roughness; }
  float GetRefraction() const override { return m_refraction; }

private:
  vec3 m_albedo;
  float m_roughness;
  float m_refraction;
};

// Reflecting / refracting dielectric; reports roughness 0.
class Glass : public Material
{
public:
  Glass(vec3 const & albedo, float refraction) : m_albedo(albedo), m_refraction(refraction) {}

  uint8_t GetType() const override { return kMaterialGlassType; }
  vec3 GetAlbedo() const override { return m_albedo; }
  float GetRoughness() const override { return 0.0f; }
  float GetRefraction() const override { return m_refraction; }

private:
  vec3 m_albedo;
  float m_refraction;
};
}  // namespace material
}  // namespace ray_tracing
