// synthetic_spheres.hpp -- demo 3: a large field of random spheres for the brute-force stress case.
// Not in the reference.  `count` spheres sit one per cell of a square unit grid centred on the origin
// (cells are filled row by row), jittered inside their cell so that no two overlap; radius U(0.25, 0.5),
// material type uniform over matte / metal / glass, albedo U(0,1)^3, metal roughness and refraction
// U(0, 0.2), glass refraction U(0.8, 0.9); the ground sphere and the sun are those of the other demos.
#pragma once

#include "demo.hpp"
#include "sphere_field.hpp"

#include <cmath>

namespace demo
{
class SyntheticSpheres : public Demo
{
public:
  SyntheticSpheres(std::unique_ptr<ray_tracing::Frame> && frame, uint32_t count = 10000, uint64_t seed = 1)
    : m_frame(std::move(frame)), m_count(count), m_seed(seed)
  {}

  bool Initialize(std::shared_ptr<ray_tracing::ColorBuffer> buffer, uint32_t width, uint32_t height,
                  uint32_t samplesInRowCount) override
  {
    if (!m_frame || !m_frame->Initialize(buffer, width, height, samplesInRowCount))
      return false;

    using namespace ray_tracing;
    uint32_t const side = static_cast<uint32_t>(std::ceil(std::sqrt(static_cast<double>(m_count))));
    float const half = 0.5f * static_cast<float>(side);

    m_frame->SetBackgroundColor(vec3(0.9f, 0.9f, 0.9f));
    m_frame->SetCameraPosition(vec3(half + 2.5f, 0.16f * half + 1.0f, -half - 2.5f));
    m_frame->SetCameraDirection(vec3(-2.0f, -0.6f, 2.0f));

    SceneRandom rnd(m_seed);
    for (uint32_t i = 0; i < m_count; ++i)
    {
      uint32_t const row = i / side;
      uint32_t const col = i - row * side;
      float const radius = mix(0.25f, 0.5f, rnd.Float01());
      float const slack = 0.5f - radius;  // keeps the sphere inside its unit cell
      float const x = (static_cast<float>(col) + 0.5f - half) + slack * (2.0f * rnd.Float01() - 1.0f);
      float const z = (static_cast<float>(row) + 0.5f - half) + slack * (2.0f * rnd.Float01() - 1.0f);
      vec3 const albedo(rnd.Float01(), rnd.Float01(), rnd.Float01());
      int const kind = rnd.Int(0, 2);
      std::shared_ptr<Material> mat;
      if (kind == 0)
        mat = std::make_shared<material::Matte>(albedo);
      else if (kind == 1)
      {
        float const roughness = mix(0.0f, 0.2f, rnd.Float01());
        float const refraction = mix(0.0f, 0.2f, rnd.Float01());
        mat = std::make_shared<material::Metal>(albedo, roughness, refraction);
      }
      else
        mat = std::make_shared<material::Glass>(albedo, mix(0.8f, 0.9f, rnd.Float01()));
      m_frame->AddObject(std::make_unique<Sphere>(vec3(x, radius - 1.0f, z), radius, mat));
    }
    AddGroundAndSun(*m_frame, 0.65f);
    return true;
  }

  std::unique_ptr<ray_tracing::Frame> & GetFrame() override { return m_frame; }

private:
  std::unique_ptr<ray_tracing::Frame> m_frame;
  uint32_t m_count;
  uint64_t m_seed;
};
}  // namespace demo
