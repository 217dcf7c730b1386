// glass_spheres.hpp -- demo 2: 40 glass spheres on a grid, a darker ground sphere and one sun
// (scene recipe of the reference's demos/glass_spheres.hpp:33-73, drawn from a fixed seed).
#pragma once

#include "demo.hpp"
#include "palette.hpp"
#include "sphere_field.hpp"

namespace demo
{
class GlassSpheres : public Demo
{
public:
  explicit GlassSpheres(std::unique_ptr<ray_tracing::Frame> && frame, uint64_t seed = 1)
    : m_frame(std::move(frame)), m_seed(seed)
  {}

  bool Initialize(std::shared_ptr<ray_tracing::ColorBuffer> buffer, uint32_t width, uint32_t height,
                  uint32_t samplesInRowCount) override
  {
    if (!m_frame || !m_frame->Initialize(buffer, width, height, samplesInRowCount))
      return false;

    using namespace ray_tracing;
    m_frame->SetBackgroundColor(vec3(0.9f, 0.9f, 0.9f));
    m_frame->SetCameraPosition(vec3(0.0f, 2.0f, -10.0f));
    m_frame->SetCameraDirection(vec3(0.0f, -1.0f, 3.0f));

    SceneRandom rnd(m_seed);
    GridPositions positions;
    for (int i = 0; i < 40; ++i)
    {
      auto const cell = positions.Next(rnd);
      auto const color = Palette::RandomFromAll(rnd);
      auto mat = std::make_shared<material::Glass>(color, mix(0.8f, 0.9f, rnd.Float01()));
      float const radius = mix(0.25f, 0.5f, rnd.Float01());
      m_frame->AddObject(std::make_unique<Sphere>(
          vec3(static_cast<float>(cell.first), radius - 1.0f, static_cast<float>(cell.second)), radius, mat));
    }
    AddGroundAndSun(*m_frame, 0.4f);
    return true;
  }

  std::unique_ptr<ray_tracing::Frame> & GetFrame() override { return m_frame; }

private:
  std::unique_ptr<ray_tracing::Frame> m_frame;
  uint64_t m_seed;
};
}  // namespace demo
