// directional_light.hpp -- parallel light with a direction and a colour
// (reference: ray_tracing/directional_light.hpp:10-29).
#pragma once

#include "light.hpp"
#include "types.hpp"
#include "vec3.hpp"

namespace ray_tracing
{
class DirectionalLight : public Light
{
public:
  DirectionalLight(vec3 const & direction, vec3 const & color) : m_direction(direction), m_color(color) {}

  uint8_t GetType() const override { return kLightSourceDirectionalType; }

  vec3 const & GetDirection() const { return m_direction; }
  vec3 const & GetColor() const { return m_color; }

private:
  vec3 const m_direction;
  vec3 const m_color;
};
}  // namespace ray_tracing
