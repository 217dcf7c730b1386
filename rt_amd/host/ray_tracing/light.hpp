// light.hpp -- base of the light sources handed to Frame::AddLightSource
// (reference: ray_tracing/light.hpp:13-24; TraceLight/GetSpecular run in the HIP kernel).
#pragma once

#include <cstdint>

namespace ray_tracing
{
class Light
{
public:
  virtual ~Light() = default;
  virtual uint8_t GetType() const = 0;
};
}  // namespace ray_tracing
