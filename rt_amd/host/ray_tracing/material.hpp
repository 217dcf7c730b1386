// material.hpp -- what a backend needs to know about a surface.
//
// Mirrors the data side of the reference's Material (ray_tracing/material.hpp:8-27): the type tag and
// the three getters a GPU backend flattens (rt_cuda/demo_frame_cuda.cpp:87-111), with the same
// defaults -- albedo 0, roughness 0.7, refraction 0.  The per-ray evaluation (Scatter) is not a host
// virtual here: it runs inside the HIP kernel.
#pragma once

#include "types.hpp"
#include "vec3.hpp"

#include <cstdint>

namespace ray_tracing
{
class Material
{
public:
  virtual ~Material() = default;

  virtual uint8_t GetType() const = 0;

  virtual vec3 GetAlbedo() const { return vec3(0.0f, 0.0f, 0.0f); }
  virtual float GetRoughness() const { return 0.7f; }
  virtual float GetRefraction() const { return 0.0f; }
};
}  // namespace ray_tracing
