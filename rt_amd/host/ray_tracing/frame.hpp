// frame.hpp -- the plug-in boundary a rendering backend implements
// (reference: ray_tracing/frame.hpp:24-73; same virtuals, setters, members and defaults).
//
// A Demo owns one Frame and drives it the way App does in the reference: Initialize, AddObject /
// AddLightSource while building the scene, TraceAllRays (asynchronous), then InProgress / CopyToBuffer
// for progressive display and HasFinished for completion.  The CPU-side per-ray hooks of the reference
// (RayTrace, ForEachRay, OnStartRow, OnEndRow) belong to its CPU backend and have no counterpart here.
#pragma once

#include "hitable_object.hpp"
#include "light.hpp"
#include "types.hpp"
#include "vec3.hpp"

#include <cstdint>
#include <memory>
#include <vector>

namespace ray_tracing
{
using ColorBuffer = std::vector<vec3>;

class Frame
{
public:
  Frame() = default;
  virtual ~Frame() = default;

  virtual bool Initialize(std::shared_ptr<ColorBuffer> buffer, uint32_t width, uint32_t height,
                          uint32_t samplesInRowCount)
  {
    m_buffer = std::move(buffer);
    m_width = width;
    m_height = height;
    m_samplesInRowCount = samplesInRowCount;
    return true;
  }
  virtual void Uninitialize() {}

  virtual void TraceAllRays() {}

  virtual bool HasFinished() { return true; }
  virtual bool InProgress() { return false; }
  virtual void CopyToBuffer(ColorBuffer & buffer) { (void)buffer; }

  virtual void AddObject(std::unique_ptr<HitableObject> && object) { (void)object; }
  virtual void AddLightSource(std::unique_ptr<Light> && light) { (void)light; }

  uint32_t GetSamplesInRowCount() const { return m_samplesInRowCount; }
  void SetSamplesInRowCount(uint32_t samplesInRowCount) { m_samplesInRowCount = samplesInRowCount; }

  void SetBackgroundColor(vec3 const & color) { m_backgroundColor = color; }

  void SetCameraPosition(vec3 const & pos) { m_cameraPosition = pos; }
  // Stored normalised, as the reference does (ray_tracing/frame.cpp:79-82).
  void SetCameraDirection(vec3 const & dir) { m_cameraDirection = normalize(dir); }

  // Field of view (radians) and the near / far planes.  Not in the reference, whose lens is fixed at the defaults
  // below (ray_tracing/frame.hpp:70-72); scene files may carry another one.
  void SetLens(float fov, float znear, float zfar)
  {
    m_fov = fov;
    m_znear = znear;
    m_zfar = zfar;
  }

protected:
  std::shared_ptr<ColorBuffer> m_buffer;
  uint32_t m_width = 0;
  uint32_t m_height = 0;
  uint32_t m_samplesInRowCount = 1;

  vec3 m_backgroundColor = vec3(1.0f, 1.0f, 1.0f);

  vec3 m_cameraPosition = vec3(0.0f, 0.0f, 0.0f);
  vec3 m_cameraDirection = vec3(0.0f, 0.0f, 1.0f);
  float m_fov = 2.0f * kPi / 3.0f;
  float m_znear = 0.001f;
  float m_zfar = 1000.0f;
};
}  // namespace ray_tracing
