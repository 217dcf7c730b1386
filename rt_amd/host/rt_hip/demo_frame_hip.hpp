// demo_frame_hip.hpp -- ray_tracing::Frame backed by the MI355X HIP library (librt_hip).
//
// Takes the place DemoFrameCUDA has in the reference (rt_cuda/demo_frame_cuda.hpp:15-48): same overrides,
// constructed the same way by the demo selector.  It needs no worker thread, mutex or progress callback:
// TraceAllRays enqueues the frame and returns, InProgress/HasFinished poll the job, CopyToBuffer pulls a
// snapshot.  There is no CPU fallback: if the device or the library is unusable, TraceAllRays records the
// error, Failed() turns true and nothing is rendered.
#pragma once

#include "ray_tracing/frame.hpp"
#include "scene_flattener.hpp"

#include "rt_hip.h"

#include <string>

namespace ray_tracing_hip
{
class DemoFrameHIP : public ray_tracing::Frame
{
public:
  // device < 0 keeps the process's current device.
  explicit DemoFrameHIP(int device = -1) : m_device(device) {}
  ~DemoFrameHIP() override { Uninitialize(); }

  bool Initialize(std::shared_ptr<ray_tracing::ColorBuffer> buffer, uint32_t width, uint32_t height,
                  uint32_t samplesInRowCount) override;
  void Uninitialize() override;

  void TraceAllRays() override;

  bool HasFinished() override;
  bool InProgress() override;
  void CopyToBuffer(ray_tracing::ColorBuffer & buffer) override;

  void AddObject(std::unique_ptr<ray_tracing::HitableObject> && object) override;
  void AddLightSource(std::unique_ptr<ray_tracing::Light> && light) override;

  // Knobs the reference does not have.
  void SetSeed(uint32_t seed) { m_seed = seed; }
  void SetDeterministic(bool on) { m_deterministic = on; }
  void SetMaxDepth(uint32_t depth) { m_maxDepth = depth; }
  void SetSampleSlice(uint32_t index, uint32_t count) { m_sliceIndex = index; m_sliceCount = count; m_pixelRuns = false; }
  // The same slice cut by runs of 64 pixels (RT_HIP_FLAG_PIXEL_RUNS): every sample of every count-th run.
  void SetPixelRunSlice(uint32_t index, uint32_t count) { m_sliceIndex = index; m_sliceCount = count; m_pixelRuns = true; }

  bool Failed() const { return m_failed; }
  std::string const & LastError() const { return m_error; }
  rt_hip_stats const & LastStats() const { return m_stats; }

  SceneFlattener const & Scene() const { return m_flat; }
  rt_hip_camera Camera() const;

private:
  void Fail(char const * where);

  int m_device;
  uint32_t m_seed = 1;
  bool m_deterministic = false;
  uint32_t m_maxDepth = 0;
  uint32_t m_sliceIndex = 0;
  uint32_t m_sliceCount = 1;
  bool m_pixelRuns = false;

  SceneFlattener m_flat;  // must not change while a frame is in flight
  bool m_sceneDirty = true;
  rt_hip_scene * m_scene = nullptr;
  rt_hip_job * m_job = nullptr;
  bool m_deviceReady = false;
  bool m_failed = false;
  std::string m_error;
  rt_hip_stats m_stats = {};
};
}  // namespace ray_tracing_hip
