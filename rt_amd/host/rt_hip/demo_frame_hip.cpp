#include "demo_frame_hip.hpp"

namespace ray_tracing_hip
{
bool DemoFrameHIP::Initialize(std::shared_ptr<ray_tracing::ColorBuffer> buffer, uint32_t width,
                              uint32_t height, uint32_t samplesInRowCount)
{
  if (!buffer || buffer->size() < static_cast<size_t>(width) * height)
    return false;
  return Frame::Initialize(std::move(buffer), width, height, samplesInRowCount);
}

void DemoFrameHIP::Fail(char const * where)
{
  m_failed = true;
  m_error = std::string(where) + ": " + rt_hip_last_error();
}

void DemoFrameHIP::Uninitialize()
{
  if (m_job != nullptr)
  {
    // Same order as the reference's shutdown: interrupt, wait, finish (rt_cuda/demo_frame_cuda.cpp:29-41).
    rt_hip_job_cancel(m_job);
    rt_hip_job_finish(m_job, nullptr, nullptr);
    m_job = nullptr;
  }
  if (m_scene != nullptr)
  {
    rt_hip_scene_destroy(m_scene);
    m_scene = nullptr;
    m_sceneDirty = true;
  }
}

rt_hip_camera DemoFrameHIP::Camera() const
{
  rt_hip_camera cam;
  cam.position[0] = m_cameraPosition.x; cam.position[1] = m_cameraPosition.y; cam.position[2] = m_cameraPosition.z;
  cam.direction[0] = m_cameraDirection.x; cam.direction[1] = m_cameraDirection.y; cam.direction[2] = m_cameraDirection.z;
  cam.fov = m_fov;
  cam.znear = m_znear;
  cam.zfar = m_zfar;
  cam.background[0] = m_backgroundColor.x; cam.background[1] = m_backgroundColor.y; cam.background[2] = m_backgroundColor.z;
  return cam;
}

void DemoFrameHIP::TraceAllRays()
{
  if (m_job != nullptr)
    return;  // one frame in flight, as in the reference (rt_cuda/demo_frame_cuda.cpp:45-46)

  m_failed = false;
  m_error.clear();
  if (!m_deviceReady)
  {
    if (rt_hip_init(m_device) != RT_HIP_OK)
      return Fail("rt_hip_init");
    m_deviceReady = true;
  }
  if (m_sceneDirty)
  {
    if (m_scene != nullptr)
      rt_hip_scene_destroy(m_scene);
    m_scene = nullptr;
    if (rt_hip_scene_create(m_flat.CentersRadii().data(), m_flat.MaterialIndex().data(), m_flat.SphereCount(),
                            m_flat.Materials().data(), static_cast<uint32_t>(m_flat.Materials().size()),
                            m_flat.Lights().data(), static_cast<uint32_t>(m_flat.Lights().size()),
                            &m_scene) != RT_HIP_OK)
      return Fail("rt_hip_scene_create");
    m_sceneDirty = false;
  }

  rt_hip_trace_params params = {};
  params.width = m_width;
  params.height = m_height;
  params.samples_in_row = m_samplesInRowCount;
  params.flags = (m_deterministic ? RT_HIP_FLAG_DETERMINISTIC : 0u) | (m_pixelRuns ? RT_HIP_FLAG_PIXEL_RUNS : 0u);
  params.seed = m_seed;
  params.max_depth = m_maxDepth;
  params.slice_index = m_sliceIndex;
  params.slice_count = m_sliceCount;
  rt_hip_camera const cam = Camera();
  if (rt_hip_trace_async(m_scene, &cam, &params, &m_job) != RT_HIP_OK)
  {
    m_job = nullptr;
    Fail("rt_hip_trace_async");
  }
}

bool DemoFrameHIP::HasFinished()
{
  if (m_job == nullptr)
    return true;
  if (rt_hip_job_poll(m_job) == RT_HIP_ERR_NOT_READY)
    return false;
  // Finishing is a side effect of this query, as in the reference (rt_cuda/demo_frame_cuda.cpp:146-149):
  // the image lands in the ColorBuffer given to Initialize, viewed as packed floats.
  if (rt_hip_job_finish(m_job, reinterpret_cast<float *>(m_buffer->data()), &m_stats) != RT_HIP_OK)
    Fail("rt_hip_job_finish");
  m_job = nullptr;
  return true;
}

bool DemoFrameHIP::InProgress()
{
  if (m_job == nullptr)
    return false;
  return rt_hip_job_poll(m_job) == RT_HIP_ERR_NOT_READY;
}

void DemoFrameHIP::CopyToBuffer(ray_tracing::ColorBuffer & buffer)
{
  if (m_job == nullptr || buffer.size() < static_cast<size_t>(m_width) * m_height)
    return;
  if (rt_hip_job_snapshot(m_job, reinterpret_cast<float *>(buffer.data())) != RT_HIP_OK)
    Fail("rt_hip_job_snapshot");
}

void DemoFrameHIP::AddObject(std::unique_ptr<ray_tracing::HitableObject> && object)
{
  if (m_job != nullptr)
    return;
  m_flat.AddObject(object.get());
  m_sceneDirty = true;
}

void DemoFrameHIP::AddLightSource(std::unique_ptr<ray_tracing::Light> && light)
{
  if (m_job != nullptr)
    return;
  m_flat.AddLightSource(light.get());
  m_sceneDirty = true;
}
}  // namespace ray_tracing_hip
