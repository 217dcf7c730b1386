// frame_cpu.cpp -- the oracle behind the Frame plug-in interface.  TEST INFRASTRUCTURE ONLY.
//
// MultiThreadedFrame and DemoFrameCPU here stand where the reference's ray_tracing::MultiThreadedFrame
// (ray_tracing/multithreaded_frame.hpp:20-53) and demo::DemoFrameCPU (demos/demo_frame_cpu.hpp:11-36) stand: a
// ray_tracing::Frame that a demo::Demo is built around, constructed with the number of worker threads, tracing
// asynchronously after TraceAllRays.  The tracing itself is the C oracle (orc_render: the reference's per-ray algorithm on
// MultiThreadedFrame's row interleave, run to true completion).  With it the same scene-building code
// (demo::PrettySpheres, demo::GlassSpheres, demo::SceneFile) drives the CPU checker and the HIP backend through
// the one interface both implement, which is what the drop-in claim rests on.  Nothing that ships links this.
#include "demos/glass_spheres.hpp"
#include "demos/pretty_spheres.hpp"
#include "demos/scene_file.hpp"
#include "demos/synthetic_spheres.hpp"
#include "rt_hip/scene_flattener.hpp"

#include "rt_oracle.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <thread>

namespace
{
// Where the reference's ray_tracing::MultiThreadedFrame stands (ray_tracing/multithreaded_frame.hpp:20-53): a
// Frame constructed with the number of ray-tracing threads whose TraceAllRays returns at once and whose
// HasFinished / InProgress / CopyToBuffer report on the frame being traced in the background.  The row
// interleave over the threads (multithreaded_frame.cpp:49) is the C oracle's (orc_render); a subclass says what a
// frame is by overriding TraceFrame, as the reference's subclass does by overriding RayTrace.
class MultiThreadedFrame : public ray_tracing::Frame
{
public:
  explicit MultiThreadedFrame(uint32_t rayTracingThreadsCount)
    : m_rayTracingThreadsCount(rayTracingThreadsCount ? rayTracingThreadsCount : 1u)
  {}
  ~MultiThreadedFrame() override { JoinWorker(); }

  void Uninitialize() override { JoinWorker(); }

  // Asynchronous, as MultiThreadedFrame::ForEachRay is (multithreaded_frame.cpp:99-108): returns at once.
  void TraceAllRays() override
  {
    if (m_running.load())
      return;
    JoinWorker();
    m_running.store(true);
    m_worker = std::thread([this]() {
      TraceFrame(m_rayTracingThreadsCount);
      m_running.store(false);
    });
  }

  // True completion of every row (the reference's counter can report early, multithreaded_frame.cpp:51-57).
  bool HasFinished() override { return !m_running.load(); }
  bool InProgress() override { return m_running.load(); }
  // Finished frames only; the checker has no use for partial ones.
  void CopyToBuffer(ray_tracing::ColorBuffer & buffer) override
  {
    if (!m_running.load() && m_buffer && buffer.size() >= m_buffer->size())
      std::memcpy(buffer.data(), m_buffer->data(), m_buffer->size() * sizeof(ray_tracing::vec3));
  }

protected:
  // Traces the whole frame into m_buffer on `threads` threads and returns when every row is done.
  virtual void TraceFrame(uint32_t threads) = 0;

private:
  void JoinWorker()
  {
    if (m_worker.joinable())
      m_worker.join();
  }

  uint32_t m_rayTracingThreadsCount;
  std::thread m_worker;
  std::atomic<bool> m_running{false};
};

// Where the reference's demo::DemoFrameCPU stands (demos/demo_frame_cpu.hpp:11-36): owns the objects and lights
// and traces them with the reference's per-ray algorithm, here the C oracle's restatement of it.
class DemoFrameCPU : public MultiThreadedFrame
{
public:
  explicit DemoFrameCPU(uint32_t rayTracingThreadsCount) : MultiThreadedFrame(rayTracingThreadsCount) {}
  ~DemoFrameCPU() override { Uninitialize(); }

  void SetSeed(uint32_t seed) { m_seed = seed; }
  void SetFlags(uint32_t flags) { m_flags = flags; }
  void SetMaxDepth(uint32_t depth) { m_maxDepth = depth; }
  orc_stats const & LastStats() const { return m_stats; }

  void AddObject(std::unique_ptr<ray_tracing::HitableObject> && object) override { m_flat.AddObject(object.get()); }
  void AddLightSource(std::unique_ptr<ray_tracing::Light> && light) override { m_flat.AddLightSource(light.get()); }

protected:
  void TraceFrame(uint32_t threads) override
  {
    orc_scene scene;
    scene.spheres = m_flat.CentersRadii().data();
    scene.material_index = m_flat.MaterialIndex().data();
    scene.sphere_count = m_flat.SphereCount();
    // rt_hip_material / rt_hip_light and orc_material / orc_light have the same layout (checked below)
    scene.materials = reinterpret_cast<const orc_material *>(m_flat.Materials().data());
    scene.material_count = static_cast<uint32_t>(m_flat.Materials().size());
    scene.lights = reinterpret_cast<const orc_light *>(m_flat.Lights().data());
    scene.light_count = static_cast<uint32_t>(m_flat.Lights().size());
    orc_camera cam;
    cam.position[0] = m_cameraPosition.x; cam.position[1] = m_cameraPosition.y; cam.position[2] = m_cameraPosition.z;
    cam.direction[0] = m_cameraDirection.x; cam.direction[1] = m_cameraDirection.y; cam.direction[2] = m_cameraDirection.z;
    cam.fov = m_fov;
    cam.znear = m_znear;
    cam.zfar = m_zfar;
    cam.background[0] = m_backgroundColor.x; cam.background[1] = m_backgroundColor.y; cam.background[2] = m_backgroundColor.z;
    orc_params p;
    std::memset(&p, 0, sizeof(p));
    p.width = m_width;
    p.height = m_height;
    p.samples_in_row = m_samplesInRowCount;
    p.flags = m_flags;
    p.seed = m_seed;
    p.max_depth = m_maxDepth;
    p.slice_count = 1;
    // ColorBuffer is three packed floats per pixel, as the reference's GPU path assumes too
    // (rt_cuda/demo_frame_cuda.cpp:39).
    orc_render(&scene, &cam, &p, threads, reinterpret_cast<float *>(m_buffer->data()), &m_stats);
  }

private:
  uint32_t m_seed = 1, m_flags = 0, m_maxDepth = 0;
  ray_tracing_hip::SceneFlattener m_flat;
  orc_stats m_stats = {};
};

static_assert(sizeof(orc_material) == sizeof(rt_hip_material) && sizeof(orc_light) == sizeof(rt_hip_light),
              "the oracle's and the C ABI's material / light records must agree");
static_assert(sizeof(ray_tracing::vec3) == 3 * sizeof(float), "ColorBuffer must be packed floats");

int Drive(demo::Demo & d, DemoFrameCPU & frame, const orc_params * params, float * rgb, orc_stats * stats)
{
  size_t const pixels = static_cast<size_t>(params->width) * params->height;
  auto buffer = std::make_shared<ray_tracing::ColorBuffer>(pixels);
  if (!d.Initialize(buffer, params->width, params->height, params->samples_in_row))
    return 2;
  // Driven as the reference's App drives a frame (app.cpp:61-77): start, poll, take the buffer.
  d.GetFrame()->TraceAllRays();
  while (!d.GetFrame()->HasFinished())
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  std::memcpy(rgb, buffer->data(), pixels * sizeof(ray_tracing::vec3));
  if (stats)
    *stats = frame.LastStats();
  d.GetFrame()->Uninitialize();
  return 0;
}

std::unique_ptr<DemoFrameCPU> MakeFrame(const orc_params * params, uint32_t threads, DemoFrameCPU *& raw)
{
  auto f = std::make_unique<DemoFrameCPU>(threads);
  f->SetSeed(params->seed);
  f->SetFlags(params->flags);
  f->SetMaxDepth(params->max_depth);
  raw = f.get();
  return f;
}
}  // namespace

// A built-in demo (1 pretty, 2 glass, 3 synthetic) rendered through the Frame interface on the CPU checker.
extern "C" int orc_frame_render_demo(uint32_t demoIndex, uint32_t sceneSeed, uint32_t syntheticSpheres,
                                     const orc_params * params, uint32_t threads, float * rgb, orc_stats * stats)
{
  if (params == nullptr || rgb == nullptr)
    return 1;
  DemoFrameCPU * raw = nullptr;
  auto frame = MakeFrame(params, threads, raw);
  std::unique_ptr<demo::Demo> d;
  switch (demoIndex)
  {
  case 1: d = std::make_unique<demo::PrettySpheres>(std::move(frame), sceneSeed); break;
  case 2: d = std::make_unique<demo::GlassSpheres>(std::move(frame), sceneSeed); break;
  case 3: d = std::make_unique<demo::SyntheticSpheres>(std::move(frame), syntheticSpheres, sceneSeed); break;
  default: return 1;
  }
  return Drive(*d, *raw, params, rgb, stats);
}

// The same for a scene file.
extern "C" int orc_frame_render_scene_file(const char * path, const orc_params * params, uint32_t threads, float * rgb,
                                           orc_stats * stats)
{
  if (path == nullptr || params == nullptr || rgb == nullptr)
    return 1;
  DemoFrameCPU * raw = nullptr;
  auto frame = MakeFrame(params, threads, raw);
  demo::SceneFile d(std::move(frame), path);
  return Drive(d, *raw, params, rgb, stats);
}
