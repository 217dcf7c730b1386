#include "rt_host.h"

#include "demos/glass_spheres.hpp"
#include "demos/pretty_spheres.hpp"
#include "demos/scene_file.hpp"
#include "demos/synthetic_spheres.hpp"
#include "rt_hip/demo_frame_hip.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>

namespace
{
// A Frame that only records the scene; lets the demos be flattened without a device.
class RecordingFrame : public ray_tracing::Frame
{
public:
  void AddObject(std::unique_ptr<ray_tracing::HitableObject> && object) override { m_flat.AddObject(object.get()); }
  void AddLightSource(std::unique_ptr<ray_tracing::Light> && light) override { m_flat.AddLightSource(light.get()); }

  ray_tracing_hip::SceneFlattener m_flat;

  rt_hip_camera Camera() const
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
};

std::unique_ptr<demo::Demo> MakeDemo(uint32_t index, std::unique_ptr<ray_tracing::Frame> && frame, uint32_t seed,
                                     uint32_t syntheticSpheres)
{
  switch (index)
  {
  case RT_HOST_DEMO_PRETTY: return std::make_unique<demo::PrettySpheres>(std::move(frame), seed);
  case RT_HOST_DEMO_GLASS: return std::make_unique<demo::GlassSpheres>(std::move(frame), seed);
  case RT_HOST_DEMO_SYNTHETIC:
    return std::make_unique<demo::SyntheticSpheres>(std::move(frame), syntheticSpheres, seed);
  default: return nullptr;
  }
}
}  // namespace

struct rt_host_scene
{
  ray_tracing_hip::SceneFlattener flat;
  rt_hip_camera camera;
};

extern "C" int rt_host_scene_build(uint32_t demoIndex, uint32_t seed, uint32_t syntheticSpheres,
                                   rt_host_scene ** out)
{
  if (out == nullptr)
    return 1;
  *out = nullptr;
  auto d = MakeDemo(demoIndex, std::make_unique<RecordingFrame>(), seed, syntheticSpheres);
  if (!d)
    return 1;
  auto buffer = std::make_shared<ray_tracing::ColorBuffer>(1);
  if (!d->Initialize(buffer, 1, 1, 1))
    return 2;
  auto * frame = static_cast<RecordingFrame *>(d->GetFrame().get());
  auto * scene = new rt_host_scene;
  scene->flat = frame->m_flat;
  scene->camera = frame->Camera();
  *out = scene;
  return 0;
}

extern "C" int rt_host_scene_load(const char * path, rt_host_scene ** out, char * err, size_t errLen)
{
  if (out == nullptr || path == nullptr)
    return 1;
  *out = nullptr;
  demo::SceneFile d(std::make_unique<RecordingFrame>(), path);
  auto buffer = std::make_shared<ray_tracing::ColorBuffer>(1);
  if (!d.Initialize(buffer, 1, 1, 1))
  {
    if (err && errLen)
      std::snprintf(err, errLen, "%s", d.Error().c_str());
    return 2;
  }
  auto * frame = static_cast<RecordingFrame *>(d.GetFrame().get());
  auto * scene = new rt_host_scene;
  scene->flat = frame->m_flat;
  scene->camera = frame->Camera();
  *out = scene;
  return 0;
}

extern "C" int rt_host_scene_save(const char * path, const rt_hip_camera * camera, const float * centersRadii,
                                  const uint32_t * materialIndex, uint32_t sphereCount,
                                  const rt_hip_material * materials, uint32_t materialCount,
                                  const rt_hip_light * lights, uint32_t lightCount)
{
  if (path == nullptr || camera == nullptr)
    return 1;
  return demo::WriteSceneFile(path, *camera, centersRadii, materialIndex, sphereCount, materials, materialCount,
                              lights, lightCount)
             ? 0
             : 2;
}

extern "C" void rt_host_scene_counts(const rt_host_scene * scene, uint32_t * spheres, uint32_t * materials,
                                     uint32_t * lights)
{
  if (spheres) *spheres = scene->flat.SphereCount();
  if (materials) *materials = static_cast<uint32_t>(scene->flat.Materials().size());
  if (lights) *lights = static_cast<uint32_t>(scene->flat.Lights().size());
}

extern "C" void rt_host_scene_copy(const rt_host_scene * scene, float * centersRadii, uint32_t * materialIndex,
                                   rt_hip_material * materials, rt_hip_light * lights, rt_hip_camera * camera)
{
  auto const & f = scene->flat;
  if (centersRadii) std::copy(f.CentersRadii().begin(), f.CentersRadii().end(), centersRadii);
  if (materialIndex) std::copy(f.MaterialIndex().begin(), f.MaterialIndex().end(), materialIndex);
  if (materials) std::copy(f.Materials().begin(), f.Materials().end(), materials);
  if (lights) std::copy(f.Lights().begin(), f.Lights().end(), lights);
  if (camera) *camera = scene->camera;
}

extern "C" void rt_host_scene_free(rt_host_scene * scene) { delete scene; }

extern "C" int rt_host_render_demo(uint32_t demoIndex, uint32_t sceneSeed, uint32_t syntheticSpheres,
                                   const rt_hip_trace_params * params, float * rgb, rt_hip_stats * stats,
                                   char * err, size_t errLen)
{
  auto fail = [&](char const * msg, int code) {
    if (err && errLen)
      std::snprintf(err, errLen, "%s", msg);
    return code;
  };
  if (params == nullptr || rgb == nullptr)
    return fail("null argument", 1);

  auto hipFrame = std::make_unique<ray_tracing_hip::DemoFrameHIP>();
  auto * hip = hipFrame.get();
  hip->SetSeed(params->seed);
  hip->SetDeterministic((params->flags & RT_HIP_FLAG_DETERMINISTIC) != 0u);
  hip->SetMaxDepth(params->max_depth);
  hip->SetSampleSlice(params->slice_index, params->slice_count ? params->slice_count : 1u);
  auto d = MakeDemo(demoIndex, std::move(hipFrame), sceneSeed, syntheticSpheres);
  if (!d)
    return fail("unknown demo index", 1);

  size_t const pixels = static_cast<size_t>(params->width) * params->height;
  auto buffer = std::make_shared<ray_tracing::ColorBuffer>(pixels);
  if (!d->Initialize(buffer, params->width, params->height, params->samples_in_row))
    return fail("Demo::Initialize failed", 2);

  auto & frame = d->GetFrame();
  frame->TraceAllRays();
  if (hip->Failed())
    return fail(hip->LastError().c_str(), 3);
  while (!frame->HasFinished())
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  if (hip->Failed())
    return fail(hip->LastError().c_str(), 3);

  std::memcpy(rgb, buffer->data(), pixels * sizeof(ray_tracing::vec3));
  if (stats)
    *stats = hip->LastStats();
  frame->Uninitialize();
  return 0;
}

extern "C" uint32_t rt_host_trace_hits(const float * centersRadii, uint32_t sphereCount, const float origin[3],
                                       const float direction[3], float tmin, float tmax, uint32_t mode,
                                       rt_host_hit * out, uint32_t capacity)
{
  using namespace ray_tracing;
  // every sphere gets a material of its own, so that a hit's material pointer names the sphere it lies on
  std::vector<std::unique_ptr<HitableObject>> objects;
  objects.reserve(sphereCount);
  for (uint32_t i = 0; i < sphereCount; ++i)
  {
    const float * s = centersRadii + 4u * i;
    objects.emplace_back(std::make_unique<Sphere>(vec3(s[0], s[1], s[2]), s[3],
                                                  std::make_shared<material::Matte>(vec3(0.5f, 0.5f, 0.5f))));
  }
  Ray const ray(vec3(origin[0], origin[1], origin[2]), vec3(direction[0], direction[1], direction[2]));
  std::vector<Hit> hits;
  if (mode == 0u)
    hits = TraceInHitableCollection(objects, ray, tmin, tmax);
  else if (auto nearest = TraceNearestInHitableCollection(objects, ray, tmin, tmax))
    hits.push_back(*nearest);
  for (uint32_t k = 0; k < hits.size() && k < capacity; ++k)
  {
    Hit const & h = hits[k];
    out[k].t = h.m_parameterT;
    out[k].position[0] = h.m_position.x; out[k].position[1] = h.m_position.y; out[k].position[2] = h.m_position.z;
    out[k].normal[0] = h.m_normal.x; out[k].normal[1] = h.m_normal.y; out[k].normal[2] = h.m_normal.z;
    out[k].sphere = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < sphereCount; ++i)
      if (objects[i]->GetMaterial() == h.m_material)
        out[k].sphere = i;
  }
  return static_cast<uint32_t>(hits.size());
}

extern "C" void rt_host_update_texture(const float * rgb, size_t pixels, uint8_t * rgba)
{
  auto const clamp01 = [](float v) { return std::min(std::max(v, 0.0f), 1.0f); };
  for (size_t i = 0; i < pixels; ++i)
  {
    rgba[i * 4 + 0] = static_cast<uint8_t>(255 * clamp01(rgb[i * 3 + 0]));
    rgba[i * 4 + 1] = static_cast<uint8_t>(255 * clamp01(rgb[i * 3 + 1]));
    rgba[i * 4 + 2] = static_cast<uint8_t>(255 * clamp01(rgb[i * 3 + 2]));
    rgba[i * 4 + 3] = 255;
  }
}
