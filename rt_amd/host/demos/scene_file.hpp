// scene_file.hpp -- a scene read from (or written to) a small text file, so that the same spheres can feed any
// backend.  The reference builds its scenes in code only (demos/pretty_spheres.hpp, demos/glass_spheres.hpp);
// this is the file form of what those builders hand to a Frame: camera, background, materials, spheres, lights.
//
//   rt-scene 1
//   # comment
//   camera     px py pz  dx dy dz          (direction need not be normalised)
//   lens       fov znear zfar            (optional; field of view in radians.  Default: the reference's fixed
//                                         2 pi / 3, 0.001, 1000 of ray_tracing/frame.hpp:70-72)
//   background r g b
//   material   <name> matte r g b
//   material   <name> metal r g b roughness refraction
//   material   <name> glass r g b refraction
//   sphere     cx cy cz radius <material name>
//   light      directional dx dy dz  r g b
#pragma once

#include "demo.hpp"
#include "ray_tracing/default_materials.hpp"
#include "ray_tracing/directional_light.hpp"
#include "ray_tracing/sphere.hpp"

#include "rt_hip.h"

#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>
#include <string>

namespace demo
{
class SceneFile : public Demo
{
public:
  SceneFile(std::unique_ptr<ray_tracing::Frame> && frame, std::string path)
    : m_frame(std::move(frame)), m_path(std::move(path))
  {}

  bool Initialize(std::shared_ptr<ray_tracing::ColorBuffer> buffer, uint32_t width, uint32_t height,
                  uint32_t samplesInRowCount) override
  {
    if (!m_frame || !m_frame->Initialize(buffer, width, height, samplesInRowCount))
      return false;
    return Parse();
  }

  std::unique_ptr<ray_tracing::Frame> & GetFrame() override { return m_frame; }
  std::string const & Error() const { return m_error; }

private:
  bool Fail(size_t line, std::string const & what)
  {
    m_error = m_path + ":" + std::to_string(line) + ": " + what;
    return false;
  }

  bool Parse()
  {
    using namespace ray_tracing;
    std::ifstream f(m_path);
    if (!f)
      return Fail(0, "cannot open");
    std::map<std::string, std::shared_ptr<Material>> materials;
    std::string text;
    size_t line = 0;
    bool header = false;
    while (std::getline(f, text))
    {
      ++line;
      auto const hash = text.find('#');
      if (hash != std::string::npos)
        text.erase(hash);
      std::istringstream in(text);
      std::string key;
      if (!(in >> key))
        continue;
      if (!header)
      {
        int version = 0;
        if (key != "rt-scene" || !(in >> version) || version != 1)
          return Fail(line, "expected the header 'rt-scene 1'");
        header = true;
        continue;
      }
      if (key == "camera")
      {
        vec3 p, d;
        if (!(in >> p.x >> p.y >> p.z >> d.x >> d.y >> d.z))
          return Fail(line, "camera needs a position and a direction");
        m_frame->SetCameraPosition(p);
        m_frame->SetCameraDirection(d);
      }
      else if (key == "lens")
      {
        float fov = 0.0f, znear = 0.0f, zfar = 0.0f;
        if (!(in >> fov >> znear >> zfar) || !(fov > 0.0f) || !(znear > 0.0f) || !(zfar > znear))
          return Fail(line, "lens needs a field of view, znear and zfar (0 < znear < zfar)");
        m_frame->SetLens(fov, znear, zfar);
      }
      else if (key == "background")
      {
        vec3 c;
        if (!(in >> c.x >> c.y >> c.z))
          return Fail(line, "background needs a colour");
        m_frame->SetBackgroundColor(c);
      }
      else if (key == "material")
      {
        std::string name, type;
        vec3 c;
        if (!(in >> name >> type >> c.x >> c.y >> c.z))
          return Fail(line, "material needs a name, a type and a colour");
        if (type == "matte")
          materials[name] = std::make_shared<material::Matte>(c);
        else if (type == "metal")
        {
          float roughness = 0.0f, refraction = 0.0f;
          if (!(in >> roughness >> refraction))
            return Fail(line, "metal needs roughness and refraction");
          materials[name] = std::make_shared<material::Metal>(c, roughness, refraction);
        }
        else if (type == "glass")
        {
          float refraction = 0.0f;
          if (!(in >> refraction))
            return Fail(line, "glass needs a refraction index");
          materials[name] = std::make_shared<material::Glass>(c, refraction);
        }
        else
          return Fail(line, "unknown material type '" + type + "'");
      }
      else if (key == "sphere")
      {
        vec3 c;
        float radius = 0.0f;
        std::string name;
        if (!(in >> c.x >> c.y >> c.z >> radius >> name))
          return Fail(line, "sphere needs a centre, a radius and a material name");
        auto const it = materials.find(name);
        if (it == materials.end())
          return Fail(line, "unknown material '" + name + "'");
        m_frame->AddObject(std::make_unique<Sphere>(c, radius, it->second));
      }
      else if (key == "light")
      {
        std::string type;
        vec3 d, c;
        if (!(in >> type >> d.x >> d.y >> d.z >> c.x >> c.y >> c.z) || type != "directional")
          return Fail(line, "light needs 'directional', a direction and a colour");
        m_frame->AddLightSource(std::make_unique<DirectionalLight>(d, c));
      }
      else
        return Fail(line, "unknown keyword '" + key + "'");
    }
    if (!header)
      return Fail(line, "empty file");
    return true;
  }

  std::unique_ptr<ray_tracing::Frame> m_frame;
  std::string m_path;
  std::string m_error;
};

// Writes flattened arrays in the format above; floats with nine significant digits, so reading them back gives
// the same binary32 values.
inline bool WriteSceneFile(std::string const & path, rt_hip_camera const & cam, float const * centersRadii,
                           uint32_t const * materialIndex, uint32_t sphereCount, rt_hip_material const * materials,
                           uint32_t materialCount, rt_hip_light const * lights, uint32_t lightCount)
{
  std::FILE * f = std::fopen(path.c_str(), "w");
  if (f == nullptr)
    return false;
  std::fprintf(f, "rt-scene 1\n");
  std::fprintf(f, "camera %.9g %.9g %.9g %.9g %.9g %.9g\n", cam.position[0], cam.position[1], cam.position[2],
               cam.direction[0], cam.direction[1], cam.direction[2]);
  std::fprintf(f, "lens %.9g %.9g %.9g\n", cam.fov, cam.znear, cam.zfar);
  std::fprintf(f, "background %.9g %.9g %.9g\n", cam.background[0], cam.background[1], cam.background[2]);
  for (uint32_t i = 0; i < materialCount; ++i)
  {
    rt_hip_material const & m = materials[i];
    if (m.type == RT_HIP_MATERIAL_MATTE)
      std::fprintf(f, "material m%u matte %.9g %.9g %.9g\n", i, m.albedo[0], m.albedo[1], m.albedo[2]);
    else if (m.type == RT_HIP_MATERIAL_METAL)
      std::fprintf(f, "material m%u metal %.9g %.9g %.9g %.9g %.9g\n", i, m.albedo[0], m.albedo[1], m.albedo[2],
                   m.roughness, m.refraction);
    else
      std::fprintf(f, "material m%u glass %.9g %.9g %.9g %.9g\n", i, m.albedo[0], m.albedo[1], m.albedo[2],
                   m.refraction);
  }
  for (uint32_t i = 0; i < sphereCount; ++i)
    std::fprintf(f, "sphere %.9g %.9g %.9g %.9g m%u\n", centersRadii[4 * i], centersRadii[4 * i + 1],
                 centersRadii[4 * i + 2], centersRadii[4 * i + 3], materialIndex[i]);
  for (uint32_t i = 0; i < lightCount; ++i)
    std::fprintf(f, "light directional %.9g %.9g %.9g %.9g %.9g %.9g\n", lights[i].direction[0], lights[i].direction[1],
                 lights[i].direction[2], lights[i].color[0], lights[i].color[1], lights[i].color[2]);
  return std::fclose(f) == 0;
}
}  // namespace demo
