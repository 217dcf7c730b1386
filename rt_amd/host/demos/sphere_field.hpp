// sphere_field.hpp -- helpers shared by the sphere demos: unique integer grid positions and the common
// ground sphere + light (reference: demos/pretty_spheres.hpp:42-56,77-83; demos/glass_spheres.hpp:42-56,67-73).
#pragma once

#include "scene_random.hpp"

#include "ray_tracing/default_materials.hpp"
#include "ray_tracing/directional_light.hpp"
#include "ray_tracing/frame.hpp"
#include "ray_tracing/sphere.hpp"

#include <memory>
#include <set>
#include <utility>

namespace demo
{
// Draws (x, z) in [-5, 5]^2 until the cell is free (at most 5000 tries, then reuses the cell).
class GridPositions
{
public:
  std::pair<int, int> Next(SceneRandom & rnd)
  {
    std::pair<int, int> p;
    int tries = 0;
    do
    {
      int const x = rnd.Int(-5, 5);
      int const z = rnd.Int(-5, 5);
      p = std::make_pair(x, z);
      ++tries;
    } while (m_taken.count(p) != 0 && tries < 5000);
    m_taken.insert(p);
    return p;
  }

private:
  std::set<std::pair<int, int>> m_taken;
};

inline void AddGroundAndSun(ray_tracing::Frame & frame, float groundAlbedo)
{
  using namespace ray_tracing;
  frame.AddObject(std::make_unique<Sphere>(
      vec3(0.0f, -1001.0f, 0.0f), 1000.0f,
      std::make_shared<material::Matte>(vec3(groundAlbedo, groundAlbedo, groundAlbedo))));
  frame.AddLightSource(std::make_unique<DirectionalLight>(normalize(vec3(-0.4f, -1.0f, 0.6f)),
                                                           vec3(1.0f, 1.0f, 1.0f)));
}
}  // namespace demo
