// core_emu.cpp -- host build of the trace kernel's per-lane state machine (rt_amd/csrc/rt_core.h), one
// lane at a time.  Test-only: it lets the CPU test-suite compare the kernel's control flow and shading
// with the oracle where no GPU exists.  It is not part of the shipped library and is never used as a
// fallback for it.
#include "rt_prep.h"

#include <vector>

namespace
{
struct HostEnv
{
  const rtc::TraceArgs & A;
  uint32_t total;
  uint32_t next = 0;

  bool any(bool p) { return p; }
  uint32_t fetch_pixel(bool need)
  {
    if (!need)
      return rtc::RT_INVALID;
    uint32_t const p = next++;
    return p < total ? p : rtc::RT_INVALID;
  }
  rtc::f4 sphere(int i) const { return A.spheres[i]; }
  uint32_t sphere_material(int i) const { return A.sphere_mat[i]; }
  void run_pass(rtc::f3 o, const rtc::f3 (&dir)[rtc::RT_RAYS], const float (&a)[rtc::RT_RAYS],
                const float (&tlo_a)[rtc::RT_RAYS], float (&best)[rtc::RT_RAYS], int (&idx)[rtc::RT_RAYS])
  {
    rtc::ray_pass(A.spheres, A.n_spheres, 0u, o, dir, a, tlo_a, best, idx);
  }
};
}  // namespace

extern "C" int emu_render(const float * centers_radii, const uint32_t * material_index, uint32_t n,
                          const rt_hip_material * mats, uint32_t m, const rt_hip_light * lights, uint32_t l,
                          const rt_hip_camera * cam, const rt_hip_trace_params * p, float * rgb,
                          unsigned long long * stats5)
{
  rtc::FlatScene fs;
  rtc::flatten_scene(centers_radii, material_index, n, mats, m, lights, l, fs);
  rtc::TraceArgs a{};
  rtc::prepare_frame(*cam, *p, a);
  a.spheres = fs.spheres.data();
  a.sphere_mat = fs.sphere_mat.data();
  a.materials = fs.materials.data();
  a.lights = fs.lights.data();
  a.n_spheres = n;
  a.n_materials = m;
  a.n_lights = l;
  uint32_t const np = p->width * p->height;
  std::vector<float> planes(3u * (size_t)np, 0.0f);
  a.planes = planes.data();
  a.queue_counter = nullptr;
  a.stats = nullptr;

  std::vector<rtc::f4> stack(rtc::RT_MAX_DEPTH);
  rtc::LaneCounters cnt{};
  HostEnv env{a, np};
  rtc::trace_loop(a, env, stack.data(), 1u, cnt);

  float const inv = 1.0f / static_cast<float>(p->samples_in_row * p->samples_in_row);
  for (uint32_t i = 0; i < np; ++i)
    for (uint32_t c = 0; c < 3; ++c)
      rgb[3u * (size_t)i + c] = planes[(size_t)c * np + i] * inv;
  if (stats5)
  {
    stats5[0] = cnt.samples;
    stats5[1] = cnt.queries;
    stats5[2] = cnt.hits;
    stats5[3] = cnt.shaded;
    stats5[4] = (unsigned long long)cnt.passes * rtc::RT_RAYS * n;
  }
  return 0;
}
