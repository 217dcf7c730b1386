// rt_prep.h -- host-side preparation of the kernel arguments: the per-frame camera quantities of
// Frame::ForEachRayImpl (ray_tracing/frame.cpp:31-43), computed once on the host with the same float
// operations, and the flattened scene records the kernel reads.
#pragma once

#include "rt_core.h"
#include "rt_hip.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace rtc
{
inline f3 cross3(f3 x, f3 y) { return mk3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }

// Fills everything of TraceArgs that depends on the camera and the frame parameters.
inline void prepare_frame(const rt_hip_camera & cam, const rt_hip_trace_params & p, TraceArgs & a)
{
  a.cam_pos = mk3(cam.position[0], cam.position[1], cam.position[2]);
  a.cam_dir = mk3(cam.direction[0], cam.direction[1], cam.direction[2]);
  a.background = mk3(cam.background[0], cam.background[1], cam.background[2]);
  a.cam_right = cross3(mk3(0.0f, 1.0f, 0.0f), a.cam_dir);  // frame.cpp:33
  a.cam_up = cross3(a.cam_dir, a.cam_right);               // frame.cpp:34
  a.znear = cam.znear;
  a.zfar = cam.zfar;
  a.dw = cam.znear / std::tan(0.5f * cam.fov);                                     // frame.cpp:36
  float const aspect = static_cast<float>(p.height) / static_cast<float>(p.width);  // frame.cpp:37
  a.dh = a.dw * aspect;                                                             // frame.cpp:38
  a.cell_x = 2.0f * a.dw / static_cast<float>(p.width);                             // frame.cpp:40
  a.cell_y = 2.0f * a.dh / static_cast<float>(p.height);                            // frame.cpp:41
  a.width = p.width;
  a.height = p.height;
  a.s_row = p.samples_in_row;
  a.s_count = p.samples_in_row * p.samples_in_row;
  a.flags = p.flags;
  a.seed = p.seed;
  a.max_depth = (p.max_depth == 0u || p.max_depth > RT_MAX_DEPTH) ? RT_MAX_DEPTH : p.max_depth;
  a.slice_count = p.slice_count ? p.slice_count : 1u;
  a.slice_index = p.slice_index % a.slice_count;
  a.s_row_mod = p.samples_in_row % a.slice_count;
}

// Slices by runs of pixels: the frame's items are dealt out in runs of run_pixels pixels (every column of them) to the
// slices a.slice_index of a.slice_count in turn.  Sets up this slice's queue (TraceArgs::run_items and
// queue_item, rt_core.h) and switches the slicing by samples off: the slice traces every sample of its columns.
// run_pixels * s_row must be a multiple of the queue's window.
inline void set_pixel_runs(TraceArgs & a, uint32_t run_pixels)
{
  a.run_items = run_pixels * a.s_row;
  a.run_stride = a.slice_count;
  a.run_first = a.slice_index;
  unsigned long long const runs = ((unsigned long long)a.n_items + a.run_items - 1u) / a.run_items;
  unsigned long long const own = runs > a.run_first ? (runs - a.run_first + a.run_stride - 1u) / a.run_stride : 0u;
  unsigned long long positions = own * a.run_items;
  if (own != 0u && (runs - 1u) % a.run_stride == a.run_first)
    positions -= runs * a.run_items - a.n_items;  // the frame's last run is this slice's, and short
  a.queue_items = (uint32_t)positions;
  a.slice_count = 1u;
  a.slice_index = 0u;
  a.s_row_mod = 0u;
}

// Kernels whose waves note their queue reservations in a log of `log_entries` words each and add up their own pixels
// when they run out of work (rt_hip.hip kLogEntries): the reservation sizes for a queue of `queue_items` positions
// worked off by `waves` waves.  A wave makes its share of the reservations and a few dozen shrinking ones at the
// end of the queue; four times the average has to fit its log, so queues of many windows get longer reservations
// (`exact`: tests with small logs skip this, they want full logs).  And however the waves divide the queue, the
// logs together hold every reservation it can be cut into: a wave whose log is full takes no more, every
// reservation but the last takes batch_min positions or more, and a wave's last entry may lie past the end of the
// queue -- so waves * (log_entries - 1) * batch_min must cover the queue.  window: positions per queue window (64);
// max_windows: the most a log word can record; quantum_windows: every reservation is a multiple of this many windows
// (so that it holds whole pixels).  ok = false: no such sizes (the frame keeps the separate passes).
struct LogPlan
{
  bool ok;
  uint32_t batch_items, batch_min;
};
inline LogPlan plan_wave_logs(uint32_t queue_items, uint32_t waves, uint32_t batch_items, uint32_t batch_min,
                              uint32_t log_entries, bool exact, uint32_t window, uint32_t max_windows,
                              uint32_t quantum_windows = 1u)
{
  LogPlan plan = {false, batch_items, batch_min};
  if (queue_items == 0u || waves == 0u || log_entries < 2u || window == 0u || quantum_windows == 0u)
    return plan;
  unsigned long long const windows = ((unsigned long long)queue_items + window - 1u) / window;
  unsigned long long batch_windows = std::max<unsigned long long>(batch_items / window, 1u);
  if (!exact)
  {
    if (log_entries <= 64u)
      return plan;
    unsigned long long const room = (unsigned long long)waves * (log_entries - 64u);
    batch_windows = std::max(batch_windows, (4u * windows + room - 1u) / room);
  }
  unsigned long long const hold = (unsigned long long)waves * (log_entries - 1u);
  unsigned long long min_windows =
      std::max<unsigned long long>(std::max<unsigned long long>(batch_min / window, 1u), (windows + hold - 1u) / hold);
  min_windows = (min_windows + quantum_windows - 1u) / quantum_windows * quantum_windows;
  batch_windows = std::max(batch_windows, min_windows);
  batch_windows = (batch_windows + quantum_windows - 1u) / quantum_windows * quantum_windows;
  if (batch_windows > max_windows)
    return plan;
  plan.ok = true;
  plan.batch_items = (uint32_t)batch_windows * window;
  plan.batch_min = (uint32_t)min_windows * window;
  return plan;
}

// Bound on |D_s - D_c| between the direction through the middle of a sub-sample column and the direction of
// any of its samples: the unnormalised directions differ by at most |up| * cell_y / 2 and are at least
// znear * |dir| long; the slack covers the approximate normalisation of D_c and float rounding.
inline float column_delta(const TraceArgs & a)
{
  double const up = std::sqrt((double)a.cam_up.x * a.cam_up.x + (double)a.cam_up.y * a.cam_up.y +
                              (double)a.cam_up.z * a.cam_up.z);
  double const dir = std::sqrt((double)a.cam_dir.x * a.cam_dir.x + (double)a.cam_dir.y * a.cam_dir.y +
                               (double)a.cam_dir.z * a.cam_dir.z);
  double const d = up * std::fabs((double)a.cell_y) * 0.5 / (std::fabs((double)a.znear) * dir);
  return (float)(d * 1.001 + 1e-6);
}

// True when the early exit of the reverse accumulation loop (demos/demo_frame_cpu.cpp:70-71, |rest| < kEps)
// can never trigger with these materials, so a path's colour is prod(l_i) * prod(e_k ^ k) / 2.  rest is a
// product of at most RT_MAX_DEPTH emissivities: 0.7 for Matte, 1 for Glass, 0.9 (1 - roughness) for Metal
// (ray_tracing/default_materials.cpp:18,30,66); a Metal whose value is below kEps always ends the path instead
// (demos/demo_frame_cpu.cpp:39-45) and never enters a product.
inline bool forward_accumulation_ok(const rt_hip_material * mats, uint32_t m)
{
  double lo = 1.0;
  for (uint32_t i = 0; i < m; ++i)
  {
    float e = 1.0f;
    if (mats[i].type == RT_HIP_MATERIAL_MATTE)
      e = 0.7f;
    else if (mats[i].type == RT_HIP_MATERIAL_METAL)
    {
      e = 0.9f * (1.0f - mats[i].roughness);
      if (std::fabs(e) < RT_EPS)
        continue;
    }
    if (!(e > 0.0f))
      return false;
    lo = std::min(lo, (double)e);
  }
  return std::pow(lo, (double)RT_MAX_DEPTH) >= 2.0 * (double)RT_EPS;
}

// The table jitter_ball reads: the functions of the uniform-in-ball sampler at the 256 cell centres, with the
// C library's float functions.
inline void build_jitter_table(float * tab)
{
  for (uint32_t k = 0; k < RT_JITTER_CELLS; ++k)
  {
    float const u = (float)k * (1.0f / 256.0f) + (0.5f / 256.0f);
    float const z = 1.0f - 2.0f * u;
    float const zz = 1.0f - z * z;
    float const phi = 6.28318530717958647692f * u;
    tab[2u * k] = z;
    tab[2u * k + 1u] = std::sqrt(zz < 0.0f ? 0.0f : zz);
    tab[2u * RT_JITTER_CELLS + 2u * k] = std::cos(phi);
    tab[2u * RT_JITTER_CELLS + 2u * k + 1u] = std::sin(phi);
    tab[4u * RT_JITTER_CELLS + k] = 0.25f * std::cbrt(u);
  }
}

// The uniform grid grid_trace walks.  Spheres much larger than the typical one (a ground sphere) stay out of it
// and are tested for every ray.  Cells are cubes sized for a few spheres each and never smaller than a typical
// sphere; a sphere is listed in every cell its bounding box, inflated by a thousandth of a cell, touches.
struct GridHost
{
  bool valid = false;
  float org[3] = {0, 0, 0};
  float cell = 1.0f;
  uint32_t dim[3] = {1, 1, 1};
  std::vector<uint32_t> start;  // cells + 1
  std::vector<uint32_t> items;
  std::vector<uint32_t> big;
};

inline void build_grid(const float * centers_radii, uint32_t n, GridHost & g)
{
  g = GridHost();
  if (n < 2u)
    return;
  std::vector<float> radii(n);
  for (uint32_t i = 0; i < n; ++i)
  {
    radii[i] = std::fabs(centers_radii[4u * i + 3u]);
    for (uint32_t k = 0; k < 4u; ++k)
      if (!std::isfinite(centers_radii[4u * i + k]))
        return;  // no grid for a scene with non-finite spheres: such a scene is traced sphere by sphere
  }
  std::vector<float> sorted(radii);
  std::nth_element(sorted.begin(), sorted.begin() + n / 2u, sorted.end());
  double const typical = std::max((double)sorted[n / 2u], 1e-6);
  std::vector<uint32_t> small;
  for (uint32_t i = 0; i < n; ++i)
    (radii[i] > 16.0 * typical ? g.big : small).push_back(i);
  if (small.empty() || g.big.size() > 64u)
  {
    g.big.clear();
    return;
  }
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (uint32_t i : small)
    for (int k = 0; k < 3; ++k)
    {
      lo[k] = std::min(lo[k], (double)centers_radii[4u * i + k] - radii[i]);
      hi[k] = std::max(hi[k], (double)centers_radii[4u * i + k] + radii[i]);
    }
  double ext[3];
  for (int k = 0; k < 3; ++k)
  {
    double const pad = 1e-3 * (hi[k] - lo[k]) + 1e-3 * typical;
    lo[k] -= pad;
    hi[k] += pad;
    ext[k] = hi[k] - lo[k];
  }
  // the largest cell that still gives about two cells per sphere, but at least a typical diameter
  auto cells_for = [&](double s) {
    double c = 1.0;
    for (int k = 0; k < 3; ++k)
      c *= std::max(1.0, std::ceil(ext[k] / s));
    return c;
  };
  double s_lo = 2.0 * typical, s_hi = std::max({ext[0], ext[1], ext[2], s_lo});
  double const want = 2.0 * (double)small.size();
  if (cells_for(s_lo) > want)
  {
    for (int it = 0; it < 60; ++it)
    {
      double const mid = 0.5 * (s_lo + s_hi);
      (cells_for(mid) > want ? s_lo : s_hi) = mid;
    }
    s_lo = s_hi;
  }
  s_lo = std::max(s_lo, std::max({ext[0], ext[1], ext[2]}) / 1024.0);  // at most 1024 cells along an axis
  g.cell = (float)s_lo;
  if (!(g.cell > 0.0f) || !std::isfinite(g.cell))
    return;
  size_t cells = 1;
  for (int k = 0; k < 3; ++k)
  {
    g.org[k] = (float)lo[k];
    g.dim[k] = (uint32_t)std::max(1.0, std::ceil(ext[k] / (double)g.cell));
    cells *= g.dim[k];
  }
  double const inflate = 1e-3 * (double)g.cell;
  auto range = [&](uint32_t i, int k, int & a, int & b) {
    double const c = (double)centers_radii[4u * i + k] - (double)g.org[k];
    a = (int)std::floor((c - radii[i] - inflate) / (double)g.cell);
    b = (int)std::floor((c + radii[i] + inflate) / (double)g.cell);
    a = std::max(a, 0);
    b = std::min(b, (int)g.dim[k] - 1);
  };
  std::vector<uint32_t> count(cells + 1u, 0u);
  for (int pass = 0; pass < 2; ++pass)
  {
    for (uint32_t i : small)  // ascending sphere index, so every cell's list comes out ascending
    {
      int x0, x1, y0, y1, z0, z1;
      range(i, 0, x0, x1);
      range(i, 1, y0, y1);
      range(i, 2, z0, z1);
      for (int z = z0; z <= z1; ++z)
        for (int y = y0; y <= y1; ++y)
          for (int x = x0; x <= x1; ++x)
          {
            size_t const c = ((size_t)z * g.dim[1] + (size_t)y) * g.dim[0] + (size_t)x;
            if (pass == 0)
              count[c + 1u]++;
            else
              g.items[count[c]++] = i;
          }
    }
    if (pass == 0)
    {
      for (size_t c = 0; c < cells; ++c)
        count[c + 1u] += count[c];
      g.start = count;
      g.items.resize(count[cells]);
    }
  }
  g.valid = true;
}

// Scene records in the layout the kernel reads.
struct FlatScene
{
  std::vector<f4> spheres;           // (cx, cy, cz, r*r)
  std::vector<f4> filter;            // (centre - scene centre, rw^2 - (1 - m - slack) |centre - scene centre|^2): the
                                     // shade filter's record of a sphere, rt_core.h filter_sphere_2
  std::vector<float> aux_rw;         // the widened radius rw
  std::vector<float> aux_cone;       // cone_constant of the sphere for the first light's shadow cone
  f3 centre;                         // the scene centre: mean centre of the spheres of ordinary size
  std::vector<uint32_t> sphere_mat;  // material index per sphere
  std::vector<f4> materials;         // RT_MAT_RECORDS per material
  std::vector<f4> lights;            // RT_LIGHT_RECORDS per light
};

// The cone around the direction to a light that holds every shadow ray DirectionalLight::TraceLight can shoot
// (ray_tracing/directional_light.cpp:20: normalize(-L + ballRand(0.25))); see ShadowCone in rt_core.h.  The
// jitter vector is at most 0.25 long, so the rays stay within asin(0.25 / |L|) of -L; cos and sin of that angle
// are moved to the safe side by a thousandth, which dwarfs the rounding of the kernel's float arithmetic.
// use_cone = false (or a light shorter than about 0.28) gives the cone that holds every direction.
inline ShadowCone shadow_cone(const float * light_direction, bool use_cone)
{
  ShadowCone c;
  c.axis = mk3(0.0f, 1.0f, 0.0f);
  c.cos_m = 0.0f;
  c.sin_m = 3.0e38f;
  c.cos2_m = 0.0f;
  c.oa = 0.0f;
  if (light_direction == nullptr)
    return c;
  double const x = light_direction[0], y = light_direction[1], z = light_direction[2];
  double const len = std::sqrt(x * x + y * y + z * z);
  if (!(len > 1e-20) || !std::isfinite(len))
    return c;
  c.axis = mk3((float)(-x / len), (float)(-y / len), (float)(-z / len));
  double const sin_t = 0.25 * (1.0 + 1e-5) / len;
  if (!use_cone || !(sin_t < 0.9))
    return c;
  double const sin_m = sin_t * 1.001 + 1e-4;
  c.sin_m = (float)sin_m;
  c.cos_m = (float)(std::sqrt(1.0 - sin_m * sin_m) * 0.999);
  c.cos2_m = c.cos_m * c.cos_m;
  return c;
}

// The part of a shadow-cone test that depends on the sphere alone (rt_core.h filter_sphere_2): (dot(C, axis) +
// sin_m rw) / cos_m.  The kernel subtracts dot(O, axis) / cos_m from it in float, so the value is raised by more
// than the rounding errors of storing it and of that subtraction (each at most 6e-8 of the terms' magnitudes): the
// test stays on the accepting side.  The cone that holds every direction gets a value no subtraction brings below
// zero.
inline float cone_constant(const double * centre, float rw, const ShadowCone & cone)
{
  if (!(cone.sin_m < 1e30f) || !(cone.cos_m > 0.0f))
    return 3.0e38f;
  double const tx = (double)centre[0] * cone.axis.x, ty = (double)centre[1] * cone.axis.y, tz = (double)centre[2] * cone.axis.z;
  double const tr = (double)cone.sin_m * rw;
  double const k = tx + ty + tz + tr;
  double const raised = (k + 4e-7 * (std::fabs(tx) + std::fabs(ty) + std::fabs(tz) + tr)) / cone.cos_m;
  return std::nextafter(static_cast<float>(raised), INFINITY);
}

RT_HD f4 mk4(float x, float y, float z, float w)
{
  f4 r;
  r.x = x;
  r.y = y;
  r.z = z;
  r.w = w;
  return r;
}

// shadow_cones = false: every light gets the cone that holds all directions, so the shadow rays are filtered one
// by one against every sphere (for comparisons; same frames).
inline void flatten_scene(const float * centers_radii, const uint32_t * material_index, uint32_t n,
                          const rt_hip_material * mats, uint32_t m, const rt_hip_light * lights, uint32_t l,
                          FlatScene & out, bool shadow_cones = true)
{
  out.spheres.resize(n);
  out.filter.resize(n);
  out.aux_rw.resize(n);
  out.aux_cone.resize(n);
  ShadowCone const cone0 = shadow_cone(l != 0u ? lights[0].direction : nullptr, shadow_cones);
  // The scene centre: the mean centre of the spheres no more than sixteen times the median radius (a ground sphere
  // of radius 1000 says nothing about where the scene is), as a float vector, since the kernel subtracts it from
  // its origins in float.
  out.centre = mk3(0.0f, 0.0f, 0.0f);
  if (n != 0u)
  {
    std::vector<float> radii(n);
    for (uint32_t i = 0; i < n; ++i)
      radii[i] = std::fabs(centers_radii[4u * i + 3u]);
    std::nth_element(radii.begin(), radii.begin() + n / 2u, radii.end());
    double const limit = 16.0 * radii[n / 2u];
    double sum[3] = {0.0, 0.0, 0.0};
    uint32_t used = 0;
    for (uint32_t i = 0; i < n; ++i)
    {
      const float * s = centers_radii + 4u * i;
      if (std::fabs(s[3]) <= limit && std::isfinite(s[0]) && std::isfinite(s[1]) && std::isfinite(s[2]))
      {
        sum[0] += s[0];
        sum[1] += s[1];
        sum[2] += s[2];
        used += 1u;
      }
    }
    if (used != 0u)
      out.centre = mk3((float)(sum[0] / used), (float)(sum[1] / used), (float)(sum[2] / used));
  }
  out.sphere_mat.assign(material_index, material_index + n);
  for (uint32_t i = 0; i < n; ++i)
  {
    const float * s = centers_radii + 4u * i;
    float const r2 = s[3] * s[3];  // m_radius * m_radius, sphere.cpp:68
    out.spheres[i] = mk4(s[0], s[1], s[2], r2);
    // the widened radius, rounded up, and its square, rounded up
    double const widened = std::fabs(static_cast<double>(s[3])) * (1.0 + RT_FILTER_MARGIN);
    float const rw = std::max(std::nextafter(static_cast<float>(widened), INFINITY), 1e-30f);
    // the sphere relative to the scene centre (the differences of two floats, exact in double, rounded once) and
    // its constant rw^2 - (1 - m) |C'|^2 + slack |C'|^2, taken from the rounded C' the kernel multiplies with and
    // rounded up
    double const cd[3] = {(double)s[0] - out.centre.x, (double)s[1] - out.centre.y, (double)s[2] - out.centre.z};
    float const cf[3] = {(float)cd[0], (float)cd[1], (float)cd[2]};
    double const cc = (double)cf[0] * cf[0] + (double)cf[1] * cf[1] + (double)cf[2] * cf[2];
    double const q = (double)rw * rw - (1.0 - RT_FILTER_MARGIN - RT_FILTER_SLACK) * cc;
    out.filter[i] = mk4(cf[0], cf[1], cf[2], std::nextafter(static_cast<float>(q), INFINITY));
    out.aux_rw[i] = rw;
    double const cfd[3] = {cf[0], cf[1], cf[2]};
    out.aux_cone[i] = cone_constant(cfd, rw, cone0);
  }
  // Per-material constants of Scatter and GetSpecular, computed here once with the float operations the
  // reference uses per hit (ray_tracing/default_materials.cpp:30,50,61-62; directional_light.cpp:49-56).
  out.materials.resize((size_t)RT_MAT_RECORDS * m);
  for (uint32_t i = 0; i < m; ++i)
  {
    float type_bits;
    uint32_t const t = mats[i].type;
    std::memcpy(&type_bits, &t, 4);
    float const refraction = mats[i].refraction;
    float const f0 = (1.0f - refraction) / (1.0f + refraction);
    float const f2 = f0 * f0;
    float const inv_refraction = 1.0f / refraction;
    float const rough = (mats[i].roughness < 0.03f) ? 0.03f : mats[i].roughness;  // std::max(roughness, 0.03f)
    float const r2 = rough * rough;
    float const e_metal = 0.9f * (1.0f - mats[i].roughness);
    out.materials[RT_MAT_RECORDS * i] = mk4(mats[i].albedo[0], mats[i].albedo[1], mats[i].albedo[2], mats[i].roughness);
    out.materials[RT_MAT_RECORDS * i + 1u] = mk4(refraction, type_bits, f2, inv_refraction);
    out.materials[RT_MAT_RECORDS * i + 2u] = mk4(r2, 4.0f * r2, e_metal, 0.0f);
  }
  out.lights.resize((size_t)RT_LIGHT_RECORDS * l);
  for (uint32_t i = 0; i < l; ++i)
  {
    ShadowCone const cone = shadow_cone(lights[i].direction, shadow_cones);
    out.lights[RT_LIGHT_RECORDS * i] = mk4(lights[i].direction[0], lights[i].direction[1], lights[i].direction[2], cone.cos_m);
    out.lights[RT_LIGHT_RECORDS * i + 1u] = mk4(lights[i].color[0], lights[i].color[1], lights[i].color[2], cone.sin_m);
    out.lights[RT_LIGHT_RECORDS * i + 2u] = mk4(cone.axis.x, cone.axis.y, cone.axis.z, 0.0f);
  }
}
}  // namespace rtc
