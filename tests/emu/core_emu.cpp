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
  static constexpr uint32_t kFilterUnroll = 4;
  static constexpr uint32_t kPrimaryUnroll = 4;
  static constexpr bool kLibraryMath = true;  // (the host build has no other)
  static constexpr bool kCompactsRays = false;
  const rtc::TraceArgs & A;
  uint32_t total;
  uint32_t tile_size;
  uint32_t next = 0;
  uint32_t cur_count = 0;
  const rtc::f4 * cur = nullptr;
  const float * aux_src = nullptr;  // A.aux_cone when the loop is the one specialised for one light, else A.aux_rw
  const float * cur_aux = nullptr;

  // Sample-sharing test mode.  The kernel shares samples between the lanes of a wave once its queue is used
  // up; this one-lane build exercises the same bookkeeping (shortened columns, shared-sample entries, split
  // records, the fold) by making every column hand its last samples to a list that the lane works off when the
  // queue is empty.
  struct SharedSample
  {
    uint32_t region_base, pixel, sx, cm0, cm1, sy;
    float sdx, dyp;
  };
  bool share_test = false;
  std::vector<SharedSample> waiting;
  std::vector<rtc::u4> splits;
  bool endgame() const { return share_test; }
  void share(rtc::Lane & L)
  {
    uint32_t const region_base = (uint32_t)splits.size() * A.share_depth;  // one region per split column
    uint32_t const spare = rtc::spare_samples(A, L);
    if (spare != 0u)
    {
      // a varying number, so that columns are cut at different places (and some not at all)
      uint32_t const give = (L.pixel + L.sx) % (spare + 1u);
      for (uint32_t j = 0; j < give; ++j)
        waiting.push_back({region_base, L.pixel, L.sx, L.cm0, L.cm1, rtc::spare_sample_sy(A, L.sy, L.sy_end, j), L.sdx, L.dyp});
      if (give != 0u)
        L.sy_end = rtc::spare_sample_sy(A, L.sy, L.sy_end, give - 1u);
    }
    else if (!L.alive && !waiting.empty())
    {
      SharedSample const w = waiting.back();
      waiting.pop_back();
      rtc::take_shared_sample(A, L, w.region_base, w.pixel, w.sx, w.sdx, w.dyp, w.cm0, w.cm1, w.sy);
    }
  }
  void record_split(uint32_t slot, uint32_t first_shared_sy)
  {
    splits.push_back({slot, first_shared_sy, (uint32_t)splits.size() * A.share_depth, 0u});
  }

  bool any(bool p) { return p; }
  int count(bool p) { return p ? 1 : 0; }
  bool any_wave(bool p) { return p; }
  uint32_t column_sx(const rtc::Lane & L) const { return L.sx; }
  void finish_columns(rtc::Lane & L, bool finishing)
  {
    if (finishing)
      rtc::store_column(A, L, *this);
  }
  void fetch_item(const rtc::Lane &, bool need, rtc::ItemInfo & info)
  {
    info.pos = rtc::RT_INVALID;
    info.stalled = false;
    info.prepared = false;
    info.filter_tests = 0;
    if (!need || next >= A.n_items)
      return;
    info.pos = next++;
    info.slot = info.pos;
    if (A.column_masks != 0u)
    {
      rtc::ColumnEntry const e = rtc::prepare_column(A, A.camtab, info.pos);
      info.prepared = true;
      info.sdx = e.sdx;
      info.dyp = e.dyp;
      info.cm0 = e.cm0;
      info.cm1 = e.cm1;
      info.pixel = e.pixel;
      info.sx = e.sx;
      info.sy0 = e.sy0;
      info.filter_tests = 1;
    }
  }
  rtc::f4 sphere(int i) const { return A.spheres[i]; }
  uint32_t sphere_material(int i) const { return A.sphere_mat[i]; }
  rtc::f4 material(uint32_t m, uint32_t k) const { return A.materials[rtc::RT_MAT_RECORDS * m + k]; }
  rtc::f4 light(uint32_t l, uint32_t k) const { return A.lights[rtc::RT_LIGHT_RECORDS * l + k]; }
  const float * jitter_table() const { return A.jitter_tab; }
  uint32_t tiles() const { return (A.n_padded + tile_size - 1) / tile_size; }
  uint32_t tile_base(uint32_t k) const { return k * tile_size; }
  uint32_t load_tile(uint32_t k, bool primary)
  {
    uint32_t const base = k * tile_size;
    cur = (primary ? A.camtab : A.filter) + base;
    cur_aux = aux_src + base;
    cur_count = (A.n_padded - base < tile_size) ? (A.n_padded - base) : tile_size;
    return cur_count;
  }
  const rtc::f4 * tile_data(bool) const { return cur; }
  const float * tile_aux() const { return cur_aux; }
};
}  // namespace

// tile_size: records per staged tile (exercises the multi-tile path when smaller than the scene).
// mode: bit 0 = column masks (scenes of at most 64 spheres, one tile), bit 1 = forward accumulation,
// bit 2 = rays walk the uniform grid, bit 3 = sample-sharing test mode (see HostEnv), bit 4 = the general loop
// even for a scene with one light, bit 5 = no shadow cones (shadow rays filtered one by one).
extern "C" int emu_render_mode(const float * centers_radii, const uint32_t * material_index, uint32_t n,
                               const rt_hip_material * mats, uint32_t m, const rt_hip_light * lights, uint32_t l,
                               const rt_hip_camera * cam, const rt_hip_trace_params * p, uint32_t tile_size,
                               uint32_t mode, float * rgb, unsigned long long * stats6);

// Columns of the last frame whose sum was completed by the fold (sample-sharing test mode).
static unsigned long long g_last_split_count = 0;
extern "C" unsigned long long emu_last_split_count() { return g_last_split_count; }

// The queue of slice `index` of `count` when a frame of n_pixels x s_row items is divided by runs of run_pixels
// pixels (rt_prep.h set_pixel_runs, rt_core.h queue_item): writes the item of every queue position to `items`
// (room for n_pixels * s_row entries) and returns the number of positions.
extern "C" uint32_t emu_pixel_run_queue(uint32_t n_pixels, uint32_t s_row, uint32_t run_pixels, uint32_t index, uint32_t count,
                                        uint32_t * items)
{
  rtc::TraceArgs a;
  std::memset(&a, 0, sizeof(a));
  a.s_row = s_row;
  a.n_items = n_pixels * s_row;
  a.queue_items = a.n_items;
  a.slice_count = count ? count : 1u;
  a.slice_index = index % a.slice_count;
  rtc::set_pixel_runs(a, run_pixels);
  for (uint32_t k = 0; k < a.queue_items; ++k)
    items[k] = rtc::queue_item(a, k);
  return a.queue_items;
}

// rt_prep.h plan_wave_logs: out = {ok, batch_items, batch_min}.
extern "C" void emu_plan_wave_logs(uint32_t queue_items, uint32_t waves, uint32_t batch_items, uint32_t batch_min,
                                   uint32_t log_entries, int exact, uint32_t quantum_windows, uint32_t * out3)
{
  rtc::LogPlan const plan = rtc::plan_wave_logs(queue_items, waves, batch_items, batch_min, log_entries, exact != 0, 64u, 64u,
                                                quantum_windows);
  out3[0] = plan.ok ? 1u : 0u;
  out3[1] = plan.batch_items;
  out3[2] = plan.batch_min;
}

extern "C" int emu_forward_ok(const rt_hip_material * mats, uint32_t m)
{
  return rtc::forward_accumulation_ok(mats, m) ? 1 : 0;
}

extern "C" int emu_render(const float * centers_radii, const uint32_t * material_index, uint32_t n,
                          const rt_hip_material * mats, uint32_t m, const rt_hip_light * lights, uint32_t l,
                          const rt_hip_camera * cam, const rt_hip_trace_params * p, uint32_t tile_size,
                          float * rgb, unsigned long long * stats6)
{
  return emu_render_mode(centers_radii, material_index, n, mats, m, lights, l, cam, p, tile_size, 0u, rgb, stats6);
}

extern "C" int emu_render_mode(const float * centers_radii, const uint32_t * material_index, uint32_t n,
                          const rt_hip_material * mats, uint32_t m, const rt_hip_light * lights, uint32_t l,
                               const rt_hip_camera * cam, const rt_hip_trace_params * p, uint32_t tile_size,
                               uint32_t mode, float * rgb, unsigned long long * stats6)
{
  rtc::FlatScene fs;
  rtc::flatten_scene(centers_radii, material_index, n, mats, m, lights, l, fs, (mode & 32u) == 0u);
  rtc::TraceArgs a{};
  rtc::prepare_frame(*cam, *p, a);
  if (l != 0u)
  {
    a.cone_axis = rtc::mk3(fs.lights[2].x, fs.lights[2].y, fs.lights[2].z);
    a.cone_cos = fs.lights[0].w;
    a.cone_sin = fs.lights[1].w;
    a.cone_inv_cos = a.cone_cos > 0.0f ? 1.0f / a.cone_cos : 1.0f;
  }
  uint32_t const padded = static_cast<uint32_t>(fs.filter.size());
  bool const column_masks = (mode & 1u) != 0u && padded > 0 && padded <= 64;
  a.column_masks = column_masks ? 1u : 0u;
  a.col_delta = column_masks ? rtc::column_delta(a) : 0.0f;
  std::vector<rtc::f4> camtab(padded);
  for (uint32_t i = 0; i < padded; ++i)
  {
    float const dx = fs.spheres[i].x - a.cam_pos.x, dy = fs.spheres[i].y - a.cam_pos.y,
                dz = fs.spheres[i].z - a.cam_pos.z;
    float const len = std::sqrt(dx * dx + dy * dy + dz * dz) * 1.000001f;
    camtab[i] = rtc::mk4(dx, dy, dz, rtc::filter_bound(dx, dy, dz, fs.aux_rw[i] * fs.aux_rw[i]) - len * a.col_delta);
  }
  std::vector<float> jitter(rtc::RT_JITTER_FLOATS);
  rtc::build_jitter_table(jitter.data());
  a.jitter_tab = jitter.data();
  a.spheres = fs.spheres.data();
  a.filter = fs.filter.data();
  a.aux_rw = fs.aux_rw.data();
  a.aux_cone = fs.aux_cone.data();
  a.centre = fs.centre;
  a.camtab = camtab.data();
  a.n_padded = padded;
  a.sphere_mat = fs.sphere_mat.data();
  a.materials = fs.materials.data();
  a.lights = fs.lights.data();
  a.n_spheres = n;
  a.n_materials = m;
  a.n_lights = l;
  uint32_t const np = p->width * p->height;
  uint32_t const s_row = p->samples_in_row;
  a.n_items = np * s_row;
  std::vector<float> partial(3u * (size_t)a.n_items, 0.0f);
  a.partial = partial.data();
  a.queue_counter = nullptr;
  a.stats = nullptr;

  std::vector<rtc::f4> stack(rtc::RT_MAX_DEPTH);
  rtc::LaneCounters cnt{};
  HostEnv env{a, a.n_items, (tile_size && !column_masks) ? tile_size : (padded ? padded : 4u)};
  std::vector<float> shared;
  if ((mode & 8u) != 0u)
  {
    env.share_test = true;
    a.share_depth = s_row;
    shared.assign(3u * (size_t)a.n_items * a.share_depth, 0.0f);
    // the colours of shared samples are addressed relative to the column sums: one buffer holds both here too
    partial.resize(partial.size() + shared.size(), 0.0f);
    a.partial = partial.data();
    a.shared = partial.data() + 3u * (size_t)a.n_items;
    a.shared_off = 3u * (size_t)a.n_items;
  }
  // the forward-accumulating loop comes specialised for one light, as in the library (mode bit 4 keeps the
  // general loop for such scenes)
  bool const one_light = l == 1u && (mode & 16u) == 0u;
  // (only the forward-accumulating loops come specialised for one light)
  env.aux_src = (one_light && (mode & 2u) != 0u) ? a.aux_cone : a.aux_rw;
  rtc::GridHost grid;
  if ((mode & 4u) != 0u)
  {
    rtc::build_grid(centers_radii, n, grid);
    if (!grid.valid)
      return 3;
    a.grid_start = grid.start.data();
    a.grid_items = grid.items.data();
    a.grid_big = grid.big.data();
    a.grid_big_count = static_cast<uint32_t>(grid.big.size());
    for (int k = 0; k < 3; ++k)
      a.grid_dim[k] = grid.dim[k];
    a.grid_org = rtc::mk3(grid.org[0], grid.org[1], grid.org[2]);
    a.grid_cell = grid.cell;
    a.grid_inv_cell = 1.0f / grid.cell;
    if ((mode & 2u) != 0u && one_light)
      rtc::trace_loop<true, true, true>(a, env, stack.data(), 1u, cnt);
    else if ((mode & 2u) != 0u)
      rtc::trace_loop<true, true, false>(a, env, stack.data(), 1u, cnt);
    else
      rtc::trace_loop<false, true, false>(a, env, stack.data(), 1u, cnt);
  }
  else if ((mode & 2u) != 0u && one_light)
    rtc::trace_loop<true, false, true>(a, env, stack.data(), 1u, cnt);
  else if ((mode & 2u) != 0u)
    rtc::trace_loop<true, false, false>(a, env, stack.data(), 1u, cnt);
  else
    rtc::trace_loop<false, false, false>(a, env, stack.data(), 1u, cnt);

  g_last_split_count = (unsigned long long)env.splits.size();
  for (rtc::u4 const & rec : env.splits)
    rtc::fold_split(partial.data(), a.shared, rec, s_row, a.slice_count);
  if (env.share_test && !env.waiting.empty())
    return 4;

  // accumulate (columns in increasing sx) + resolve, as colsum_kernel and resolve_kernel do
  float const inv = 1.0f / static_cast<float>(s_row * s_row);
  for (uint32_t i = 0; i < np; ++i)
    for (uint32_t c = 0; c < 3; ++c)
    {
      float sum = 0.0f;
      for (uint32_t sx = 0; sx < s_row; ++sx)
        sum += partial[3u * ((size_t)i * s_row + sx) + c];
      rgb[3u * (size_t)i + c] = sum * inv;
    }
  if (stats6)
  {
    stats6[0] = cnt.samples;
    stats6[1] = cnt.queries;
    stats6[2] = cnt.hits;
    stats6[3] = cnt.shaded;
    stats6[4] = ((unsigned long long)cnt.filter_tests + cnt.prepared) * n;
    stats6[5] = cnt.exact_tests;
  }
  return 0;
}
