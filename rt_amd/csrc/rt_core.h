// rt_core.h -- per-lane path state machine of the gfx950 trace kernel.
//
// One lane carries one camera sample at a time.  Every trip round the main loop a lane runs one
// "ray pass": RT_RAYS (6) rays that share an origin are tested against every sphere of the scene.
// A pass is either
//   * a shade pass: the 5 jittered shadow rays of DirectionalLight::TraceLight
//     (ray_tracing/directional_light.cpp:15-35) plus the scattered ray whose nearest hit starts the
//     next bounce (demos/demo_frame_cpu.cpp:33,50) -- all six leave the same hit point; or
//   * a primary pass: the next 6 sub-samples of the lane's pixel (ray_tracing/frame.cpp:56-66) -- all
//     six leave the camera.
// Sharing the origin lets (origin - centre) and c = |origin - centre|^2 - r^2 be computed once per
// sphere for six rays.  A lane whose path ends starts its next sample at once, and a lane whose
// pixel is finished pulls a new pixel from the frame-wide queue, so the wave keeps all 64 lanes on
// live rays whatever the bounce depth of its neighbours.
//
// The file is plain C++ over a handful of macros so the same state machine also compiles for the
// host, one lane at a time, which is how the logic is debugged without a GPU (tests/emu).
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_FN __device__ __forceinline__
#define RT_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#include <cstring>
#define RT_FN static inline
#define RT_HD static inline
#endif

namespace rtc
{
constexpr int RT_RAYS = 6;           // rays per pass (5 shadow + 1 scatter, or 6 primaries)
constexpr int RT_LIGHT_SAMPLES = 5;  // ray_tracing/directional_light.cpp:13
constexpr uint32_t RT_MAX_DEPTH = 10;  // demos/demo_frame_cpu.cpp:10
constexpr float RT_EPS = 1e-5f;      // ray_tracing/types.hpp:9
constexpr uint32_t RT_INVALID = 0xFFFFFFFFu;

constexpr uint32_t RT_MATTE = 0, RT_METAL = 1, RT_GLASS = 2;  // ray_tracing/types.hpp:15-17
constexpr uint32_t RT_FLAG_DETERMINISTIC = 1u;

struct f3
{
  float x, y, z;
};
struct f4
{
  float x, y, z, w;
};

RT_HD f3 mk3(float x, float y, float z)
{
  f3 r;
  r.x = x;
  r.y = y;
  r.z = z;
  return r;
}
RT_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
RT_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_HD float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// ---- hardware math: one instruction each on gfx950 (about 1 ulp); libm on the host ----
#if defined(__HIPCC__)
RT_FN float hw_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
RT_FN float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
RT_FN float hw_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
RT_FN float hw_sin_rev(float r) { return __builtin_amdgcn_sinf(r); }  // sin(2*pi*r)
RT_FN float hw_cos_rev(float r) { return __builtin_amdgcn_cosf(r); }
RT_FN float hw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
RT_FN float hw_log2(float x) { return __builtin_amdgcn_logf(x); }
RT_FN float hw_min(float a, float b) { return fminf(a, b); }
RT_FN float hw_max(float a, float b) { return fmaxf(a, b); }
RT_FN float hw_abs(float a) { return fabsf(a); }
RT_FN uint32_t hw_mulhi(uint32_t a, uint32_t b) { return __umulhi(a, b); }
#else
RT_FN float hw_sqrt(float x) { return sqrtf(x); }
RT_FN float hw_rcp(float x) { return 1.0f / x; }
RT_FN float hw_rsq(float x) { return 1.0f / sqrtf(x); }
RT_FN float hw_sin_rev(float r) { return sinf(6.28318530717958647692f * r); }
RT_FN float hw_cos_rev(float r) { return cosf(6.28318530717958647692f * r); }
RT_FN float hw_exp2(float x) { return exp2f(x); }
RT_FN float hw_log2(float x) { return log2f(x); }
RT_FN float hw_min(float a, float b) { return fminf(a, b); }
RT_FN float hw_max(float a, float b) { return fmaxf(a, b); }
RT_FN float hw_abs(float a) { return fabsf(a); }
RT_FN uint32_t hw_mulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
#endif

RT_FN f3 normalize3(f3 v) { return v * hw_rsq(dot3(v, v)); }
RT_FN float pow5(float x)
{
  float const x2 = x * x;
  return x2 * x2 * x;
}
RT_FN float clampf(float x, float lo, float hi) { return hw_min(hw_max(x, lo), hi); }
RT_FN f3 mix3(f3 x, f3 y, float a) { return x * (1.0f - a) + y * a; }

// ---- Philox4x32-10, key = (pixel, sample ^ seed), counter = (bounce, block, 0, 0) ----
struct u4
{
  uint32_t x, y, z, w;
};

RT_FN u4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
  for (int r = 0; r < 10; ++r)
  {
    uint32_t const hi0 = hw_mulhi(0xD2511F53u, c0);
    uint32_t const lo0 = 0xD2511F53u * c0;
    uint32_t const hi1 = hw_mulhi(0xCD9E8D57u, c2);
    uint32_t const lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  u4 o;
  o.x = c0;
  o.y = c1;
  o.z = c2;
  o.w = c3;
  return o;
}

RT_FN float uniform01(uint32_t bits) { return (float)(bits >> 8) * (1.0f / 16777216.0f); }

// Uniform point in a ball of radius R from three uniforms: direction uniform on the sphere, radius
// R * u^(1/3).  Same law as glm::ballRand with a fixed draw count, so no lane waits on a rejection loop.
RT_FN f3 ball3(float radius, float u1, float u2, float u3)
{
  float const z = 1.0f - 2.0f * u1;
  float const s = hw_sqrt(hw_max(0.0f, 1.0f - z * z));
  float const rad = radius * hw_exp2(hw_log2(u3) * (1.0f / 3.0f));
  return mk3((s * hw_cos_rev(u2)) * rad, (s * hw_sin_rev(u2)) * rad, z * rad);
}

// ---- kernel arguments (wave-uniform) ----
struct TraceArgs
{
  // scene, device memory
  const f4 * spheres;          // n_spheres x (cx, cy, cz, r*r)
  const uint32_t * sphere_mat; // n_spheres
  const f4 * materials;        // n_materials x 2: (albedo rgb, roughness), (refraction, type bits, 0, 0)
  const f4 * lights;           // n_lights x 2: (direction xyz, 0), (colour rgb, 0)
  uint32_t n_spheres, n_materials, n_lights;
  // camera, precomputed on the host exactly as Frame::ForEachRayImpl does (ray_tracing/frame.cpp:33-41)
  f3 cam_pos, cam_dir, cam_right, cam_up, background;
  float znear, zfar, dw, dh, cell_x, cell_y;
  // frame
  uint32_t width, height, s_row, s_count;  // s_count = s_row * s_row
  uint32_t flags, seed, max_depth, slice_index, slice_count;
  // outputs
  float * planes;            // 3 planes of width*height un-normalised sums
  uint32_t * queue_counter;  // next pixel to hand out
  unsigned long long * stats; // 5 counters: samples, ray queries, hit queries, shaded bounces, sphere tests
};

struct LaneCounters
{
  uint32_t samples, queries, hits, shaded, passes;
};

// Per-lane state.  Arrays are only ever indexed by compile-time constants (unrolled loops), so they
// stay in registers.
struct Lane
{
  // pixel being accumulated
  uint32_t pixel;
  uint32_t s_gen;  // next owned sample index not yet put into a primary batch (>= s_count: none left)
  float dxp, dyp;  // dx, dy of the pixel's top-left corner (ray_tracing/frame.cpp:49,52)
  f3 sum;
  bool alive, need_pixel;
  // primary batch: nearest hits of up to 6 consecutive owned samples, consumed front first
  float bt[RT_RAYS];
  int bi[RT_RAYS];
  uint32_t b_cnt;
  uint32_t b_s;  // sample index of the front entry
  // current path
  uint32_t sample;  // sample index (RNG key)
  uint32_t depth;   // entries pushed so far ("scatterResultsCount")
  bool has_hit;     // a hit waits to be shaded
  f3 hp, hn, hd;    // hit point, normal, incoming direction
  int hidx;         // sphere hit
  // shading in flight (between ray set-up and the end of the last light's pass)
  int mode;  // 0 = none, 1 = primary pass, 2 = shade pass
  uint32_t li;  // light of the current shade pass
  f3 spec, diffuse;
  float e;
  float w[RT_LIGHT_SAMPLES];  // max(0, dot(shadow dir, n))
  f3 m_albedo;
  float m_refraction;
  f3 sdir;  // scattered direction
  float s_a;  // dot(sdir, sdir)
  float nx_ta;
  int nx_i;  // nearest hit of the scattered ray (from the first light's pass)
};

constexpr int MODE_NONE = 0, MODE_PRIMARY = 1, MODE_SHADE = 2;

// Frame::ForEachRayImpl, ray_tracing/frame.cpp:58-62: direction of sub-sample `sample` = sx * S + sy.
RT_FN f3 primary_dir(const TraceArgs & A, float dxp, float dyp, uint32_t sample)
{
  uint32_t const sx = sample / A.s_row;
  uint32_t const sy = sample - sx * A.s_row;
  float const sdx = dxp + A.cell_x * (float)sx / (float)A.s_row;
  float const sdy = dyp - A.cell_y * (float)sy / (float)A.s_row;
  f3 const v = (A.cam_dir * A.znear + A.cam_up * sdy) + A.cam_right * sdx;
  return normalize3(v);
}

RT_FN void begin_pixel(const TraceArgs & A, Lane & L, uint32_t pixel)
{
  uint32_t const y = pixel / A.width;
  uint32_t const x = pixel - y * A.width;
  L.pixel = pixel;
  L.dyp = (-2.0f * (float)y / (float)A.height + 1.0f) * A.dh;
  L.dxp = (2.0f * (float)x / (float)A.width - 1.0f) * A.dw;
  L.sum = mk3(0.0f, 0.0f, 0.0f);
  uint32_t const g = A.slice_count;
  // first sample s with (s + pixel) % g == slice_index
  L.s_gen = (A.slice_index + g - pixel % g) % g;
  L.b_cnt = 0;
}

// Material::Scatter for the three materials, ray_tracing/default_materials.cpp:11-67.
RT_FN void scatter(uint32_t type, float roughness, float refraction, f3 rd, f3 n, f3 ball, float xi,
                   f3 & out_dir, float & out_e)
{
  float const dn = dot3(rd, n);
  f3 const refl = rd - n * dn * 2.0f;  // glm::reflect
  if (type == RT_MATTE)
  {
    out_dir = normalize3(n + ball);
    out_e = 0.7f;
  }
  else if (type == RT_METAL)
  {
    f3 const r = normalize3(refl + ball * roughness);
    out_dir = r;
    out_e = (dot3(n, r) <= 0.0f) ? 0.0f : 0.9f * (1.0f - roughness);
  }
  else
  {
    float vdn = dn;
    f3 outward;
    float eta;
    if (vdn > 0.0f)
    {
      outward = -n;
      eta = refraction;
      vdn *= eta;
    }
    else
    {
      outward = n;
      eta = 1.0f / refraction;
      vdn = -vdn;
    }
    // glm::refract(I, N, eta)
    float const d = dot3(outward, rd);
    float const k = 1.0f - eta * eta * (1.0f - d * d);
    f3 refr = mk3(0.0f, 0.0f, 0.0f);
    if (k >= 0.0f)
      refr = rd * eta - outward * (eta * d + hw_sqrt(k));
    float ref = 1.0f;
    if (dot3(refr, refr) > RT_EPS)
    {
      float const f0 = (1.0f - refraction) / (1.0f + refraction);
      float const f2 = f0 * f0;
      ref = f2 + (1.0f - f2) * pow5(1.0f - vdn);
    }
    out_dir = (xi < ref) ? refl : refr;
    out_e = 1.0f;
  }
}

// DirectionalLight::GetSpecular, ray_tracing/directional_light.cpp:37-60.
RT_FN f3 specular(f3 ldir, f3 lcol, f3 albedo, float m_roughness, float m_refraction, f3 rd, f3 n)
{
  float const vdn = clampf(-dot3(rd, n), 0.0f, 1.0f);
  float const ndl = clampf(-dot3(n, ldir), 0.0f, 1.0f);
  f3 const h = -normalize3(ldir + rd);
  float const ndh = clampf(dot3(n, h), RT_EPS, 1.0f);
  float const vdh = clampf(-dot3(rd, h), 0.0f, 1.0f);
  float const g = (hw_abs(vdh) >= RT_EPS) ? hw_min(1.0f, 2.0f * ndh * hw_min(vdn, ndl) / vdh) : 1.0f;
  float const ndh2 = ndh * ndh;
  float const roughness = hw_max(m_roughness, 0.03f);
  float const r2 = roughness * roughness;
  float const dm = hw_exp2(((ndh2 - 1.0f) / (r2 * ndh2)) * 1.44269504088896340736f) /
                   (4.0f * r2 * ndh2 * ndh2);
  float const f0 = (1.0f - m_refraction) / (1.0f + m_refraction);
  float const f2 = f0 * f0;
  float const f = f2 + (1.0f - f2) * pow5(1.0f - vdn);
  float const spec = clampf(g * dm * f, 0.0f, 1.0f);
  return mix3(lcol, albedo, 0.5f) * spec;
}

// Reverse accumulation, demos/demo_frame_cpu.cpp:60-75, over the lane's stack of (l.rgb, emissivity).
RT_FN f3 accumulate(const f4 * stack, uint32_t stride, uint32_t count)
{
  int i = (int)count - 1;
  f3 att = mk3(1.0f, 1.0f, 1.0f);
  float rest = 1.0f;
  while (i >= 0)
  {
    f4 const s = stack[(uint32_t)i * stride];
    att = att * (mk3(s.x, s.y, s.z) * rest);
    rest *= s.w;
    i--;
    if (hw_abs(rest) < RT_EPS)
      break;
  }
  return att * 0.5f;
}

RT_FN void push(f4 * stack, uint32_t stride, uint32_t depth, f3 l, float e)
{
  f4 s;
  s.x = l.x;
  s.y = l.y;
  s.z = l.z;
  s.w = e;
  stack[depth * stride] = s;
}

// The ray pass over one resident tile of spheres: Sphere::TraceNearest (ray_tracing/sphere.cpp:63-81)
// for RT_RAYS rays with a common origin, keeping the nearest accepted root per ray as
// TraceNearestInHitableCollection does (ray_tracing/hitable_object.hpp:71-87; strictly smaller t
// replaces, so the first sphere wins ties).  Works on ta = t * a (a = dot(D, D)), so the range test is
// ta >= tlo * a and ta < best where best starts at thi * a.  A negative discriminant gives a NaN root
// which fails both comparisons.  `base` is the scene index of sph[0].
RT_FN void ray_pass(const f4 * sph, uint32_t count, uint32_t base, f3 o, const f3 (&dir)[RT_RAYS],
                    const float (&a)[RT_RAYS], const float (&tlo_a)[RT_RAYS], float (&best)[RT_RAYS],
                    int (&idx)[RT_RAYS])
{
#pragma unroll 2
  for (uint32_t i = 0; i < count; ++i)
  {
    f4 const s = sph[i];
    float const dx = o.x - s.x, dy = o.y - s.y, dz = o.z - s.z;
    float const c = (dx * dx + dy * dy + dz * dz) - s.w;
    int const id = (int)(base + i);
#pragma unroll
    for (int r = 0; r < RT_RAYS; ++r)
    {
      float const hb = dx * dir[r].x + dy * dir[r].y + dz * dir[r].z;
      float const disc = hb * hb - a[r] * c;
      float const ta = -hb - hw_sqrt(disc);
      bool const take = (ta >= tlo_a[r]) && (ta < best[r]);
      best[r] = take ? ta : best[r];
      idx[r] = take ? id : idx[r];
    }
  }
}

// What the enclosing kernel provides.  any(): true if the predicate holds for any lane that shares this
// lane's loop trip (the wave, or the block when sphere tiles are re-staged inside the loop);
// fetch_pixel(): hands a fresh pixel index to each lane that asks (RT_INVALID when the frame is used up);
// run_pass(): executes ray_pass over every sphere of the scene.
template <class Env>
RT_FN void trace_loop(const TraceArgs & A, Env & env, f4 * stack, uint32_t stride, LaneCounters & cnt)
{
  Lane L;
  L.alive = true;
  L.need_pixel = true;
  L.has_hit = false;
  L.mode = MODE_NONE;
  L.b_cnt = 0;
  L.b_s = 0;
  L.s_gen = 0;
  L.pixel = 0;
  L.depth = 0;
  L.sample = 0;
  L.li = 0;
  L.sum = mk3(0.0f, 0.0f, 0.0f);
  L.dxp = L.dyp = 0.0f;
#pragma unroll
  for (int r = 0; r < RT_RAYS; ++r)
  {
    L.bt[r] = 0.0f;
    L.bi[r] = -1;
  }
  bool const deterministic = (A.flags & RT_FLAG_DETERMINISTIC) != 0u;
  uint32_t const stride_g = A.slice_count;

  for (;;)
  {
    // ---- 1. cheap transitions: consume primary results, finish pixels, fetch new pixels ----
    for (int it = 0; it < 4; ++it)
    {
      while (L.alive && !L.need_pixel && !L.has_hit && L.mode == MODE_NONE)
      {
        if (L.b_cnt > 0)
        {
          float const ta = L.bt[0];
          int const hi = L.bi[0];
          uint32_t const s = L.b_s;
#pragma unroll
          for (int r = 0; r + 1 < RT_RAYS; ++r)
          {
            L.bt[r] = L.bt[r + 1];
            L.bi[r] = L.bi[r + 1];
          }
          L.b_cnt -= 1;
          L.b_s += stride_g;
          cnt.samples += 1;
          cnt.queries += 1;
          if (hi < 0)
          {
            // No hit at all: the sample is the environment colour (demos/demo_frame_cpu.cpp:57-58).
            L.sum = L.sum + A.background;
          }
          else
          {
            cnt.hits += 1;
            f3 const d = primary_dir(A, L.dxp, L.dyp, s);
            float const t = ta / dot3(d, d);
            f4 const sp = env.sphere(hi);
            L.hp = A.cam_pos + d * t;
            L.hn = normalize3(L.hp - mk3(sp.x, sp.y, sp.z));
            L.hd = d;
            L.hidx = hi;
            L.has_hit = true;
            L.depth = 0;
            L.sample = s;
          }
        }
        else if (L.s_gen < A.s_count)
        {
          L.mode = MODE_PRIMARY;
        }
        else
        {
          // Pixel complete: one plain store per plane, each pixel written by exactly one lane.
          uint32_t const np = A.width * A.height;
          A.planes[L.pixel] = L.sum.x;
          A.planes[np + L.pixel] = L.sum.y;
          A.planes[2u * np + L.pixel] = L.sum.z;
          L.need_pixel = true;
        }
      }
      bool const need = L.alive && L.need_pixel;
      if (!env.any(need))
        break;
      uint32_t const p = env.fetch_pixel(need);
      if (need)
      {
        if (p == RT_INVALID)
        {
          L.alive = false;
        }
        else
        {
          begin_pixel(A, L, p);
          L.need_pixel = false;
        }
      }
    }
    if (!env.any(L.alive))
      break;

    // ---- 2. ray set-up ----
    f3 origin = mk3(0.0f, 0.0f, 0.0f);
    f3 dir[RT_RAYS];
    float a[RT_RAYS], tlo_a[RT_RAYS], best[RT_RAYS];
    int idx[RT_RAYS];
#pragma unroll
    for (int r = 0; r < RT_RAYS; ++r)
    {
      // A disabled ray: zero direction and a negative upper bound, so no root is ever accepted.
      dir[r] = mk3(0.0f, 0.0f, 0.0f);
      a[r] = 0.0f;
      tlo_a[r] = 0.0f;
      best[r] = -1.0f;
      idx[r] = -1;
    }

    if (L.alive && L.has_hit && L.mode == MODE_NONE)
    {
      // A new hit: Material::Scatter first (demos/demo_frame_cpu.cpp:38).
      uint32_t const mi = env.sphere_material(L.hidx);
      f4 const m0 = A.materials[2u * mi];
      f4 const m1 = A.materials[2u * mi + 1u];
      uint32_t mtype;
#if defined(__HIPCC__)
      mtype = __float_as_uint(m1.y);
#else
      memcpy(&mtype, &m1.y, 4);
#endif
      L.m_albedo = mk3(m0.x, m0.y, m0.z);
      float const m_rough = m0.w;
      L.m_refraction = m1.x;

      f3 ball = mk3(0.0f, 0.0f, 0.0f);
      float xi = 0.5f;
      if (!deterministic)
      {
        u4 const w = philox4x32_10(L.depth, 0u, 0u, 0u, L.pixel, L.sample ^ A.seed);
        ball = ball3(1.0f, uniform01(w.x), uniform01(w.y), uniform01(w.z));
        xi = uniform01(w.w);
      }
      scatter(mtype, m_rough, L.m_refraction, L.hd, L.hn, ball, xi, L.sdir, L.e);
      L.has_hit = false;
      if (hw_abs(L.e) < RT_EPS)
      {
        // Absorbed: the entry becomes the environment colour and the path ends
        // (demos/demo_frame_cpu.cpp:39-45).
        push(stack, stride, L.depth, A.background, 1.0f);
        L.depth += 1;
        L.sum = L.sum + accumulate(stack, stride, L.depth);
      }
      else
      {
        L.spec = mk3(0.0f, 0.0f, 0.0f);
        for (uint32_t li = 0; li < A.n_lights; ++li)
        {
          f4 const l0 = A.lights[2u * li];
          f4 const l1 = A.lights[2u * li + 1u];
          L.spec = L.spec + specular(mk3(l0.x, l0.y, l0.z), mk3(l1.x, l1.y, l1.z), L.m_albedo, m_rough,
                                     L.m_refraction, L.hd, L.hn);
        }
        L.diffuse = mk3(0.0f, 0.0f, 0.0f);
        L.li = 0;
        L.s_a = dot3(L.sdir, L.sdir);
        L.nx_i = -1;
        L.nx_ta = 0.0f;
        L.mode = MODE_SHADE;
      }
    }

    if (L.alive && L.mode == MODE_SHADE)
    {
      origin = L.hp;
      if (L.li < A.n_lights)
      {
        f4 const l0 = A.lights[2u * L.li];
        f3 const to_light = mk3(-l0.x, -l0.y, -l0.z);
#pragma unroll
        for (int j = 0; j < RT_LIGHT_SAMPLES; ++j)
        {
          f3 jit = mk3(0.0f, 0.0f, 0.0f);
          if (!deterministic)
          {
            u4 const w = philox4x32_10(L.depth, 1u + L.li * RT_LIGHT_SAMPLES + (uint32_t)j, 0u, 0u, L.pixel,
                                       L.sample ^ A.seed);
            jit = ball3(0.25f, uniform01(w.x), uniform01(w.y), uniform01(w.z));
          }
          f3 const rd = normalize3(to_light + jit);
          dir[j] = rd;
          a[j] = dot3(rd, rd);
          tlo_a[j] = 0.001f * a[j];
          best[j] = 1000.0f * a[j];
          L.w[j] = hw_max(0.0f, dot3(rd, L.hn));
        }
      }
      if (L.li == 0)
      {
        dir[RT_RAYS - 1] = L.sdir;
        a[RT_RAYS - 1] = L.s_a;
        tlo_a[RT_RAYS - 1] = 0.001f * L.s_a;  // secondary hits start at 0.001 (demos/demo_frame_cpu.cpp:53)
        best[RT_RAYS - 1] = A.zfar * L.s_a;
      }
      cnt.passes += 1;
    }
    else if (L.alive && L.mode == MODE_PRIMARY)
    {
      origin = A.cam_pos;
      L.b_s = L.s_gen;
      uint32_t n = 0;
#pragma unroll
      for (int r = 0; r < RT_RAYS; ++r)
      {
        if (L.s_gen < A.s_count)
        {
          f3 const d = primary_dir(A, L.dxp, L.dyp, L.s_gen);
          dir[r] = d;
          a[r] = dot3(d, d);
          tlo_a[r] = A.znear * a[r];
          best[r] = A.zfar * a[r];
          L.s_gen += stride_g;
          n += 1;
        }
      }
      L.b_cnt = n;
      cnt.passes += 1;
    }

    // ---- 3. the pass: every ray against every sphere ----
    env.run_pass(origin, dir, a, tlo_a, best, idx);

    // ---- 4. take the results ----
    if (L.alive && L.mode == MODE_PRIMARY)
    {
#pragma unroll
      for (int r = 0; r < RT_RAYS; ++r)
      {
        L.bt[r] = best[r];
        L.bi[r] = idx[r];
      }
      L.mode = MODE_NONE;
    }
    else if (L.alive && L.mode == MODE_SHADE)
    {
      if (L.li < A.n_lights)
      {
        // DirectionalLight::TraceLight, ray_tracing/directional_light.cpp:15-35.
        f4 const l1 = A.lights[2u * L.li + 1u];
        f3 const lcol = mk3(l1.x, l1.y, l1.z);
        // An occluded sample is tinted by the shaded surface's own material.
        f3 const tint = mix3(L.m_albedo, lcol, 0.75f) * (L.m_refraction * 0.75f);
        f3 color = mk3(0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int j = 0; j < RT_LIGHT_SAMPLES; ++j)
        {
          bool const occluded = idx[j] >= 0;
          color = color + (occluded ? tint : lcol) * L.w[j];
          cnt.hits += occluded ? 1u : 0u;
        }
        L.diffuse = L.diffuse + color * (1.0f / (float)RT_LIGHT_SAMPLES);
        cnt.queries += RT_LIGHT_SAMPLES;
      }
      if (L.li == 0)
      {
        L.nx_i = idx[RT_RAYS - 1];
        L.nx_ta = best[RT_RAYS - 1];
      }
      L.li += 1;
      if (L.li >= A.n_lights)
      {
        // l = attenuation + diffuse + specular (demos/demo_frame_cpu.cpp:66); every material's
        // attenuation is its albedo (ray_tracing/default_materials.cpp:14,26,40).
        push(stack, stride, L.depth, (L.m_albedo + L.diffuse) + L.spec, L.e);
        L.depth += 1;
        cnt.shaded += 1;
        L.mode = MODE_NONE;
        bool ended = true;
        if (L.depth < A.max_depth)
        {
          // The reference traces the scattered ray only while the depth limit allows another entry.
          cnt.queries += 1;
          if (L.nx_i >= 0)
          {
            cnt.hits += 1;
            float const t = L.nx_ta / L.s_a;
            f4 const sp = env.sphere(L.nx_i);
            f3 const p = L.hp + L.sdir * t;
            L.hn = normalize3(p - mk3(sp.x, sp.y, sp.z));
            L.hp = p;
            L.hd = L.sdir;
            L.hidx = L.nx_i;
            L.has_hit = true;
            ended = false;
          }
        }
        if (ended)
          L.sum = L.sum + accumulate(stack, stride, L.depth);
      }
    }
  }
}
}  // namespace rtc
