/*
 * rt_oracle.c -- CPU restatement of the rokuz/rt trace loop.  TEST INFRASTRUCTURE ONLY; see rt_oracle.h.
 * PARITY UNPINNED (the reference has no fixtures for this path and cannot be built here).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fPIC -shared -pthread rt_oracle.c -lm
 *
 * Citations are relative to the reference tree (ray_tracing/..., demos/...).
 */
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct
{
  float x, y, z;
} v3;

/* ---- glm equivalents (GLSL definitions; glm/geometric.hpp, glm/common.hpp) ---- */
static inline v3 mk(float x, float y, float z)
{
  v3 r = {x, y, z};
  return r;
}
static inline v3 ld(const float * p) { return mk(p[0], p[1], p[2]); }
static inline void st(float * p, v3 v)
{
  p[0] = v.x;
  p[1] = v.y;
  p[2] = v.z;
}
static inline v3 add(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mulv(v3 a, v3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 muls(v3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
static inline v3 divs(v3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
static inline v3 neg(v3 a) { return mk(-a.x, -a.y, -a.z); }
/* glm::dot(vec3): tmp = a*b; tmp.x + tmp.y + tmp.z */
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 cross(v3 x, v3 y)
{
  return mk(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
/* glm::normalize(v) = v * inversesqrt(dot(v, v)) */
static inline v3 normalize(v3 v) { return muls(v, 1.0f / sqrtf(dot(v, v))); }
/* glm::reflect(I, N) = I - N * dot(N, I) * 2 */
static inline v3 reflect(v3 i, v3 n) { return sub(i, muls(muls(n, dot(n, i)), 2.0f)); }
/* glm::refract(I, N, eta) */
static inline v3 refract(v3 i, v3 n, float eta)
{
  float const d = dot(n, i);
  float const k = 1.0f - eta * eta * (1.0f - d * d);
  if (k >= 0.0f)
    return sub(muls(i, eta), muls(n, eta * d + sqrtf(k)));
  return mk(0.0f, 0.0f, 0.0f);
}
/* glm::mix(x, y, a) = x * (1 - a) + y * a */
static inline v3 mix(v3 x, v3 y, float a) { return add(muls(x, 1.0f - a), muls(y, a)); }
static inline float minf(float a, float b) { return (b < a) ? b : a; } /* std::min */
static inline float maxf(float a, float b) { return (a < b) ? b : a; } /* std::max */
static inline float clampf(float x, float lo, float hi) { return minf(maxf(x, lo), hi); }

/* ---- random source ---- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r)
  {
    uint64_t const p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t const p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t const n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t const n1 = (uint32_t)p1;
    uint32_t const n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t const n3 = (uint32_t)p0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

float orc_uniform(uint32_t bits) { return (float)(bits >> 8) * (1.0f / 16777216.0f); }

/* Stands in for glm::ballRand(R) (call sites default_materials.cpp:15,24; directional_light.cpp:20):
 * the same uniform-in-ball law, drawn with a fixed number of uniforms: direction uniform on the sphere
 * (z uniform in [-1,1], azimuth uniform), radius R * u^(1/3). */
void orc_ball(float radius, float u1, float u2, float u3, float out[3])
{
  float const z = 1.0f - 2.0f * u1;
  float const s = sqrtf(maxf(0.0f, 1.0f - z * z));
  float const phi = 6.28318530717958647692f * u2;
  float const rad = radius * cbrtf(u3);
  out[0] = (s * cosf(phi)) * rad;
  out[1] = (s * sinf(phi)) * rad;
  out[2] = z * rad;
}

static void philox_block(const orc_params * params, uint32_t pixel, uint32_t sample, uint32_t bounce,
                         uint32_t block, uint32_t w[4])
{
  uint32_t const ctr[4] = {bounce, block, sample, 0u};
  uint32_t const key[2] = {pixel, params->seed};
  orc_philox4x32_10(ctr, key, w);
}

/* Block 0 of a bounce: three 24-bit uniforms for the scatter ball and one for the glass branch. */
static void draw_scatter(const orc_params * params, uint32_t pixel, uint32_t sample, uint32_t bounce, float u[4])
{
  uint32_t w[4];
  philox_block(params, pixel, sample, bounce, 0u, w);
  for (int i = 0; i < 4; ++i)
    u[i] = orc_uniform(w[i]);
}

/* Block 1 + light of a bounce: its sixteen bytes b[0..15] (word 0 low byte first); shadow sample j takes
 * b[3j], b[3j+1], b[3j+2] as uniforms (b + 0.5) / 256 (the middle of each of the 256 cells). */
static void draw_shadow_jitter(const orc_params * params, uint32_t pixel, uint32_t sample, uint32_t bounce,
                               uint32_t light, v3 jitter[ORC_LIGHT_SAMPLES])
{
  uint32_t w[4];
  philox_block(params, pixel, sample, bounce, 1u + light, w);
  for (uint32_t j = 0; j < ORC_LIGHT_SAMPLES; ++j)
  {
    float u[3], b[3];
    for (uint32_t q = 0; q < 3; ++q)
    {
      uint32_t const h = 3u * j + q;
      uint32_t const byte = (w[h >> 2] >> (8u * (h & 3u))) & 0xFFu;
      u[q] = (float)byte * (1.0f / 256.0f) + (0.5f / 256.0f);
    }
    orc_ball(0.25f, u[0], u[1], u[2], b);
    jitter[j] = ld(b);
  }
}

/* ---- camera: Frame::ForEachRayImpl, frame.cpp:31-62 ---- */
typedef struct
{
  v3 pos, dir, right, up;
  float znear, dw, dh, cell_x, cell_y;
} cam_basis;

static cam_basis make_basis(const orc_camera * cam, uint32_t width, uint32_t height)
{
  cam_basis b;
  b.pos = ld(cam->position);
  b.dir = ld(cam->direction);
  b.right = cross(mk(0.0f, 1.0f, 0.0f), b.dir); /* frame.cpp:33 */
  b.up = cross(b.dir, b.right);                 /* frame.cpp:34 */
  b.znear = cam->znear;
  b.dw = cam->znear / tanf(0.5f * cam->fov);          /* frame.cpp:36 */
  float const aspect = (float)height / (float)width; /* frame.cpp:37 */
  b.dh = b.dw * aspect;                               /* frame.cpp:38 */
  b.cell_x = 2.0f * b.dw / (float)width;              /* frame.cpp:40 */
  b.cell_y = 2.0f * b.dh / (float)height;             /* frame.cpp:41 */
  return b;
}

static v3 primary_dir(const cam_basis * b, uint32_t width, uint32_t height, uint32_t s_row, uint32_t x,
                      uint32_t y, uint32_t sx, uint32_t sy)
{
  float const dy = (-2.0f * (float)y / (float)height + 1.0f) * b->dh; /* frame.cpp:49 */
  float const dx = (2.0f * (float)x / (float)width - 1.0f) * b->dw;   /* frame.cpp:52 */
  float const sdx = dx + b->cell_x * (float)sx / (float)s_row;        /* frame.cpp:58 */
  float const sdy = dy - b->cell_y * (float)sy / (float)s_row;        /* frame.cpp:61 */
  /* frame.cpp:62 */
  return normalize(add(add(muls(b->dir, b->znear), muls(b->up, sdy)), muls(b->right, sdx)));
}

void orc_primary_direction(const orc_camera * cam, uint32_t width, uint32_t height, uint32_t samples_in_row,
                           uint32_t x, uint32_t y, uint32_t sx, uint32_t sy, float out_dir[3])
{
  cam_basis const b = make_basis(cam, width, height);
  st(out_dir, primary_dir(&b, width, height, samples_in_row, x, y, sx, sy));
}

/* ---- Sphere::TraceNearest, sphere.cpp:63-81 ---- */
static int sphere_nearest(v3 o, v3 dir, v3 center, float radius, float tmin, float tmax, float * t_out)
{
  v3 const d = sub(o, center);
  float const a = dot(dir, dir);
  float const b = 2.0f * dot(d, dir);
  float const c = dot(d, d) - radius * radius;
  float const discriminant = b * b - 4 * a * c;
  if (discriminant < 0.0f)
    return 0;
  float const sqrt_d = sqrtf(discriminant);
  float const t = minf((-b - sqrt_d) / (2.0f * a), (-b + sqrt_d) / (2.0f * a));
  if (t < tmin || t > tmax)
    return 0;
  *t_out = t;
  return 1;
}

int orc_sphere_trace_nearest(const float origin[3], const float dir[3], const float center[3], float radius,
                             float tmin, float tmax, float * out_t, float out_pos[3], float out_normal[3])
{
  float t;
  v3 const o = ld(origin), d = ld(dir), c = ld(center);
  if (!sphere_nearest(o, d, c, radius, tmin, tmax, &t))
    return 0;
  v3 const p = add(o, muls(d, t)); /* Ray::PointAt, ray.hpp:19 */
  v3 const n = normalize(sub(p, c));
  *out_t = t;
  st(out_pos, p);
  st(out_normal, n);
  return 1;
}

/* Sphere::Trace, ray_tracing/sphere.cpp:11-61: every intersection in [tmin, tmax], nearest first (at most two;
 * one when the discriminant is within kEps of zero).  Unused by either of the reference's backends; kept for
 * the completeness of the HitableObject interface. */
int orc_sphere_trace(const float origin[3], const float dir[3], const float center[3], float radius, float tmin,
                     float tmax, orc_hit out[2])
{
  v3 const o = ld(origin), dv = ld(dir), cv = ld(center);
  v3 const d = sub(o, cv);
  float const a = dot(dv, dv);
  float const b = 2.0f * dot(d, dv);
  float const c = dot(d, d) - radius * radius;
  float const disc = b * b - 4 * a * c;
  float ts[2];
  int n = 0;
  if (disc < 0.0f)
    return 0;
  if (fabsf(disc) < ORC_EPS)
  {
    float const t = -b / (2.0f * a);
    if (t < tmin || t > tmax)
      return 0;
    ts[n++] = t;
  }
  else
  {
    float const sq = sqrtf(disc);
    float const t1 = (-b - sq) / (2.0f * a);
    float const t2 = (-b + sq) / (2.0f * a);
    int const has1 = (t1 >= tmin && t1 <= tmax);
    int const has2 = (t2 >= tmin && t2 <= tmax);
    if (has1 && has2)
    {
      ts[n++] = minf(t1, t2);
      ts[n++] = maxf(t1, t2);
    }
    else if (has1)
      ts[n++] = t1;
    else if (has2)
      ts[n++] = t2;
  }
  for (int i = 0; i < n; ++i)
  {
    v3 const p = add(o, muls(dv, ts[i]));
    out[i].t = ts[i];
    st(out[i].position, p);
    st(out[i].normal, normalize(sub(p, cv)));
    out[i].sphere = 0u;
  }
  return n;
}

/* TraceInHitableCollection, ray_tracing/hitable_object.hpp:53-69: the hits of all spheres, sorted by t
 * (std::sort there; a stable insertion sort here, so equal t keep collection order).  Returns the number of
 * hits, of which the first min(count, capacity) are stored. */
uint32_t orc_trace_all(const orc_scene * scene, const float origin[3], const float dir[3], float tmin, float tmax,
                       orc_hit * out, uint32_t capacity)
{
  uint32_t total = 0, stored = 0;
  for (uint32_t i = 0; i < scene->sphere_count; ++i)
  {
    const float * s = scene->spheres + 4u * i;
    orc_hit h[2];
    int const n = orc_sphere_trace(origin, dir, s, s[3], tmin, tmax, h);
    for (int k = 0; k < n; ++k)
    {
      total++;
      h[k].sphere = i;
      /* insert keeping the stored prefix sorted; drop the largest when full */
      uint32_t pos = stored;
      while (pos > 0 && out[pos - 1].t > h[k].t)
        pos--;
      if (pos >= capacity)
        continue;
      uint32_t const last = (stored < capacity) ? stored : capacity - 1u;
      for (uint32_t j = last; j > pos; --j)
        out[j] = out[j - 1];
      out[pos] = h[k];
      if (stored < capacity)
        stored++;
    }
  }
  return total;
}

typedef struct
{
  float t;
  v3 pos, normal;
  int index;
} hit_rec;

/* TraceNearestInHitableCollection, hitable_object.hpp:71-87: linear scan, strictly smaller t replaces. */
static int trace_nearest(const orc_scene * scene, v3 o, v3 dir, float tmin, float tmax, hit_rec * out,
                         orc_stats * stats)
{
  int best = -1;
  float best_t = 0.0f;
  for (uint32_t i = 0; i < scene->sphere_count; ++i)
  {
    const float * s = scene->spheres + 4u * i;
    float t;
    if (!sphere_nearest(o, dir, ld(s), s[3], tmin, tmax, &t))
      continue;
    if (best < 0 || t < best_t)
    {
      best = (int)i;
      best_t = t;
    }
  }
  if (stats)
  {
    stats->ray_queries += 1;
    stats->sphere_tests += scene->sphere_count;
    if (best >= 0)
      stats->hit_queries += 1;
  }
  if (best < 0)
    return 0;
  const float * s = scene->spheres + 4u * (uint32_t)best;
  out->t = best_t;
  out->pos = add(o, muls(dir, best_t));
  out->normal = normalize(sub(out->pos, ld(s)));
  out->index = best;
  return 1;
}

int orc_trace_nearest(const orc_scene * scene, const float origin[3], const float dir[3], float tmin, float tmax,
                      float * out_t, float out_pos[3], float out_normal[3])
{
  hit_rec h;
  if (!trace_nearest(scene, ld(origin), ld(dir), tmin, tmax, &h, NULL))
    return -1;
  *out_t = h.t;
  st(out_pos, h.pos);
  st(out_normal, h.normal);
  return h.index;
}

/* ---- Material::Scatter, default_materials.cpp:11-67 ---- */
typedef struct
{
  v3 attenuation;
  v3 dir;
  float emissivity;
} scatter_res;

static scatter_res scatter(const orc_material * mat, v3 ray_dir, v3 n, v3 ball, float xi)
{
  scatter_res r;
  r.attenuation = ld(mat->albedo);
  if (mat->type == ORC_MATERIAL_MATTE)
  {
    /* default_materials.cpp:11-19 */
    r.dir = normalize(add(n, ball));
    r.emissivity = 0.7f;
  }
  else if (mat->type == ORC_MATERIAL_METAL)
  {
    /* default_materials.cpp:21-33 */
    v3 refl = reflect(ray_dir, n);
    refl = normalize(add(refl, muls(ball, mat->roughness)));
    r.dir = refl;
    if (dot(n, refl) <= 0.0f)
      r.emissivity = 0.0f;
    else
      r.emissivity = 0.9f * (1.0f - mat->roughness);
  }
  else
  {
    /* default_materials.cpp:35-67 */
    v3 const refl = reflect(ray_dir, n);
    float const refraction = mat->refraction;
    v3 outward;
    float eta;
    float vdn = dot(ray_dir, n);
    if (vdn > 0.0f)
    {
      outward = neg(n);
      eta = refraction;
      vdn *= eta;
    }
    else
    {
      outward = n;
      eta = 1.0f / refraction;
      vdn = -vdn;
    }
    v3 const refr = refract(ray_dir, outward, eta);
    float ref = 1.0f;
    if (dot(refr, refr) > ORC_EPS)
    {
      float const f0 = (1.0f - refraction) / (1.0f + refraction);
      float const f2 = f0 * f0;
      ref = f2 + (1.0f - f2) * powf(1.0f - vdn, 5.0f);
    }
    r.dir = (xi < ref) ? refl : refr;
    r.emissivity = 1.0f;
  }
  return r;
}

void orc_scatter(const orc_material * mat, const float ray_dir[3], const float hit_normal[3], const float ball[3],
                 float xi, float out_attenuation[3], float out_dir[3], float * out_emissivity)
{
  scatter_res const r = scatter(mat, ld(ray_dir), ld(hit_normal), ld(ball), xi);
  st(out_attenuation, r.attenuation);
  st(out_dir, r.dir);
  *out_emissivity = r.emissivity;
}

/* ---- DirectionalLight::TraceLight, directional_light.cpp:15-35 ---- */
static v3 trace_light(const orc_scene * scene, const orc_light * light, const orc_material * mat, v3 pos, v3 n,
                      const v3 jitter[ORC_LIGHT_SAMPLES], orc_stats * stats)
{
  v3 const ldir = ld(light->direction);
  v3 const lcol = ld(light->color);
  v3 color = mk(0.0f, 0.0f, 0.0f);
  for (uint32_t i = 0; i < ORC_LIGHT_SAMPLES; ++i)
  {
    v3 const rd = normalize(add(neg(ldir), jitter[i]));
    hit_rec h;
    if (!trace_nearest(scene, pos, rd, 0.001f, 1000.0f, &h, stats))
    {
      color = add(color, muls(lcol, maxf(0.0f, dot(rd, n))));
    }
    else
    {
      /* The shaded surface's own material tints its shadow (directional_light.cpp:29-31). */
      v3 const c = mix(ld(mat->albedo), lcol, 0.75f);
      color = add(color, muls(muls(c, mat->refraction * 0.75f), maxf(0.0f, dot(rd, n))));
    }
  }
  return divs(color, (float)ORC_LIGHT_SAMPLES);
}

void orc_trace_light(const orc_scene * scene, const orc_light * light, const orc_material * mat,
                     const float hit_pos[3], const float hit_normal[3], const float jitter[ORC_LIGHT_SAMPLES][3],
                     float out_color[3], orc_stats * stats)
{
  v3 j[ORC_LIGHT_SAMPLES];
  for (uint32_t i = 0; i < ORC_LIGHT_SAMPLES; ++i)
    j[i] = ld(jitter[i]);
  st(out_color, trace_light(scene, light, mat, ld(hit_pos), ld(hit_normal), j, stats));
}

/* ---- DirectionalLight::GetSpecular, directional_light.cpp:37-60 (Cook-Torrance) ---- */
static v3 get_specular(const orc_light * light, const orc_material * mat, v3 rd, v3 n)
{
  v3 const ldir = ld(light->direction);
  float const vdn = clampf(dot(neg(rd), n), 0.0f, 1.0f);
  float const ndl = clampf(dot(n, neg(ldir)), 0.0f, 1.0f);
  v3 const h = neg(normalize(add(ldir, rd)));
  float const ndh = clampf(dot(n, h), ORC_EPS, 1.0f);
  float const vdh = clampf(dot(neg(rd), h), 0.0f, 1.0f);

  float const g = (fabsf(vdh) >= ORC_EPS) ? minf(1.0f, 2.0f * ndh * minf(vdn, ndl) / vdh) : 1.0f;
  float const ndh2 = ndh * ndh;

  float const roughness = maxf(mat->roughness, 0.03f);
  float const r2 = roughness * roughness;
  float const d = expf((ndh2 - 1.0f) / (r2 * ndh2)) / (4.0f * r2 * ndh2 * ndh2);

  float const ref = mat->refraction;
  float const f0 = (1.0f - ref) / (1.0f + ref);
  float const f2 = f0 * f0;
  float const f = f2 + (1.0f - f2) * powf(1.0f - vdn, 5.0f);

  float const spec = clampf(g * d * f, 0.0f, 1.0f);
  return muls(mix(ld(light->color), ld(mat->albedo), 0.5f), spec);
}

void orc_get_specular(const orc_light * light, const orc_material * mat, const float ray_dir[3],
                      const float hit_normal[3], float out_color[3])
{
  st(out_color, get_specular(light, mat, ld(ray_dir), ld(hit_normal)));
}

/* ---- reverse accumulation, demo_frame_cpu.cpp:60-75 ---- */
static v3 accumulate(uint32_t count, const v3 * l, const float * e)
{
  int i = (int)count - 1;
  v3 att = mk(1.0f, 1.0f, 1.0f);
  float rest = 1.0f;
  while (i >= 0)
  {
    att = mulv(att, muls(l[i], rest));
    rest *= e[i];
    i--;
    if (fabsf(rest) < ORC_EPS)
      break;
  }
  return muls(att, 0.5f);
}

void orc_accumulate(uint32_t count, const float (*l)[3], const float * emissivity, float out_color[3])
{
  v3 lv[64];
  if (count > 64u)
    count = 64u;
  for (uint32_t i = 0; i < count; ++i)
    lv[i] = ld(l[i]);
  st(out_color, accumulate(count, lv, emissivity));
}

/* ---- DemoFrameCPU::RayTrace, demo_frame_cpu.cpp:22-76 ---- */
static v3 ray_trace(const orc_scene * scene, const orc_camera * cam, const orc_params * params, uint32_t pixel,
                    uint32_t sample, v3 dir, orc_stats * stats)
{
  uint32_t const max_depth =
      (params->max_depth == 0u || params->max_depth > ORC_MAX_DEPTH) ? ORC_MAX_DEPTH : params->max_depth;
  int const deterministic = (params->flags & ORC_FLAG_DETERMINISTIC) != 0u;
  v3 const background = ld(cam->background);

  v3 l[ORC_MAX_DEPTH];
  float e[ORC_MAX_DEPTH];
  uint32_t count = 0;

  v3 origin = ld(cam->position);
  float znear = cam->znear;
  float const zfar = cam->zfar;
  for (; count < max_depth; ++count)
  {
    hit_rec h;
    if (!trace_nearest(scene, origin, dir, znear, zfar, &h, stats))
      break;

    const orc_material * mat = scene->materials + scene->material_index[h.index];

    float u[4] = {0.0f, 0.0f, 0.0f, 0.5f};
    v3 ball = mk(0.0f, 0.0f, 0.0f);
    if (!deterministic)
    {
      float b[3];
      draw_scatter(params, pixel, sample, count, u);
      orc_ball(1.0f, u[0], u[1], u[2], b);
      ball = ld(b);
    }
    scatter_res const sr = scatter(mat, dir, h.normal, ball, u[3]);
    if (fabsf(sr.emissivity) < ORC_EPS)
    {
      /* Absorbed: the entry becomes the environment colour and ends the path (demo_frame_cpu.cpp:39-45). */
      l[count] = background;
      e[count] = 1.0f;
      ++count;
      break;
    }

    v3 diffuse = mk(0.0f, 0.0f, 0.0f);
    v3 specular = mk(0.0f, 0.0f, 0.0f);
    for (uint32_t li = 0; li < scene->light_count; ++li)
    {
      v3 jitter[ORC_LIGHT_SAMPLES];
      for (uint32_t j = 0; j < ORC_LIGHT_SAMPLES; ++j)
        jitter[j] = mk(0.0f, 0.0f, 0.0f);
      if (!deterministic)
        draw_shadow_jitter(params, pixel, sample, count, li, jitter);
      diffuse = add(diffuse, trace_light(scene, scene->lights + li, mat, h.pos, h.normal, jitter, stats));
      specular = add(specular, get_specular(scene->lights + li, mat, dir, h.normal));
    }
    if (stats)
      stats->shaded_bounces += 1;

    l[count] = add(add(sr.attenuation, diffuse), specular);
    e[count] = sr.emissivity;

    origin = h.pos;
    dir = sr.dir;
    znear = 0.001f; /* demo_frame_cpu.cpp:53 */
  }

  if (count == 0)
    return background;
  return accumulate(count, l, e);
}

void orc_ray_trace(const orc_scene * scene, const orc_camera * cam, const orc_params * params, uint32_t pixel,
                   uint32_t sample, const float dir[3], float out_color[3], orc_stats * stats)
{
  st(out_color, ray_trace(scene, cam, params, pixel, sample, ld(dir), stats));
}

/* ---- Frame::ForEachRayImpl on MultiThreadedFrame's workers ---- */
typedef struct
{
  const orc_scene * scene;
  const orc_camera * cam;
  const orc_params * params;
  float * rgb;
  uint32_t start_row, pitch;
  uint32_t row_end; /* one past the last row of the band being rendered */
  orc_stats stats;
} worker_ctx;

static void * worker(void * arg)
{
  worker_ctx * w = (worker_ctx *)arg;
  const orc_params * p = w->params;
  uint32_t const width = p->width, height = p->height, s_row = p->samples_in_row;
  uint32_t const slices = p->slice_count ? p->slice_count : 1u;
  cam_basis const basis = make_basis(w->cam, width, height);
  float const inv_sample_count = 1.0f / (float)(s_row * s_row); /* frame.cpp:43 */
  memset(&w->stats, 0, sizeof(w->stats));

  /* Thread i takes rows i, i + T, ... (multithreaded_frame.cpp:49 -> frame.cpp:44), of the band. */
  (void)height;
  for (uint32_t y = w->start_row; y < w->row_end; y += w->pitch)
  {
    for (uint32_t x = 0; x < width; ++x)
    {
      uint32_t const pixel = y * width + x;
      v3 color = mk(0.0f, 0.0f, 0.0f);
      for (uint32_t sx = 0; sx < s_row; ++sx)
      {
        for (uint32_t sy = 0; sy < s_row; ++sy)
        {
          uint32_t const sample = sx * s_row + sy;
          /* Sharding (not in the reference): the samples of the frame, numbered (pixel, sx, sy) in this loop
           * order, are dealt out to the slices in turn. */
          if ((((uint64_t)pixel * s_row + sx) * s_row + sy) % slices != p->slice_index % slices)
            continue;
          v3 const d = primary_dir(&basis, width, height, s_row, x, y, sx, sy);
          color = add(color, ray_trace(w->scene, w->cam, p, pixel, sample, d, &w->stats));
          w->stats.samples += 1;
        }
      }
      st(w->rgb + 3u * (size_t)pixel, muls(color, inv_sample_count)); /* frame.cpp:67 */
    }
  }
  return NULL;
}

int orc_render_rows(const orc_scene * scene, const orc_camera * cam, const orc_params * params, uint32_t threads,
                    uint32_t row_begin, uint32_t row_end, float * rgb, orc_stats * stats)
{
  if (!scene || !cam || !params || !rgb || params->width == 0 || params->height == 0 ||
      params->samples_in_row == 0 || row_begin >= row_end || row_end > params->height)
    return 1;
  if (threads == 0)
    threads = 1;
  if (threads > row_end - row_begin)
    threads = row_end - row_begin;

  worker_ctx * ctx = (worker_ctx *)calloc(threads, sizeof(worker_ctx));
  pthread_t * tid = (pthread_t *)calloc(threads, sizeof(pthread_t));
  if (!ctx || !tid)
  {
    free(ctx);
    free(tid);
    return 2;
  }
  for (uint32_t i = 0; i < threads; ++i)
  {
    ctx[i].scene = scene;
    ctx[i].cam = cam;
    ctx[i].params = params;
    ctx[i].rgb = rgb;
    ctx[i].start_row = row_begin + i;
    ctx[i].row_end = row_end;
    ctx[i].pitch = threads;
  }
  int rc = 0;
  uint32_t started = 0;
  for (; started < threads; ++started)
  {
    if (pthread_create(&tid[started], NULL, worker, &ctx[started]) != 0)
    {
      rc = 3;
      break;
    }
  }
  /* True completion: every worker joined (the reference's finishing counter can fire early). */
  for (uint32_t i = 0; i < started; ++i)
    pthread_join(tid[i], NULL);

  if (stats)
  {
    memset(stats, 0, sizeof(*stats));
    for (uint32_t i = 0; i < started; ++i)
    {
      stats->samples += ctx[i].stats.samples;
      stats->ray_queries += ctx[i].stats.ray_queries;
      stats->hit_queries += ctx[i].stats.hit_queries;
      stats->shaded_bounces += ctx[i].stats.shaded_bounces;
      stats->sphere_tests += ctx[i].stats.sphere_tests;
    }
  }
  free(ctx);
  free(tid);
  return rc;
}

int orc_render(const orc_scene * scene, const orc_camera * cam, const orc_params * params, uint32_t threads,
               float * rgb, orc_stats * stats)
{
  if (!params)
    return 1;
  return orc_render_rows(scene, cam, params, threads, 0u, params->height, rgb, stats);
}
