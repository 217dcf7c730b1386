// rt_core.h -- per-lane path state machine of the gfx950 trace kernel.
//
// One lane carries one camera sample at a time and never waits for its neighbours' paths: a lane whose
// path ends starts its next sample on the next trip round the loop, and a lane whose pixel is finished
// pulls a new pixel from the frame-wide queue.  A trip has two stages.
//
//   Stage P  lanes without a hit to shade trace primary rays (ray_tracing/frame.cpp:56-66) until they
//            have one; the stage repeats while enough lanes are still looking.
//   Stage S  lanes with a hit run Material::Scatter and the specular term, then one "ray pass": the 5
//            jittered shadow rays of DirectionalLight::TraceLight (ray_tracing/directional_light.cpp:15-35)
//            and the scattered ray whose nearest hit starts the next bounce (demos/demo_frame_cpu.cpp:33,50).
//            All six leave the same point.
//
// Ray against all spheres, in two steps:
//   filter   a ray from O along unit D can only hit sphere (C, r) in front of O if
//            dot(C - O, D) >= sqrt(|O - C|^2 - r^2).  For primary rays the right-hand side is a per-frame table
//            (one origin: the camera) and a ray-sphere test is 3 FMAs and one instruction that shifts the sign
//            of the result into a per-lane candidate mask.  A shade pass takes the test in squared form, from
//            products of scene-centred coordinates of sphere and origin, without forming C - O (filter_sphere_2).
//            A small margin keeps the filter conservative.  The five shadow rays of a light all lie in a narrow
//            cone around the direction to the light, so a pass tests that cone once per sphere (cheaper than
//            one ray) and only the spheres the cone may touch go on to the exact intersection, for the five
//            rays at once.
//   resolve  the few candidates are intersected exactly as Sphere::TraceNearest does
//            (ray_tracing/sphere.cpp:63-81), operation for operation, with correctly rounded sqrt and division
//            (near_root) and without FMA contraction, keeping the nearest one as TraceNearestInHitableCollection does
//            (ray_tracing/hitable_object.hpp:71-87).  Shadow rays stop at their first accepted hit.
// Everything downstream of a hit (hit point, normal, scatter, specular, accumulation) uses the same
// un-contracted IEEE operations as the CPU path, so the only approximate arithmetic is in the filter,
// in the shadow-ray directions and in the random ball sampler.
//
// This translation unit is compiled with -ffp-contract=off; the filter asks for FMAs explicitly.
// The file is plain C++ over a handful of macros so the same state machine also compiles for the host,
// one lane at a time, which is how the logic is debugged without a GPU (tests/emu).
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_FN __device__ __forceinline__
#define RT_HD __host__ __device__ __forceinline__
#define RT_MEMBER __device__ __forceinline__
#else
#include <cmath>
#include <cstring>
#define RT_FN static inline
#define RT_HD static inline
#define RT_MEMBER inline
#endif

namespace rtc
{
constexpr int RT_RAYS = 6;             // rays per shade pass: 5 shadow + 1 scattered
constexpr int RT_LIGHT_SAMPLES = 5;    // ray_tracing/directional_light.cpp:13
constexpr uint32_t RT_MAX_DEPTH = 10;  // demos/demo_frame_cpu.cpp:10
constexpr float RT_EPS = 1e-5f;        // ray_tracing/types.hpp:9
constexpr uint32_t RT_INVALID = 0xFFFFFFFFu;
constexpr uint32_t RT_CHUNK = 32;      // spheres per candidate mask
constexpr uint32_t RT_MAT_RECORDS = 3; // float4 records per material
constexpr uint32_t RT_LIGHT_RECORDS = 3; // float4 records per light
constexpr uint32_t RT_UNROLL = 2;      // camera-table entries per trip of the column filter's main part
constexpr int RT_P_THRESHOLD = 8;      // stage P repeats while more lanes than this still need a hit
constexpr double RT_FILTER_MARGIN = 2e-6;  // relative slack of the filter's c = |O-C|^2 - r^2
// The shade filter takes |O-C|^2 and dot(C-O, D) apart into products of the two points (see filter_sphere_2), whose
// rounding errors grow with the points' distances from the scene's centre, not with |O-C|: it is made up for by
// RT_FILTER_SLACK times the sum of the two squared distances (the errors add up to less than 2.2e-6 times it).
constexpr double RT_FILTER_SLACK = 4e-6;

constexpr uint32_t RT_MATTE = 0, RT_METAL = 1, RT_GLASS = 2;  // ray_tracing/types.hpp:15-17
constexpr uint32_t RT_FLAG_DETERMINISTIC = 1u;
constexpr uint32_t RT_FLAG_GRID = 2u;  // walk the scene's uniform grid instead of testing every sphere

struct f3
{
  float x, y, z;
};
struct f4
{
  float x, y, z, w;
};
struct u4
{
  uint32_t x, y, z, w;
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
RT_HD f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
RT_HD float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// Three consecutive floats in one 12-byte store (4-byte aligned).
RT_HD void store3(float * p, f3 v)
{
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float f3v __attribute__((ext_vector_type(3), aligned(4)));
  f3v t;
  t.x = v.x;
  t.y = v.y;
  t.z = v.z;
  *reinterpret_cast<f3v *>(p) = t;
#else
  p[0] = v.x;
  p[1] = v.y;
  p[2] = v.z;
#endif
}

// ---- approximate hardware math (one instruction each on gfx950, about 1 ulp); libm on the host ----
#if defined(__HIPCC__)
RT_FN float hw_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
RT_FN float hw_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
RT_FN float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
RT_FN float hw_sin_rev(float r) { return __builtin_amdgcn_sinf(r); }  // sin(2*pi*r)
RT_FN float hw_cos_rev(float r) { return __builtin_amdgcn_cosf(r); }
RT_FN float hw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
RT_FN float hw_log2(float x) { return __builtin_amdgcn_logf(x); }
RT_FN float hw_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
RT_FN uint32_t hw_mulhi(uint32_t a, uint32_t b) { return __umulhi(a, b); }
RT_FN uint32_t hw_xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
RT_FN int hw_clz(uint32_t x) { return __clz((int)x); }
RT_FN int hw_clz64(unsigned long long x) { return __clzll((long long)x); }
RT_FN uint32_t f2u(float f) { return __float_as_uint(f); }
RT_FN float hw_max(float a, float b) { return fmaxf(a, b); }
RT_FN float hw_abs(float a) { return __builtin_fabsf(a); }
RT_FN float hw_min(float a, float b) { return fminf(a, b); }
// clamp as one instruction (the median of three); equal to min(max(x, lo), hi) for every x that is not a NaN
RT_FN float hw_clamp(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
// mask = 2 * mask + (the sign bit of x): one v_alignbit_b32.
RT_FN uint32_t push_sign(uint32_t mask, float x) { return __builtin_amdgcn_alignbit(mask, __float_as_uint(x), 31u); }
#else
RT_FN float hw_sqrt(float x) { return sqrtf(x); }
RT_FN float hw_rsq(float x) { return 1.0f / sqrtf(x); }
RT_FN float hw_rcp(float x) { return 1.0f / x; }
RT_FN float hw_sin_rev(float r) { return sinf(6.28318530717958647692f * r); }
RT_FN float hw_cos_rev(float r) { return cosf(6.28318530717958647692f * r); }
RT_FN float hw_exp2(float x) { return exp2f(x); }
RT_FN float hw_log2(float x) { return log2f(x); }
RT_FN float hw_fma(float a, float b, float c) { return a * b + c; }
RT_FN uint32_t hw_mulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
RT_FN uint32_t hw_xor3(uint32_t a, uint32_t b, uint32_t c) { return a ^ b ^ c; }
RT_FN int hw_clz(uint32_t x) { return __builtin_clz(x); }
RT_FN int hw_clz64(unsigned long long x) { return __builtin_clzll(x); }
RT_FN uint32_t f2u(float f)
{
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
RT_FN float hw_max(float a, float b) { return fmaxf(a, b); }
RT_FN float hw_abs(float a) { return fabsf(a); }
RT_FN float hw_min(float a, float b) { return fminf(a, b); }
RT_FN float hw_clamp(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
RT_FN uint32_t push_sign(uint32_t mask, float x) { return (mask << 1) | (f2u(x) >> 31); }
#endif

// ---- correctly rounded square root and quotient, in fewer instructions than the compiler's general sequences ----
// The compiler's IEEE sqrtf / division are a hardware estimate, a correction that makes the result correctly
// rounded, and -- more than a third of their instructions -- scaling and fix-ups for operands near the ends of the
// float range.  The cores below are those sequences without the scaling: bit-identical results wherever the
// scaling would not have acted, which the callers establish with one or two instructions for the whole wave and
// otherwise fall back to the full sequence.  (rt_hip_check_exact_math in rt_hip.hip compares both against the compiler's
// sequences on the device: every float for the square root and for 1 / sqrt, random and edge-case pairs for the
// quotient.)
#if defined(__HIP_DEVICE_COMPILE__)
// x: zero, +inf, NaN, at most -2^-96 (-> NaN), or finite and at least 2^-96.  The hardware square root is within one ulp;
// the two neighbours are tried with exact residuals (the compiler's own correction step).
RT_FN float sqrt_rn_core(float x)
{
  float s = __builtin_amdgcn_sqrtf(x);
  float const s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
  float const r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
  s = (r_dn <= 0.0f) ? s_dn : s;
  s = (r_up > 0.0f) ? s_up : s;
  return s;
}
// n / d for finite operands far from the ends of the float range (|d| in [2^-60, 2^60], |n| in [2^-100, 2^60] or
// zero, which may come out with the other sign): reciprocal estimate, one Newton step, quotient, two residual
// corrections -- the compiler's sequence between its scaling and its fix-up.
RT_FN float div_rn_core(float n, float d)
{
  float const r0 = __builtin_amdgcn_rcpf(d);
  float const e0 = __builtin_fmaf(-d, r0, 1.0f);
  float const r1 = __builtin_fmaf(e0, r0, r0);
  float const q0 = n * r1;
  float const e1 = __builtin_fmaf(-d, q0, n);
  float const q1 = __builtin_fmaf(e1, r1, q0);
  float const e2 = __builtin_fmaf(-d, q1, n);
  return __builtin_fmaf(e2, r1, q1);
}
// 1 / sqrt(x) as two correctly rounded operations (what v * (1.0f / sqrtf(x)) asks for), x being the squared
// length of a vector about to be normalised.  The short sequences are exact for 2^-96 <= x < 2^96 (the root lies
// in [2^-48, 2^48], where 1 / s needs no scaling) and hand a NaN through; for x = 0 they give NaN where the library
// gives infinity, and the zero vector times either is the same NaN vector.  The vectors
// the kernel normalises stay in that range when the library has established, for the whole launch (it sets
// RT_FLAG_LIBRARY_MATH otherwise), that radii, the near plane and the lights' direction vectors are between 2^-40
// and 2^40 long, roughness at most 2^40 and everything within 2^40 of the origin: a primary direction is at least
// the near plane long, a normal is hit point - centre (a radius long, or, where the hit point's own rounding
// exceeds the radius, zero or at least an ulp of coordinates that no hit closer than tmin leaves below 2^-24), and
// the sums n + ball, reflect + ball * roughness, light + view have a component of at least a half that another
// float of similar size can only cancel to zero or to 2^-25 or more.
RT_FN float inv_sqrt_rn(float x, bool library_math)
{
  if (library_math)
    return 1.0f / sqrtf(x);
  return div_rn_core(1.0f, sqrt_rn_core(x));
}
// (-b - sqrt(disc)) / d, the near root of Sphere::TraceNearest (ray_tracing/sphere.cpp:72-74), for lanes whose
// disc is not negative (the others get some value nobody uses); d = 2 dot(D, D) of a normalised direction, so
// about 2 (or NaN, which both sequences hand through).  No check of the operands is needed per call when the
// library has established two things for the whole launch (it launches the kernels built with Env::kLibraryMath otherwise):
//   - the scene and the camera lie within 2^40 of the origin, so b, disc and the numerator are finite (< 2^90);
//   - every tmin in use is at least 2^-20.  A discriminant below 2^-96, where sqrt_rn_core may be an ulp off, has a
//     root below 2^-48, less than half an ulp of any -b that could give t >= tmin: the numerator is -b whatever
//     that root's last bit.  A numerator below 2^-103, where div_rn_core may be off, gives t < tmin either way.
// So every distance the caller accepts is the library's, bit for bit, and every rejection is the same rejection.
RT_FN float near_root(float b, float disc, float d, bool library_math)
{
  if (library_math)
    return (-b - sqrtf(disc)) / d;
  return div_rn_core(-b - sqrt_rn_core(disc), d);
}
#else
RT_FN float sqrt_rn_core(float x) { return sqrtf(x); }
RT_FN float div_rn_core(float n, float d) { return n / d; }
RT_FN float inv_sqrt_rn(float x, bool) { return 1.0f / sqrtf(x); }
RT_FN float near_root(float b, float disc, float d, bool) { return (-b - sqrtf(disc)) / d; }
#endif
RT_FN float exact_sqrt(float x, bool library_math) { return library_math ? sqrtf(x) : sqrt_rn_core(x); }

// ---- exact helpers: the GLSL definitions glm follows, IEEE sqrt and division ----
RT_FN float minf(float a, float b) { return (b < a) ? b : a; }  // std::min
RT_FN float maxf(float a, float b) { return (a < b) ? b : a; }  // std::max
RT_FN float clampf(float x, float lo, float hi) { return minf(maxf(x, lo), hi); }
RT_FN f3 normalize3(f3 v, bool library_math) { return v * inv_sqrt_rn(dot3(v, v), library_math); }
RT_FN float fdot3(f3 a, f3 b) { return hw_fma(a.z, b.z, hw_fma(a.y, b.y, a.x * b.x)); }
RT_FN f3 normalize3_fast(f3 v) { return v * hw_rsq(dot3(v, v)); }
RT_FN f3 mix3(f3 x, f3 y, float a) { return x * (1.0f - a) + y * a; }
RT_FN float pow5(float x)
{
  float const x2 = x * x;
  return x2 * x2 * x;
}

// ---- Philox4x32-10, key = (pixel, seed), counter = (bounce, block, sample, 0) ----
RT_FN u4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
  for (int r = 0; r < 10; ++r)
  {
#if defined(__HIP_DEVICE_COMPILE__)
    // both halves of a 32 x 32 -> 64 bit product from one instruction (v_mad_u64_u32 issues like v_mul_hi_u32
    // alone; the compiler would take v_mul_hi_u32 and v_mul_lo_u32)
    unsigned long long p0, p1;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p0) : "v"(c0), "v"(0xD2511F53u) : "vcc");
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p1) : "v"(c2), "v"(0xCD9E8D57u) : "vcc");
    uint32_t const hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t const hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
#else
    uint32_t const hi0 = hw_mulhi(0xD2511F53u, c0);
    uint32_t const lo0 = 0xD2511F53u * c0;
    uint32_t const hi1 = hw_mulhi(0xCD9E8D57u, c2);
    uint32_t const lo1 = 0xCD9E8D57u * c2;
#endif
    c0 = hw_xor3(hi1, c1, k0);
    c1 = lo1;
    c2 = hw_xor3(hi0, c3, k1);
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

RT_FN float uniform24(uint32_t bits) { return (float)(bits >> 8) * (1.0f / 16777216.0f); }

// Uniform point in a ball of radius R from three uniforms: direction uniform on the sphere, radius
// R * u^(1/3).  Same law as glm::ballRand with a fixed draw count, so no lane waits on a rejection loop.
// The shadow-ray jitter draws its three uniforms from 256 cells each, so the functions of them are tabulated
// (built on the host with the C library, rt_prep.h build_jitter_table): per cell k, u = (k + 0.5) / 256,
//   [0, 512)     (z, s)      z = 1 - 2u, s = sqrt(1 - z^2)
//   [512, 1024)  (cos, sin)  of 2 pi u
//   [1024, 1280) 0.25 cbrt(u)
constexpr uint32_t RT_JITTER_CELLS = 256;
constexpr uint32_t RT_JITTER_FLOATS = 5u * RT_JITTER_CELLS;
RT_FN f3 jitter_ball(const float * tab, uint32_t k1, uint32_t k2, uint32_t k3)
{
  float const z = tab[2u * k1], s = tab[2u * k1 + 1u];
  float const c = tab[2u * RT_JITTER_CELLS + 2u * k2], sn = tab[2u * RT_JITTER_CELLS + 2u * k2 + 1u];
  float const rad = tab[4u * RT_JITTER_CELLS + k3];
  return mk3((s * c) * rad, (s * sn) * rad, z * rad);
}

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
  const f4 * spheres;          // n_padded x (cx, cy, cz, r*r): what the exact intersection reads
  const f4 * filter;           // n_padded x (centre - scene centre, the sphere's constant of the shade filter): see
                               // filter_sphere_2 and rt_prep.h flatten_scene
  const float * aux_rw;        // n_padded x the widened radius: what a shadow-cone test adds per sphere ...
  const float * aux_cone;      // ... and, for kernels specialised for one light, the whole sphere-dependent part of
                               // that test (rt_prep.h cone_constant)
  f3 centre;                   // the scene centre the filter records are relative to
  const f4 * camtab;           // n_padded x (centre - camera, filter bound): the primary-ray filter
  const uint32_t * sphere_mat; // n_spheres
  const f4 * materials;        // n_materials x RT_MAT_RECORDS: (albedo rgb, roughness), (refraction, type bits,
                               // f0^2, 1 / refraction), (r^2, 4 r^2, metal emissivity, 0) with r = max(roughness, 0.03)
  const f4 * lights;           // n_lights x RT_LIGHT_RECORDS: (direction xyz, cone cos), (colour rgb, cone sin),
                               // (unit direction to the light, 0); the cone is that of the light's shadow rays,
                               // see ShadowCone
  const float * jitter_tab;    // RT_JITTER_FLOATS, see jitter_ball
  uint32_t n_spheres, n_padded, n_materials, n_lights;  // n_padded: records in the sphere arrays (= n_spheres)
  uint32_t tile_size;          // filter records staged per tile when the scene is not resident
  f3 cone_axis;                // ShadowCone of light 0, for kernels specialised for one light (wave-uniform there)
  float cone_cos, cone_sin;
  float cone_inv_cos;          // 1 / cone_cos (1 for the cone that holds every direction)
  // camera, precomputed on the host exactly as Frame::ForEachRayImpl does (ray_tracing/frame.cpp:33-41)
  f3 cam_pos, cam_dir, cam_right, cam_up, background;
  float znear, zfar, dw, dh, cell_x, cell_y;
  // frame
  uint32_t width, height, s_row, s_count;  // s_count = s_row * s_row
  uint32_t flags, seed, max_depth, slice_index, slice_count;
  uint32_t s_row_mod;  // s_row % slice_count
  // Column masks (small resident scenes): the primary-ray filter runs once per sub-sample column, against the
  // column's middle direction, with every bound of the camera table lowered by |centre - camera| * col_delta;
  // col_delta bounds the distance between the middle direction and any sample direction of a column.
  uint32_t column_masks;      // 0: filter every primary ray on its own (col_delta is 0 then)
  float col_delta;
  // Uniform grid over the spheres (optional; rt_prep.h build_grid).  Cell (x, y, z) holds the indices
  // grid_items[grid_start[c] .. grid_start[c + 1]) with c = (z * dim.y + y) * dim.x + x, ascending; spheres too
  // large for the grid are listed in grid_big and tested for every ray.
  const uint32_t * grid_start;
  const uint32_t * grid_items;
  const uint32_t * grid_big;
  uint32_t grid_big_count;
  uint32_t grid_dim[3];
  f3 grid_org;
  float grid_cell, grid_inv_cell;
  // outputs
  float * partial;            // per column (item), the sum of the samples this slice owns: partial[3 * item + c]
  uint32_t n_items;           // width * height * s_row
  // Pixel sums taken inside the kernel (optional, ring_slots = 0: off; s_row must divide the 64-item window, so
  // that a window holds whole pixels).  A wave parks the column sums of the windows it works on in LDS and, when the
  // last column of a window is done, adds each pixel's columns in increasing sx (the order of the outer loop of
  // ray_tracing/frame.cpp:56) and writes the three sums to planes[c * n_pixels + pixel]: no column sum goes to
  // memory and `partial` is unused.  ring_slots: windows a wave may have unfinished at a time.
  float * planes;
  uint32_t n_pixels;
  uint32_t ring_slots;
  uint32_t s_shift;           // log2(s_row) when ring_slots != 0
  unsigned long long * ring_errors;  // raised by a wave that gave up waiting for a free slot (never, in a correct launch)
  // Slices: the samples of the frame, numbered item * s_row + sy, are dealt out to the slices in turn, so this
  // slice traces the samples sy = first_sample(item), + slice_count, ... of every column.  The work queue is
  // the items in order; position k is item k.
  uint32_t * queue_counter;   // next queue position of this slice to hand out
  // Slices by pixels instead of samples (optional, run_items = 0: off): the frame's items are dealt out in runs
  // of run_items (a multiple of the 64-item window and of s_row: whole pixels) to run_stride slices in turn, this
  // slice takes the runs run_first, run_first + run_stride, ... and traces every sample of their columns
  // (slice_count is 1 then).  Its queue has queue_items positions; position k is item queue_item(A, k).
  uint32_t queue_items;       // positions in this launch's queue (n_items without runs)
  uint32_t run_items, run_stride, run_first;
  uint32_t log_entries;       // kernels whose waves note their reservations (rt_hip.hip kLogEntries): how many a wave may note
  uint32_t log_quantum;       // ... and the number of queue positions every reservation of theirs is a multiple of
  uint32_t batch_items;       // queue positions a wave reserves at a time ...
  uint32_t batch_min;         // ... and near the end of the queue (see rt_hip.hip ItemPool)
  uint32_t guide_waves;       // waves of the launch
  const uint32_t * cancel;    // non-zero: stop handing out work (host-visible memory, read on refills only)
  unsigned long long * stats; // counters, see rt_hip.hip
  // Sample sharing at the end of the queue (optional, share_depth = 0: off).  Once a wave finds the queue used
  // up, its idle lanes take not-yet-started samples off the columns its busy lanes still work on.  A shared
  // sample's colour goes to shared[3 * (region * share_depth + (s_row - 1 - sy))], where `region` names the lane
  // that owns the column; the owner stores the sum of the samples it kept, appends (slot, first shared sy,
  // region * share_depth, 0) to `splits`, and fold_split adds the shared colours in sample order afterwards, so
  // the column sum is the same floating-point sum whoever traced the samples.
  float * shared;
  size_t shared_off;          // shared - partial, in floats: the colours of shared samples lie behind the column sums
  u4 * splits;
  uint32_t * split_count;
  uint32_t share_depth;       // samples at the end of a column that may be shared (<= s_row)
};

#define RT_REGION_COUNT 14
struct LaneCounters
{
  uint32_t samples, queries, hits, shaded, exact_tests;
  uint32_t filter_tests;  // rays and shadow cones run through the filter; each is n_spheres tests
  uint32_t prepared;      // columns the lane's wave ran through the filter while preparing windows: the same number
                          // in every lane of a wave (it counts once per wave)
  uint32_t cone_tests;    // spheres a shadow cone touched: each was intersected with the shadow rays still looking
  uint32_t trips, rounds;  // loop trips and stage-P rounds this lane sat through, busy or not
#if defined(RT_PROFILE_REGIONS)
  // Profiling builds only: wave clock ticks spent in each region of the loop (see RT_REGION).
  unsigned long long region[RT_REGION_COUNT];
  unsigned long long last_tick;
#endif
};

// RT_REGION(cnt, k) charges the time since the previous probe to region k.  Compiled in only with
// -DRT_PROFILE_REGIONS (tools/region_profile.py); the shipped library carries no probes.
#if defined(RT_PROFILE_REGIONS) && defined(__HIP_DEVICE_COMPILE__)
#define RT_REGION(cnt, k)                                                   \
  do                                                                        \
  {                                                                         \
    unsigned long long const now_ = __builtin_amdgcn_s_memtime();           \
    (cnt).region[k] += now_ - (cnt).last_tick;                              \
    (cnt).last_tick = now_;                                                 \
  } while (0)
// Slots 11 to 13 count loop trips of a wave, not time: a column's candidates against a primary ray, the spheres shadow
// cones touched, the candidates of scattered rays.
#define RT_COUNT_TRIP(cnt, k) ((cnt).region[k] += 1ull)
#elif defined(RT_ISA_MARKS) && defined(__HIP_DEVICE_COMPILE__)
// Listing aid: leaves "; MARK k" comments in the assembly so that instructions can be counted per region.
#define RT_REGION(cnt, k) asm volatile("; MARK " #k)
#define RT_COUNT_TRIP(cnt, k) ((void)0)
#else
#define RT_REGION(cnt, k) ((void)0)
#define RT_COUNT_TRIP(cnt, k) ((void)0)
#endif

// The filter.  With d = C - O and unit D, the ray can hit the sphere in front of O only if
// dot(d, D) >= sqrt(c), c = |d|^2 - r^2.  The bound is taken from c - m (|d|^2 + r^2), m = RT_FILTER_MARGIN
// (the filter records hold the radius widened to rw >= r (1 + m), so rw^2 >= r^2 (1 + m)), so that rounding
// here (FMAs, approximate sqrt: a few ulps of |d|^2) can never reject a ray the exact intersection would
// accept; an origin on or inside the widened sphere gives bound 0.  r2f: the square of the widened radius.
RT_FN float filter_bound(float dx, float dy, float dz, float r2f)
{
  float const dd = hw_fma(dz, dz, hw_fma(dy, dy, dx * dx));
  // sqrt of a negative number is a NaN, which max() turns into the 0 wanted for an origin inside the sphere
  return hw_max(hw_sqrt(hw_fma(dd, (float)(1.0 - RT_FILTER_MARGIN), -r2f)), 0.0f);
}

// Sphere::TraceNearest, ray_tracing/sphere.cpp:63-81, operation for operation (a = dot(D, D) hoisted).
// The reference takes min((-b - s) / 2a, (-b + s) / 2a); a > 0 and s >= 0, and IEEE division by a positive
// number is monotone, so the minimum is always the first quotient, bit for bit.
RT_FN bool exact_hit(f3 o, f3 dir, float a, f4 s, float tmin, float tmax, float & t_out, bool library_math)
{
  f3 const d = o - mk3(s.x, s.y, s.z);
  float const b = 2.0f * dot3(d, dir);
  float const c = dot3(d, d) - s.w;
  float const disc = b * b - 4 * a * c;
  if (disc < 0.0f)
    return false;
  float const t = near_root(b, disc, 2.0f * a, library_math);
  if (t < tmin || t > tmax)
    return false;
  t_out = t;
  return true;
}

// The same test without early exits, for resolving two candidates side by side.  Same operations, same
// results (a negative discriminant is rejected whatever its square root turns into).
RT_FN bool exact_hit_flat(f3 o, f3 dir, float a, f4 s, float tmin, float tmax, float & t_out, bool library_math)
{
#if defined(__HIP_DEVICE_COMPILE__)
  // Keeps 4a and 2a from being computed ahead of the resolution loops for all six rays (a dozen registers held
  // across the whole pass for two multiplications per test).
  asm volatile("" : "+v"(a));
#endif
  f3 const d = o - mk3(s.x, s.y, s.z);
  float const b = 2.0f * dot3(d, dir);
  float const c = dot3(d, d) - s.w;
  float const disc = b * b - 4 * a * c;
  float const t = near_root(b, disc, 2.0f * a, library_math);
  t_out = t;
  return !(disc < 0.0f) && !(t < tmin || t > tmax);  // the reference's two rejections, NaNs included
}

constexpr uint32_t RT_SHARED_SLOT = 0x80000000u;

// Per-lane state.
struct Lane
{
  // work item being accumulated: sub-sample column sx of one pixel (its s_row samples sy = 0..s_row-1)
  uint32_t slot;    // position of the item in this slice (where its column sum goes); with RT_SHARED_SLOT set,
                    // the lane traces a shared sample and the rest is its index in TraceArgs::shared
  uint32_t pixel;
  uint32_t sx;
  uint32_t sy;      // next sample of the column
  uint32_t sy_end;  // one past the last sample this lane traces itself (s_row unless samples were shared out)
  float sdx, dyp;   // sdx of the column, dy of the pixel (ray_tracing/frame.cpp:49,52,58)
  f3 sum;
  bool alive, need_item;
  uint32_t cm0, cm1;  // candidates of the column (sphere j in bit 31 - j of cm0, sphere 32 + j in bit 31 - j of cm1)
                      // when column masks are on
  // current path
  f3 acc_l;           // forward accumulation: product of the l entries pushed so far
  float acc_e;        // ... and of emissivity_k ^ k
  uint32_t sample;  // sample index (RNG key)
  uint32_t depth;   // entries pushed so far ("scatterResultsCount")
  bool has_hit;     // a hit waits to be shaded
  bool shading;     // a shade pass sequence (one pass per light) is in flight
  f3 hp, hn, hd;    // hit point, normal, incoming direction
  int hidx;         // sphere hit
  // shading in flight
  uint32_t li;      // light of the current shade pass
  f3 spec, diffuse, m_albedo, sdir;
  float e, m_refraction;
  float nx_t;
  int nx_i;  // nearest hit of the scattered ray (found during the first light's pass)
};

// Frame::ForEachRayImpl, ray_tracing/frame.cpp:61-62: direction of sub-sample (sx, sy) given the column's sdx.
RT_FN f3 primary_dir(const TraceArgs & A, float sdx, float dyp, uint32_t sy, bool library_math)
{
  float const sdy = dyp - A.cell_y * (float)sy / (float)A.s_row;
  return normalize3((A.cam_dir * A.znear + A.cam_up * sdy) + A.cam_right * sdx, library_math);
}

// First sample (sy) of column `item` that this slice owns; >= s_row when it owns none.  The owner of sample
// item * s_row + sy is that number modulo slice_count (slice_count < 65536, so the products below fit).
// Item at queue position `pos` (see TraceArgs::run_items).
RT_FN uint32_t queue_item(const TraceArgs & A, uint32_t pos)
{
  if (A.run_items == 0u)
    return pos;
  uint32_t const run = pos / A.run_items;
  return (run * A.run_stride + A.run_first) * A.run_items + (pos - run * A.run_items);
}

RT_FN uint32_t first_sample(const TraceArgs & A, uint32_t item)
{
  if (A.s_row_mod == 0u)
    return A.slice_index;  // slice_count divides s_row (or is 1): the same samples of every column
  uint32_t const at = ((item % A.slice_count) * A.s_row_mod) % A.slice_count;  // owner of sy = 0
  return (A.slice_index + A.slice_count - at) % A.slice_count;
}

RT_FN void begin_item(const TraceArgs & A, Lane & L, uint32_t item)
{
  uint32_t const pixel = item / A.s_row;
  uint32_t const sx = item - pixel * A.s_row;
  uint32_t const y = pixel / A.width;
  uint32_t const x = pixel - y * A.width;
  L.pixel = pixel;
  L.sx = sx;
  L.sy = first_sample(A, item);
  L.sy_end = A.s_row;
  L.dyp = (-2.0f * (float)y / (float)A.height + 1.0f) * A.dh;        // frame.cpp:49
  float const dxp = (2.0f * (float)x / (float)A.width - 1.0f) * A.dw;  // frame.cpp:52
  L.sdx = dxp + A.cell_x * (float)sx / (float)A.s_row;                 // frame.cpp:58
  L.sum = mk3(0.0f, 0.0f, 0.0f);
}

// Samples a lane can share out right now (see TraceArgs::shared): those of its own column it has not started,
// less the one it is about to start, and only the last share_depth of the column.
RT_FN uint32_t spare_samples(const TraceArgs & A, const Lane & L)
{
  if (!L.alive || L.need_item || (L.slot & RT_SHARED_SLOT) != 0u)
    return 0u;
  // the lane's samples are sy, sy + slice_count, ... below sy_end
  uint32_t from = L.sy + ((!L.has_hit && !L.shading) ? A.slice_count : 0u);
  uint32_t const first_shareable = (A.s_row > A.share_depth) ? A.s_row - A.share_depth : 0u;
  if (from < first_shareable)
    from += (first_shareable - from + A.slice_count - 1u) / A.slice_count * A.slice_count;
  return (L.sy_end > from) ? (L.sy_end - from + A.slice_count - 1u) / A.slice_count : 0u;
}

// The j-th last of `spare` spare samples of a lane whose samples end below sy_end and start at `from` (as in
// spare_samples): the owner keeps everything below the smallest one handed out.
RT_FN uint32_t spare_sample_sy(const TraceArgs & A, uint32_t sy, uint32_t sy_end, uint32_t j)
{
  uint32_t const last = sy + (sy_end - 1u - sy) / A.slice_count * A.slice_count;  // the lane's last sample
  return last - j * A.slice_count;
}

// Makes an idle lane trace sample sy of the column another lane owns (the owner's ray set-up is passed in).
RT_FN void take_shared_sample(const TraceArgs & A, Lane & L, uint32_t region_base, uint32_t pixel,
                              uint32_t sx, float sdx, float dyp, uint32_t cm0, uint32_t cm1, uint32_t sy)
{
  L.pixel = pixel;
  L.sx = sx;
  L.sdx = sdx;
  L.dyp = dyp;
  L.cm0 = cm0;
  L.cm1 = cm1;
  L.sy = sy;
  L.sy_end = sy + 1u;
  L.sum = mk3(0.0f, 0.0f, 0.0f);
  L.slot = RT_SHARED_SLOT | (region_base + (A.s_row - 1u - sy));
  L.alive = true;
  L.need_item = false;
  L.has_hit = false;
  L.shading = false;
}

// Completes the sum of a column whose last samples were traced by other lanes: the owner's partial sum, then
// the shared colours in increasing sy (every step-th sample from the first shared one), the additions the owner
// would have made itself.
RT_FN void fold_split(float * partial, const float * shared, u4 rec, uint32_t s_row, uint32_t step)
{
  float * const c = partial + 3u * (size_t)rec.x;
  f3 sum = mk3(c[0], c[1], c[2]);
  for (uint32_t sy = rec.y; sy < s_row; sy += step)
  {
    const float * const v = shared + 3u * (size_t)(rec.z + (s_row - 1u - sy));
    sum = sum + mk3(v[0], v[1], v[2]);
  }
  store3(c, sum);
}

// A finished column (or shared sample) goes to memory: one 12-byte store; every entry is written by exactly one
// lane.  The owner of a column some of whose samples were shared out leaves a record for the fold pass.
template <class Env>
RT_FN void store_column(const TraceArgs & A, const Lane & L, Env & env)
{
  bool const shared_sample = (L.slot & RT_SHARED_SLOT) != 0u;
  // one base pointer and an offset, not a choice between two pointers (which the compiler turns into a load
  // of the chosen pointer from the argument block, a memory round trip in every round that ends a column)
  size_t const at = 3u * (size_t)(L.slot & ~RT_SHARED_SLOT) + (shared_sample ? A.shared_off : (size_t)0);
  store3(A.partial + at, L.sum);
  if (!shared_sample && L.sy_end < A.s_row)
    env.record_split(L.slot, L.sy_end);
}

// What an environment hands out with a work item.  When `prepared`, the column's ray set-up and candidate
// masks were computed ahead (prepare_column); otherwise the lane runs begin_item itself.
struct ItemInfo
{
  uint32_t pos;           // the item; RT_INVALID: the slice is used up, or (stalled) nothing can be handed out right now
  uint32_t slot;          // what Lane::slot becomes: where the column's sum goes (the item, unless the environment
                          // keeps the sums of its windows itself)
  bool stalled;           // wave-uniform: the environment could not open the next window yet; lanes without an item
                          // ask again in a later round
  bool prepared;
  float sdx, dyp;
  uint32_t cm0, cm1, pixel, sx, sy0;
  uint32_t filter_tests;  // columns the wave ran through the filter while preparing a window in this call (the same
                          // number in every lane)
};

struct ColumnEntry
{
  float sdx, dyp;
  uint32_t cm0, cm1, pixel, sx, sy0, pad;  // sy0: first_sample of the column
};

// Filter of one ray against `count` (<= RT_CHUNK) camera-table entries; defined below.
template <uint32_t kUnroll>
RT_FN uint32_t filter_chunk_1(const f4 * tab, uint32_t count, f3 dir);

// Ray set-up of one column (the arithmetic of begin_item) and its candidate masks: the camera table, whose
// bounds were lowered for this purpose, filtered with the direction through the middle of the column.
// camtab: all A.n_padded (<= 64) entries.
RT_FN ColumnEntry prepare_column(const TraceArgs & A, const f4 * camtab, uint32_t item)
{
  ColumnEntry e;
  uint32_t const pixel = item / A.s_row;
  uint32_t const sx = item - pixel * A.s_row;
  uint32_t const y = pixel / A.width;
  uint32_t const x = pixel - y * A.width;
  e.pixel = pixel;
  e.sx = sx;
  e.dyp = (-2.0f * (float)y / (float)A.height + 1.0f) * A.dh;
  float const dxp = (2.0f * (float)x / (float)A.width - 1.0f) * A.dw;
  e.sdx = dxp + A.cell_x * (float)sx / (float)A.s_row;
  float const mid = 0.5f * (float)(A.s_row - 1u) / (float)A.s_row;
  float const sdy = e.dyp - A.cell_y * mid;
  f3 const v = (A.cam_dir * A.znear + A.cam_up * sdy) + A.cam_right * e.sdx;
  f3 const dc = normalize3_fast(v);
  // the candidates as a group mask (sphere j in bit 63 - j): cm0 the upper word, spheres 0..31, cm1 the lower
  uint32_t const c0 = (A.n_padded < RT_CHUNK) ? A.n_padded : RT_CHUNK;
  e.cm0 = (c0 != 0u) ? (filter_chunk_1<RT_UNROLL>(camtab, c0, dc) << (32u - c0)) : 0u;
  e.cm1 = (A.n_padded > RT_CHUNK)
              ? (filter_chunk_1<RT_UNROLL>(camtab + RT_CHUNK, A.n_padded - RT_CHUNK, dc) << (32u - (A.n_padded - RT_CHUNK)))
              : 0u;
  e.sy0 = first_sample(A, item);
  e.pad = 0u;
  return e;
}

// The material constants a hit needs, from the three records of its material.
struct MaterialView
{
  uint32_t type;
  float roughness, refraction;
  float f2;              // ((1 - refraction) / (1 + refraction))^2
  float inv_refraction;  // 1 / refraction
  float r2, four_r2;     // max(roughness, 0.03)^2 and four times it
  float e_metal;         // 0.9 (1 - roughness)
};

// Material::Scatter for the three materials, ray_tracing/default_materials.cpp:11-67.  Matte and Metal share
// one normalisation (of n + ball, or of reflect + ball * roughness).
RT_FN void scatter(const MaterialView & m, f3 rd, f3 n, f3 ball, float xi, f3 & out_dir, float & out_e, bool library_math)
{
  float const dn = dot3(n, rd);
  f3 const refl = rd - n * dn * 2.0f;  // glm::reflect(I, N) = I - N * dot(N, I) * 2
  if (m.type != RT_GLASS)
  {
    bool const matte = m.type == RT_MATTE;
    f3 const v = matte ? n + ball : refl + ball * m.roughness;
    f3 const r = normalize3(v, library_math);
    out_dir = r;
    out_e = matte ? 0.7f : ((dot3(n, r) <= 0.0f) ? 0.0f : m.e_metal);
  }
  else
  {
    float vdn = dot3(rd, n);
    f3 outward;
    float eta;
    if (vdn > 0.0f)
    {
      outward = -n;
      eta = m.refraction;
      vdn *= eta;
    }
    else
    {
      outward = n;
      eta = m.inv_refraction;
      vdn = -vdn;
    }
    // glm::refract(I, N, eta)
    float const d = dot3(outward, rd);
    float const k = 1.0f - eta * eta * (1.0f - d * d);
    f3 refr = mk3(0.0f, 0.0f, 0.0f);
    if (k >= 0.0f)
      refr = rd * eta - outward * (eta * d + exact_sqrt(k, library_math));  // k: zero or at least 2^-49
    float ref = 1.0f;
    if (dot3(refr, refr) > RT_EPS)
      ref = m.f2 + (1.0f - m.f2) * pow5(1.0f - vdn);
    out_dir = (xi < ref) ? refl : refr;
    out_e = 1.0f;
  }
}

// DirectionalLight::GetSpecular, ray_tracing/directional_light.cpp:37-60.  The half vector and the cosines are
// computed exactly (the Beckmann exponent multiplies their rounding by up to 1 / 0.03^2); the three quotients
// and the exponential behind them only scale the result and use the hardware reciprocal and exp2 (~1e-6).  The
// clamps and minima are the hardware's (one instruction each): the same values for every operand but a NaN, for which
// glm::clamp and std::min (compare and select) hand the NaN through and these give the lower bound, or the other
// operand.  A NaN arises in one case: a light that shines exactly along the viewing ray towards the camera
// (ldir + rd == 0), whose half vector is the normalisation of zero.  The reference's sample is a NaN then; here the
// specular term of that hit is 0 and the sample finite (tests/test_emu_vs_oracle.py pins both).
RT_FN f3 specular(f3 ldir, f3 lcol, f3 albedo, const MaterialView & m, f3 rd, f3 n, bool library_math)
{
  float const vdn = hw_clamp(dot3(-rd, n), 0.0f, 1.0f);
  float const ndl = hw_clamp(dot3(n, -ldir), 0.0f, 1.0f);
  f3 const h = -normalize3(ldir + rd, library_math);
  float const ndh = hw_clamp(dot3(n, h), RT_EPS, 1.0f);
  float const vdh = hw_clamp(dot3(-rd, h), 0.0f, 1.0f);
  float const g = (fabsf(vdh) >= RT_EPS) ? hw_min(1.0f, 2.0f * ndh * hw_min(vdn, ndl) * hw_rcp(vdh)) : 1.0f;
  float const ndh2 = ndh * ndh;
  float const q = (ndh2 - 1.0f) * hw_rcp(m.r2 * ndh2);
  float const dm = hw_exp2(q * 1.44269504088896340736f) * hw_rcp(m.four_r2 * ndh2 * ndh2);
  float const f = m.f2 + (1.0f - m.f2) * pow5(1.0f - vdn);
  float const spec = hw_clamp(g * dm * f, 0.0f, 1.0f);
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
    if (fabsf(rest) < RT_EPS)
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

// e^k for k < 16 by squaring.
RT_FN float pow_small(float e, uint32_t k)
{
  float p = (k & 1u) ? e : 1.0f;
  float b = e * e;
  p = (k & 2u) ? p * b : p;
  b = b * b;
  p = (k & 4u) ? p * b : p;
  b = b * b;
  p = (k & 8u) ? p * b : p;
  return p;
}

// Entry `depth` of a path.  kForward: the reverse loop of demos/demo_frame_cpu.cpp:60-75 multiplies
// l_i * (e_{i+1} ... e_{n-1}) over all entries, which is prod(l_i) * prod(e_k ^ k) when its early exit
// (|rest| < kEps) cannot trigger -- the host decides that per scene from the materials' emissivities -- so
// two running products replace the stack.  Otherwise the entries go to the lane's stack and path_colour
// replays the loop as written.
template <bool kForward>
RT_FN void path_push(Lane & L, f4 * stack, uint32_t stride, f3 l, float e)
{
  if constexpr (kForward)
  {
    L.acc_l = (L.depth == 0u) ? l : L.acc_l * l;
    float const ek = pow_small(e, L.depth);
    L.acc_e = (L.depth == 0u) ? 1.0f : L.acc_e * ek;
  }
  else
    push(stack, stride, L.depth, l, e);
  L.depth += 1;
}

template <bool kForward>
RT_FN f3 path_colour(const Lane & L, const f4 * stack, uint32_t stride)
{
  if constexpr (kForward)
    return (L.acc_l * L.acc_e) * 0.5f;
  else
    return accumulate(stack, stride, L.depth);
}

// The filter keeps its outcome as the sign of dot(d, D) - bound, accumulated in one FMA chain that starts from
// -bound, and shifts that sign bit into the lane's mask: a set bit means "rejected".  chunk_candidates turns
// the `count` bits shifted in into the candidate mask (entry j of the chunk in bit count - 1 - j).
RT_FN uint32_t chunk_candidates(uint32_t rejected, uint32_t count)
{
  uint32_t const valid = (count >= 32u) ? 0xFFFFFFFFu : ((1u << count) - 1u);
  return ~rejected & valid;
}

// Filter of one ray against `count` (<= RT_CHUNK) camera-table entries
// (centre - camera, bound).
template <uint32_t kUnroll>
RT_FN uint32_t filter_chunk_1(const f4 * tab, uint32_t count, f3 dir)
{
  uint32_t mask = 0;
  uint32_t j = 0;
  for (; j + kUnroll <= count; j += kUnroll)
  {
#pragma unroll
    for (uint32_t u = 0; u < kUnroll; ++u)
    {
      f4 const e = tab[j + u];
      mask = push_sign(mask, hw_fma(e.z, dir.z, hw_fma(e.y, dir.y, hw_fma(e.x, dir.x, -e.w))));
    }
  }
  for (; j < count; ++j)
  {
    f4 const e = tab[j];
    mask = push_sign(mask, hw_fma(e.z, dir.z, hw_fma(e.y, dir.y, hw_fma(e.x, dir.x, -e.w))));
  }
  return chunk_candidates(mask, count);
}

// The cone that holds the shadow rays of one light.  DirectionalLight::TraceLight
// (ray_tracing/directional_light.cpp:20) shoots along normalize(-L + ballRand(0.25)): every such direction is
// within asin(0.25 / |L|) of the unit vector `axis` = -L / |L|.  A ray inside the cone can hit sphere (C, r) only
// if the angle between d = C - O and the axis is at most theta + alpha, sin(alpha) = r / |d|, that is
//   dot(d, axis) >= cos(theta) sqrt(|d|^2 - r^2) - sin(theta) r.
// cos_m / sin_m are cos(theta) rounded down and sin(theta) rounded up by a thousandth (rt_prep.h shadow_cone), the
// square root is the filter's own bound (already on the safe side) and r the widened radius, so whatever one of the
// five per-ray tests or the exact intersection accepts, this accepts.  A light too short for a cone (|L| close to
// the jitter radius) gets cos_m = 0, sin_m = 3e38: every sphere passes.
struct ShadowCone
{
  f3 axis;
  float cos_m, sin_m;
  float cos2_m;  // cos_m^2
  float oa;      // one-light kernels: dot(origin of the pass, axis) / cos_m, lowered by its own rounding error
};

// What a shade pass knows about its origin O and its scattered ray D before it meets the spheres.
struct ShadeOrigin
{
  f3 o;       // O' = O - scene centre
  f3 a;       // 2 (1 - m) O'
  float p;    // (RT_FILTER_SLACK - (1 - m)) |O'|^2
  f3 dir;     // D
  float nod;  // -dot(O', D)
};

RT_FN ShadeOrigin shade_origin(f3 o, f3 centre, f3 dir)
{
  ShadeOrigin F;
  F.o = o - centre;
  float const twice = (float)(2.0 * (1.0 - RT_FILTER_MARGIN));
  F.a = mk3(F.o.x * twice, F.o.y * twice, F.o.z * twice);
  F.p = (float)(RT_FILTER_SLACK - (1.0 - RT_FILTER_MARGIN)) * fdot3(F.o, F.o);
  F.dir = dir;
  F.nod = -fdot3(F.o, dir);
  return F;
}

// One sphere against the shadow cone and the scattered ray of a pass (common origin O).  Both tests are taken in
// squared form, so that no square root is needed per sphere: with d = C - O and b2 = |d|^2 (1 - m) - rw^2, the
// square of filter_bound's value (negative for an origin inside the widened sphere, where filter_bound gives 0),
//   ray   u = dot(d, D) >= sqrt(b2)                  <=>  u >= 0 and u^2 >= b2           <=>  u |u| - b2 >= 0
//   cone  s = dot(d, axis) + sin rw >= cos sqrt(b2)  <=>  s >= 0 and s^2 >= cos^2 b2     <=>  s |s| - cos^2 b2 >= 0
// (for b2 < 0 the right-hand forms accept a little more than u >= 0 and s >= 0: the safe side).
// Neither needs d itself.  With C' and O' relative to the scene centre,
//   -b2 = [rw^2 - (1 - m) |C'|^2] + dot(C', 2 (1 - m) O') - (1 - m) |O'|^2,      u = dot(C', D) - dot(O', D):
// a sphere's record is (C', its bracket), a pass brings 2 (1 - m) O', its own last term and -dot(O', D), and a
// sphere costs two chains of three FMAs and an addition where the offset, its square and the bound took seven
// instructions before the ray was looked at.  The products are as large as the points are far from the centre, so
// their rounding (under 2.2e-6 (|C'|^2 + |O'|^2) for both tests together) is not covered by m |d|^2; the bracket
// and the pass's term each carry RT_FILTER_SLACK times their squared distance instead (rt_prep.h flatten_scene,
// shade_origin): the filter lets through a little more for a scene far from its own centre, never less.
// aux: rw, or -- kKnownCone, kernels whose one light is known when the scene is made -- the part of s / cos that
// depends on the sphere alone, (dot(C', axis) + sin rw) / cos, raised by the rounding error it and the subtraction
// below can have (rt_prep.h cone_constant); s / cos is then one subtraction, of cone.oa = dot(O', axis) / cos, and
// the test reads (s / cos) |s / cos| - b2 >= 0.
template <bool kKnownCone>
RT_FN void filter_sphere_2(f4 s, float aux, const ShadeOrigin & F, const ShadowCone & cone, uint32_t & cone_mask,
                           uint32_t & ray_mask)
{
  float const nb2 = hw_fma(s.z, F.a.z, hw_fma(s.y, F.a.y, hw_fma(s.x, F.a.x, s.w))) + F.p;  // -b2
  float const u = hw_fma(s.z, F.dir.z, hw_fma(s.y, F.dir.y, hw_fma(s.x, F.dir.x, F.nod)));
  ray_mask = push_sign(ray_mask, hw_fma(u, hw_abs(u), nb2));
  if constexpr (kKnownCone)
  {
    float const sc = aux - cone.oa;
    cone_mask = push_sign(cone_mask, hw_fma(sc, hw_abs(sc), nb2));
  }
  else
  {
    float const dx = s.x - F.o.x, dy = s.y - F.o.y, dz = s.z - F.o.z;
    float const sc = hw_fma(dz, cone.axis.z, hw_fma(dy, cone.axis.y, hw_fma(dx, cone.axis.x, cone.sin_m * aux)));
    cone_mask = push_sign(cone_mask, hw_fma(sc, hw_abs(sc), cone.cos2_m * nb2));
  }
}

// The shadow cone and the scattered ray against `count` (<= RT_CHUNK) spheres.
template <uint32_t kUnroll, bool kKnownCone>
RT_FN void filter_chunk_2(const f4 * sph, const float * aux, uint32_t count, const ShadeOrigin & F,
                          const ShadowCone & cone, uint32_t & cone_mask, uint32_t & ray_mask)
{
  cone_mask = 0;
  ray_mask = 0;
  uint32_t j = 0;
  for (; j + kUnroll <= count; j += kUnroll)
  {
#pragma unroll
    for (uint32_t u = 0; u < kUnroll; ++u)
      filter_sphere_2<kKnownCone>(sph[j + u], aux[j + u], F, cone, cone_mask, ray_mask);
  }
  for (; j < count; ++j)
    filter_sphere_2<kKnownCone>(sph[j], aux[j], F, cone, cone_mask, ray_mask);
  cone_mask = chunk_candidates(cone_mask, count);
  ray_mask = chunk_candidates(ray_mask, count);
}

// Candidates of a group of two chunks as one 64-bit mask with sphere j of the group in bit 63 - j (the chunk masks
// hold entry j of a chunk in bit count - 1 - j).
RT_FN unsigned long long group_mask(uint32_t lo, uint32_t count_lo, uint32_t hi, uint32_t count_hi)
{
  uint32_t const top = lo << (32u - count_lo);  // 1 <= count_lo <= 32
  uint32_t const bottom = (count_hi != 0u) ? (hi << (32u - count_hi)) : 0u;
  return ((unsigned long long)top << 32) | (unsigned long long)bottom;
}

// Takes the candidate with the lowest sphere index out of a (non-empty) group mask; returns its index in the group.
RT_FN uint32_t pop_group(unsigned long long & mask)
{
  uint32_t const j = (uint32_t)hw_clz64(mask);
  mask &= ~(0x8000000000000000ull >> j);
  return j;
}

// Takes the candidate with the lowest sphere index out of a chunk mask; returns its index in the chunk.
RT_FN uint32_t pop_candidate(uint32_t & mask, uint32_t count)
{
  uint32_t const bit = 31u - (uint32_t)hw_clz(mask);
  mask &= ~(1u << bit);
  return count - 1u - bit;
}

// What the enclosing kernel provides (Env):
//   kLibraryMath      exact square roots and quotients come from the compiler's general IEEE sequences, not from
//                     sqrt_rn_core / div_rn_core: for scenes outside the bounds under which those are exact
//   kPrimaryUnroll    camera-table entries per trip of the per-sample primary filter's main loop
//   kFilterUnroll     spheres per trip of the six-ray filter's main loop (4 costs a dozen registers more than
//                     2, the difference between three and four waves per SIMD)
//   any(p)            true if p holds for any lane sharing this lane's loop trips (the wave; the block when
//                     sphere tiles are re-staged inside the loop)
//   count(p)          number of such lanes for which p holds
//   any_wave(p)       true if p holds for any lane of the wave
//   fetch_item(L, need, info)  hands a fresh work item to each lane that asks (info.pos == RT_INVALID when the
//                     slice is used up), prepared or not (see ItemInfo)
//   column_sx(L)      sx of the lane's column: L.sx, or what an environment that names columns by their place in a
//                     window derives from L.slot (one register fewer across the loop)
//   finish_columns(L, finishing)  called by the whole wave in every round: the sums of the lanes with `finishing` (a
//                     complete column or shared sample) leave their lanes (store_column, or the environment's
//                     own pixel sums, TraceArgs::ring_slots)
//   sphere(i)         exact (cx, cy, cz, r*r) of sphere i;  sphere_material(i);  material(m, k) and
//                     light(l, k): record k of material m / light l;  jitter_table()
//   kCompactsRays     the lanes that share loop trips hand their primary rays to the first of them before a round
//                     over staged tiles: gather_rays(mine, dir, a) turns (has a ray, the ray) into what this lane
//                     traces and returns the lane the ray came from; return_hits(mine, home, active, t, i) takes
//                     the results back to the lanes with `active` (both are calls of every lane)
//   tiles()           number of sphere tiles;  load_tile(k, primary) stages tile k of the filter records
//                     (the camera table when primary) and returns its padded entry count;  tile_base(k),
//                     tile_data(primary);  tile_aux(): the per-sphere cone term of the staged filter records
//                     (TraceArgs::aux_cone for a kernel specialised for one light, else TraceArgs::aux_rw)
// One axis of a walk through the uniform grid: the cell coordinate and the ray parameter at the next wall.
struct GridAxis
{
  float oo, dd, inv;  // origin relative to the grid, direction component and its reciprocal
  int c, step;
  float t_next;

  // Narrows [t0, t1] to the slab [0, extent]; false if the ray misses it.
  RT_MEMBER bool clip(float origin, float dir, float extent, float & t0, float & t1)
  {
    oo = origin;
    dd = dir;
    if (dir == 0.0f)
    {
      inv = 0.0f;
      return origin >= 0.0f && origin <= extent;
    }
    inv = 1.0f / dir;
    float const ta = (0.0f - origin) * inv, tb = (extent - origin) * inv;
    float const lo = (ta < tb) ? ta : tb, hi = (ta < tb) ? tb : ta;
    t0 = (lo > t0) ? lo : t0;
    t1 = (hi < t1) ? hi : t1;
    return true;
  }

  RT_MEMBER void start(float t0, float cell, float inv_cell, int dim)
  {
    int ci = (int)floorf((oo + dd * t0) * inv_cell);
    ci = ci < 0 ? 0 : (ci > dim - 1 ? dim - 1 : ci);
    c = ci;
    if (dd > 0.0f)
    {
      step = 1;
      t_next = ((float)(ci + 1) * cell - oo) * inv;
    }
    else if (dd < 0.0f)
    {
      step = -1;
      t_next = ((float)ci * cell - oo) * inv;
    }
    else
    {
      step = 0;
      t_next = 3.0e38f;
    }
  }

  // Steps into the next cell; false when that leaves the grid.  The parameter of the next wall is taken from the
  // cell index each time, not summed up wall by wall: over hundreds of cells a running sum drifts by more than
  // the margin by which spheres are listed in neighbouring cells.
  RT_MEMBER bool advance(int dim, float cell)
  {
    c += step;
    if (step != 0)
      t_next = ((float)(c + (step > 0 ? 1 : 0)) * cell - oo) * inv;
    return c >= 0 && c < dim;
  }
};

// One ray through the uniform grid.  any_hit: stop at the first sphere hit in [tmin, tmax] (shadow rays);
// otherwise the nearest hit, the lowest sphere index among equal distances -- what the linear scan of
// TraceNearestInHitableCollection (ray_tracing/hitable_object.hpp:71-87) returns.  Every candidate goes through
// exact_hit, so distances are the brute-force ones bit for bit; the grid only decides which spheres are tried.
// A sphere is listed in every cell its slightly inflated bounding box touches, so the cell that holds a hit
// point always lists the sphere hit, whichever side of a cell wall rounding puts the walk on; the walk ends
// once the best hit lies before the exit of the current cell.
template <class Env>
RT_FN void grid_trace(const TraceArgs & A, Env & env, bool active, f3 o, f3 d, float a, float tmin, float tmax,
                      bool any_hit, float & best_t, int & best_i, LaneCounters & cnt)
{
  best_i = -1;
  best_t = 0.0f;
  if (!active)
    return;
  for (uint32_t k = 0; k < A.grid_big_count; ++k)
  {
    int const i = (int)A.grid_big[k];
    float t;
    cnt.exact_tests += 1;
    if (exact_hit(o, d, a, env.sphere(i), tmin, tmax, t, Env::kLibraryMath) && (best_i < 0 || t < best_t || (t == best_t && i < best_i)))
    {
      best_i = i;
      best_t = t;
      if (any_hit)
        return;
    }
  }
  // clip the ray to the grid's box (slabs), a little generously
  float t0 = tmin, t1 = tmax;
  GridAxis ax[3];
  if (!ax[0].clip(o.x - A.grid_org.x, d.x, (float)A.grid_dim[0] * A.grid_cell, t0, t1) ||
      !ax[1].clip(o.y - A.grid_org.y, d.y, (float)A.grid_dim[1] * A.grid_cell, t0, t1) ||
      !ax[2].clip(o.z - A.grid_org.z, d.z, (float)A.grid_dim[2] * A.grid_cell, t0, t1))
    return;
  float const slack = 1e-4f * A.grid_cell;
  if (t0 > t1 + slack)
    return;
  ax[0].start(t0, A.grid_cell, A.grid_inv_cell, (int)A.grid_dim[0]);
  ax[1].start(t0, A.grid_cell, A.grid_inv_cell, (int)A.grid_dim[1]);
  ax[2].start(t0, A.grid_cell, A.grid_inv_cell, (int)A.grid_dim[2]);
  for (;;)
  {
    uint32_t const cell = ((uint32_t)ax[2].c * A.grid_dim[1] + (uint32_t)ax[1].c) * A.grid_dim[0] + (uint32_t)ax[0].c;
    uint32_t const begin = A.grid_start[cell], end = A.grid_start[cell + 1u];
    for (uint32_t k = begin; k < end; ++k)
    {
      int const i = (int)A.grid_items[k];
      float t;
      cnt.exact_tests += 1;
      if (exact_hit(o, d, a, env.sphere(i), tmin, tmax, t, Env::kLibraryMath) && (best_i < 0 || t < best_t || (t == best_t && i < best_i)))
      {
        best_i = i;
        best_t = t;
        if (any_hit)
          return;
      }
    }
    // leave through the nearest wall
    bool const x_first = ax[0].t_next <= ax[1].t_next && ax[0].t_next <= ax[2].t_next;
    bool const y_first = !x_first && ax[1].t_next <= ax[2].t_next;
    float const t_exit = x_first ? ax[0].t_next : (y_first ? ax[1].t_next : ax[2].t_next);
    if (best_i >= 0 && best_t <= t_exit - slack)
      return;
    if (t_exit > t1 + slack)
      return;
    bool inside;
    if (x_first)
      inside = ax[0].advance((int)A.grid_dim[0], A.grid_cell);
    else if (y_first)
      inside = ax[1].advance((int)A.grid_dim[1], A.grid_cell);
    else
      inside = ax[2].advance((int)A.grid_dim[2], A.grid_cell);
    if (!inside)
      return;
  }
}

template <class Env>
RT_FN void resolve_primary(const TraceArgs & A, Env & env, uint32_t mask, uint32_t base, uint32_t count, f3 dir,
                           float a, float & best_t, int & best_i, LaneCounters & cnt)
{
  while (env.any_wave(mask != 0u))
  {
    if (mask != 0u)
    {
      int const i = (int)(base + pop_candidate(mask, count));
      float t;
      cnt.exact_tests += 1;
      if (exact_hit(A.cam_pos, dir, a, env.sphere(i), A.znear, A.zfar, t, Env::kLibraryMath) && (best_i < 0 || t < best_t))
      {
        best_i = i;
        best_t = t;
      }
    }
  }
}

// cm0, cm1: the column's candidate masks, used instead of a filter pass when A.column_masks is set.
template <bool kGrid, class Env>
RT_FN void trace_primary(const TraceArgs & A, Env & env, bool active, f3 dir, float a, uint32_t cm0, uint32_t cm1,
                         float & best_t, int & best_i, LaneCounters & cnt)
{
  if constexpr (kGrid)
  {
    grid_trace(A, env, active, A.cam_pos, dir, a, A.znear, A.zfar, false, best_t, best_i, cnt);
    return;
  }
  best_i = -1;
  best_t = 0.0f;
  if (A.column_masks != 0u)
  {
    // the column's candidates, sphere j in bit 63 - j (prepare_column): one loop over all of them
    RT_REGION(cnt, 2);
    unsigned long long mask = active ? (((unsigned long long)cm0 << 32) | (unsigned long long)cm1) : 0ull;
    while (env.any_wave(mask != 0ull))
    {
      RT_COUNT_TRIP(cnt, 11);
      if (mask != 0ull)
      {
        int const i = (int)pop_group(mask);
        float t;
        cnt.exact_tests += 1;
        if (exact_hit(A.cam_pos, dir, a, env.sphere(i), A.znear, A.zfar, t, Env::kLibraryMath) && (best_i < 0 || t < best_t))
        {
          best_i = i;
          best_t = t;
        }
      }
    }
    RT_REGION(cnt, 3);
    return;
  }
  // An environment whose waves move in step through the staged tiles (Env::kCompactsRays) first hands the rays of
  // the lanes that have one to its first lanes: in a round only the lanes without a hit trace a primary ray, about
  // half of them where paths are long, and a wave none of whose lanes holds a ray skips the arithmetic of the round
  // (it still takes part in staging the tiles).  Which lane computes a ray changes nothing about its result.
  bool mine = active;
  uint32_t home = 0;
  if constexpr (Env::kCompactsRays)
    home = env.gather_rays(mine, dir, a);
  for (uint32_t k = 0; k < env.tiles(); ++k)
  {
    uint32_t const tile_count = env.load_tile(k, true);
    uint32_t const tile_base = env.tile_base(k);
    const f4 * tab = env.tile_data(true);
    if (Env::kCompactsRays && !env.any_wave(mine))
      continue;
    for (uint32_t c0 = 0; c0 < tile_count; c0 += RT_CHUNK)
    {
      uint32_t const count = (tile_count - c0 < RT_CHUNK) ? (tile_count - c0) : RT_CHUNK;
      RT_REGION(cnt, 1);
      uint32_t mask = filter_chunk_1<Env::kPrimaryUnroll>(tab + c0, count, dir);
      if (!mine)
        mask = 0;
      RT_REGION(cnt, 2);
      resolve_primary(A, env, mask, tile_base + c0, count, dir, a, best_t, best_i, cnt);
      RT_REGION(cnt, 3);
    }
  }
  if constexpr (Env::kCompactsRays)
    env.return_hits(mine, home, active, best_t, best_i);
  if (active)
    cnt.filter_tests += 1u;
}

// rays_on: bit r set if ray r of this lane is to be traced.  Returns the shadow rays found occluded as bits
// 0..4; the nearest hit of the scattered ray (ray 5) goes to near_t / near_i.  `cone` holds dir[0..4].
template <bool kGrid, bool kOneLight, class Env>
RT_FN uint32_t trace_pass(const TraceArgs & A, Env & env, f3 o, const f3 (&dir)[RT_RAYS], ShadowCone cone, uint32_t rays_on, float & near_t, int & near_i, LaneCounters & cnt)
{
  if constexpr (kGrid)
  {
    uint32_t occluded = 0;
#pragma unroll
    for (int r = 0; r < RT_LIGHT_SAMPLES; ++r)
    {
      float t;
      int i;
      grid_trace(A, env, ((rays_on >> r) & 1u) != 0u, o, dir[r], dot3(dir[r], dir[r]), 0.001f, 1000.0f, true, t, i, cnt);
      occluded |= (i >= 0) ? (1u << r) : 0u;
    }
    grid_trace(A, env, ((rays_on >> (RT_RAYS - 1)) & 1u) != 0u, o, dir[RT_RAYS - 1], dot3(dir[RT_RAYS - 1], dir[RT_RAYS - 1]), 0.001f, A.zfar,
               false, near_t, near_i, cnt);
    return occluded;
  }
  uint32_t live = rays_on;  // rays still looking for a hit
  bool const library_math = Env::kLibraryMath;
  ShadeOrigin const F = shade_origin(o, A.centre, dir[RT_RAYS - 1]);
  cone.cos2_m = cone.cos_m * cone.cos_m;
  cone.oa = 0.0f;
  if constexpr (kOneLight)
  {
    // dot(O', axis) / cos, lowered by several times the rounding error its operations can have (three for the dot
    // product, the rounding of 1 / cos and the last multiplication: 6e-8 of the terms' sizes each)
    float const l1 = hw_fma(hw_abs(F.o.z), hw_abs(cone.axis.z),
                            hw_fma(hw_abs(F.o.y), hw_abs(cone.axis.y), hw_abs(F.o.x) * hw_abs(cone.axis.x)));
    cone.oa = hw_fma(-8e-7f, l1, fdot3(F.o, cone.axis)) * A.cone_inv_cos;
  }
  near_i = -1;
  near_t = 0.0f;
  constexpr uint32_t kShadowRays = (1u << RT_LIGHT_SAMPLES) - 1u;
  constexpr uint32_t kGroup = 2u * RT_CHUNK;
  for (uint32_t k = 0; k < env.tiles(); ++k)
  {
    uint32_t const tile_count = env.load_tile(k, false);
    uint32_t const tile_base = env.tile_base(k);
    const f4 * sph = env.tile_data(false);
    const float * aux = env.tile_aux();
    // The filter runs chunk by chunk (one 32-bit mask per test), the candidates are resolved per group of two
    // chunks: the loops below go round as often as the lane of the wave with the most candidates needs, so one
    // loop over 64 spheres takes fewer trips than two loops over 32.
    for (uint32_t c0 = 0; c0 < tile_count; c0 += kGroup)
    {
      uint32_t const count_lo = (tile_count - c0 < RT_CHUNK) ? (tile_count - c0) : RT_CHUNK;
      uint32_t const rest = tile_count - c0 - count_lo;
      uint32_t const count_hi = (rest < RT_CHUNK) ? rest : RT_CHUNK;
      uint32_t cone_lo, ray_lo, cone_hi = 0u, ray_hi = 0u;
      RT_REGION(cnt, 6);
      filter_chunk_2<Env::kFilterUnroll, kOneLight>(sph + c0, aux + c0, count_lo, F, cone, cone_lo, ray_lo);
      if (count_hi != 0u)
        filter_chunk_2<Env::kFilterUnroll, kOneLight>(sph + c0 + RT_CHUNK, aux + c0 + RT_CHUNK, count_hi, F, cone, cone_hi,
                                                      ray_hi);
      RT_REGION(cnt, 7);
      // sphere j of the group in bit 63 - j: the lowest index is the highest bit
      unsigned long long cone_mask = group_mask(cone_lo, count_lo, cone_hi, count_hi);
      unsigned long long ray_mask = group_mask(ray_lo, count_lo, ray_hi, count_hi);
      cone_mask = ((live & kShadowRays) != 0u) ? cone_mask : 0ull;
      ray_mask = ((live >> (RT_RAYS - 1)) & 1u) ? ray_mask : 0ull;
      if (!env.any_wave((cone_mask | ray_mask) != 0ull))
        continue;
      uint32_t const base = tile_base + c0;
      // Shadow rays, sphere by sphere: a lane takes the next sphere its cone may touch and runs Sphere::TraceNearest
      // (ray_tracing/sphere.cpp:63-81) for its five rays at once, without branches.  d = o - centre and
      // c = dot(d, d) - r^2 do not depend on the ray, and the loop goes round as often as the lane with the most
      // such spheres needs (seldom more than once) instead of once per pair of rays.  Existence of a hit in
      // [0.001, 1000] is all TraceLight uses, so a lane whose five rays are all blocked drops its other spheres.
      while (env.any_wave(cone_mask != 0ull))
      {
        RT_COUNT_TRIP(cnt, 12);
        bool const h = cone_mask != 0ull;
        unsigned long long const m = h ? cone_mask : 0x8000000000000000ull;
        uint32_t const j = (uint32_t)hw_clz64(m);
        f4 const s = env.sphere((int)(base + j));
        f3 const d = o - mk3(s.x, s.y, s.z);
        float const c = dot3(d, d) - s.w;
        uint32_t const looking = h ? (live & kShadowRays) : 0u;
#pragma unroll
        for (int r = 0; r < RT_LIGHT_SAMPLES; ++r)
        {
          float dx = dir[r].x;
#if defined(__HIP_DEVICE_COMPILE__)
          // keeps dot(D, D), 2a and 4a of the five rays from being computed ahead of the loop and held in a dozen
          // registers across it (and across every group of a tiled scene) for five multiplications per sphere
          asm volatile("" : "+v"(dx));
#endif
          f3 const dr = mk3(dx, dir[r].y, dir[r].z);
          float const a = dot3(dr, dr);
          float const b = 2.0f * dot3(d, dr);
          float const disc = b * b - 4 * a * c;
          float const t = near_root(b, disc, 2.0f * a, library_math);
          bool const hit = !(disc < 0.0f) && !(t < 0.001f || t > 1000.0f) && ((looking >> r) & 1u) != 0u;
          live = hit ? (live & ~(1u << r)) : live;
        }
        cnt.cone_tests += h ? 1u : 0u;
        cnt.exact_tests += (uint32_t)__builtin_popcount(looking);
        cone_mask = (h && (live & kShadowRays) != 0u) ? (m & ~(0x8000000000000000ull >> j)) : 0ull;
      }
      // The scattered ray: nearest hit, lowest index first so that the first of equal distances wins; secondary
      // hits start at 0.001 (demos/demo_frame_cpu.cpp:53).
      while (env.any_wave(ray_mask != 0ull))
      {
        RT_COUNT_TRIP(cnt, 13);
        bool const h1 = ray_mask != 0ull;
        unsigned long long m1 = h1 ? ray_mask : 0x8000000000000000ull;
        int const i1 = (int)(base + pop_group(m1));
        float t1;
        bool const hit1 = exact_hit_flat(o, dir[RT_RAYS - 1], dot3(dir[RT_RAYS - 1], dir[RT_RAYS - 1]), env.sphere(i1),
                                         0.001f, A.zfar, t1, library_math) && h1;
        cnt.exact_tests += h1 ? 1u : 0u;
        bool const nearer = hit1 && (near_i < 0 || t1 < near_t);
        near_i = nearer ? i1 : near_i;
        near_t = nearer ? t1 : near_t;
        ray_mask = h1 ? m1 : 0ull;
      }
      RT_REGION(cnt, 8);
    }
  }
  // tests run per sphere of the scene: the cone (once for the five shadow rays) and the scattered ray
  cnt.filter_tests += ((rays_on & kShadowRays) != 0u ? 1u : 0u) + ((rays_on >> (RT_RAYS - 1)) & 1u);
  return (rays_on & ~live) & kShadowRays;
}

// kOneLight: the scene has exactly one light, so a hit is shaded within one trip (one pass) and nothing of
// the shading state is carried from trip to trip -- some twenty registers fewer across the loop.
template <bool kForward, bool kGrid, bool kOneLight, class Env>
RT_FN void trace_loop(const TraceArgs & A, Env & env, f4 * stack, uint32_t stride, LaneCounters & cnt)
{
  uint32_t const n_lights = kOneLight ? 1u : A.n_lights;
  Lane L;
  L.cm0 = L.cm1 = 0u;
  L.acc_l = mk3(0.0f, 0.0f, 0.0f);
  L.acc_e = 1.0f;
  L.alive = true;
  L.need_item = true;
  L.has_hit = false;
  L.shading = false;
  L.slot = 0;
  L.pixel = 0;
  L.sx = 0;
  L.sy = 0;
  L.sy_end = 0;
  L.depth = 0;
  L.sample = 0;
  L.li = 0;
  L.hidx = 0;
  L.nx_i = -1;
  L.nx_t = 0.0f;
  L.e = 0.0f;
  L.m_refraction = 0.0f;
  L.sum = L.hp = L.hn = L.hd = L.spec = L.diffuse = L.m_albedo = L.sdir = mk3(0.0f, 0.0f, 0.0f);
  L.sdx = L.dyp = 0.0f;
  bool const deterministic = (A.flags & RT_FLAG_DETERMINISTIC) != 0u;
  bool const library_math = Env::kLibraryMath;

  for (;;)
  {
    cnt.trips += 1;
    // ================= stage P: primary rays for lanes with nothing to shade =================
    for (;;)
    {
      cnt.rounds += 1;
      bool const need = L.alive && !L.has_hit && !L.shading;
      // Column (or shared sample) complete: its sum leaves the lane.
      bool const finishing = need && !L.need_item && L.sy >= L.sy_end;
      env.finish_columns(L, finishing);
      if (finishing)
        L.need_item = true;
      bool const ask = need && L.need_item;
      bool waiting = false;  // the lane asked and the environment has to finish older windows first
      if (env.any(ask))
      {
        ItemInfo info;
        env.fetch_item(L, ask, info);
        cnt.prepared += info.filter_tests;
        if (ask)
        {
          if (info.pos == RT_INVALID)
          {
            if (info.stalled)
              waiting = true;
            else
              L.alive = false;
          }
          else
          {
            L.slot = info.slot;
            if (info.prepared)
            {
              L.pixel = info.pixel;
              L.sx = info.sx;
              L.sy = info.sy0;
              L.sy_end = A.s_row;
              L.sdx = info.sdx;
              L.dyp = info.dyp;
              L.cm0 = info.cm0;
              L.cm1 = info.cm1;
              L.sum = mk3(0.0f, 0.0f, 0.0f);
            }
            else
              begin_item(A, L, info.pos);
            L.need_item = false;
          }
        }
      }
      // The queue is used up for this wave: idle lanes take samples off the busy lanes' columns.
      if (env.endgame())
        env.share(L);
      // (a lane that asked has its item, or is idle now; a column of which the slice owns no sample is completed
      // in the next round)
      bool const active = L.alive && !L.has_hit && !L.shading && L.sy < L.sy_end;
      RT_REGION(cnt, 0);
      if (env.any(active))
      {
        uint32_t const s = env.column_sx(L) * A.s_row + L.sy;
        f3 d = mk3(0.0f, 0.0f, 1.0f);
        if (active)
        {
          d = primary_dir(A, L.sdx, L.dyp, L.sy, library_math);
          L.sy += A.slice_count;
        }
        float const a = dot3(d, d);
        float t;
        int hi;
        trace_primary<kGrid>(A, env, active, d, a, L.cm0, L.cm1, t, hi, cnt);
        if (active)
        {
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
            f4 const sp = env.sphere(hi);
            L.hp = A.cam_pos + d * t;  // Ray::PointAt
            L.hn = normalize3(L.hp - mk3(sp.x, sp.y, sp.z), library_math);
            L.hd = d;
            L.hidx = hi;
            L.has_hit = true;
            L.depth = 0;
            L.sample = s;
          }
        }
      }
      RT_REGION(cnt, 4);
      // (a lane that waits for a window is not looking for a hit: it can do nothing until other lanes' columns end)
      bool const still = L.alive && !L.has_hit && !L.shading && !waiting;
      int const n_still = env.count(still);
      if (n_still == 0)
        break;
      bool const someone_ready = env.any(L.alive && (L.has_hit || L.shading));
      if (someone_ready && n_still <= RT_P_THRESHOLD)
        break;
    }
    if (!env.any(L.alive))
      break;

    // ================= stage S: shade the hits =================
    f3 dir[RT_RAYS];
    uint32_t rays_on = 0;
    // Philox block of the first light's shadow jitter, drawn with the scatter block (the first light's pass is
    // always in the trip that shades the new hit)
    u4 shadow_words;
    shadow_words.x = shadow_words.y = shadow_words.z = shadow_words.w = 0u;

    if (L.alive && L.has_hit)
    {
      // A new hit: Material::Scatter first (demos/demo_frame_cpu.cpp:38).
      uint32_t const mi = env.sphere_material(L.hidx);
      f4 const m0 = env.material(mi, 0u);
      f4 const m1 = env.material(mi, 1u);
      f4 const m2 = env.material(mi, 2u);
      MaterialView mv;
      mv.type = f2u(m1.y);
      mv.roughness = m0.w;
      mv.refraction = m1.x;
      mv.f2 = m1.z;
      mv.inv_refraction = m1.w;
      mv.r2 = m2.x;
      mv.four_r2 = m2.y;
      mv.e_metal = m2.z;
      L.m_albedo = mk3(m0.x, m0.y, m0.z);
      L.m_refraction = mv.refraction;

      f3 ball = mk3(0.0f, 0.0f, 0.0f);
      float xi = 0.5f;
      if (!deterministic)
      {
        // The first light's shadow block is drawn here too: the two Philox chains are independent and
        // interleave, where each alone is a long run of dependent instructions.
        u4 const rw = philox4x32_10(L.depth, 0u, L.sample, 0u, L.pixel, A.seed);
        shadow_words = philox4x32_10(L.depth, 1u, L.sample, 0u, L.pixel, A.seed);
        ball = ball3(1.0f, uniform24(rw.x), uniform24(rw.y), uniform24(rw.z));
        xi = uniform24(rw.w);
      }
      scatter(mv, L.hd, L.hn, ball, xi, L.sdir, L.e, library_math);
      L.has_hit = false;
      if (fabsf(L.e) < RT_EPS)
      {
        // Absorbed: the entry becomes the environment colour and the path ends
        // (demos/demo_frame_cpu.cpp:39-45).
        path_push<kForward>(L, stack, stride, A.background, 1.0f);
        L.sum = L.sum + path_colour<kForward>(L, stack, stride);
      }
      else
      {
        L.spec = mk3(0.0f, 0.0f, 0.0f);
        for (uint32_t li = 0; li < n_lights; ++li)
        {
          f4 const l0 = env.light(li, 0u);
          f4 const l1 = env.light(li, 1u);
          L.spec = L.spec + specular(mk3(l0.x, l0.y, l0.z), mk3(l1.x, l1.y, l1.z), L.m_albedo, mv, L.hd, L.hn, library_math);
        }
        L.diffuse = mk3(0.0f, 0.0f, 0.0f);
        L.li = 0;
        L.nx_i = -1;
        L.nx_t = 0.0f;
        L.shading = true;
      }
    }

    bool const in_pass = L.alive && L.shading;
    RT_REGION(cnt, 5);
    // The ray set-up below runs for every lane, in a pass or not: lanes that are not compute throw-away rays
    // from their stale (always initialised) state, and rays_on keeps those rays out of the trace.  That is
    // cheaper than a divergent region whose results must be merged with defaults.
    bool const lit = in_pass && L.li < n_lights;
    ShadowCone cone;
    {
      uint32_t const li_c = (L.li < n_lights) ? L.li : 0u;
      f3 to_light = mk3(0.0f, 1.0f, 0.0f);
      cone.axis = to_light;
      cone.cos_m = 0.0f;
      cone.sin_m = 3.0e38f;
      if (n_lights != 0u)
      {
        f4 const l0 = env.light(li_c, 0u);
        to_light = mk3(-l0.x, -l0.y, -l0.z);
        if constexpr (kOneLight)
        {
          cone.axis = A.cone_axis;
          cone.cos_m = A.cone_cos;
          cone.sin_m = A.cone_sin;
        }
        else
        {
          f4 const l1 = env.light(li_c, 1u);
          f4 const l2 = env.light(li_c, 2u);
          cone.axis = mk3(l2.x, l2.y, l2.z);
          cone.cos_m = l0.w;
          cone.sin_m = l1.w;
        }
      }
      if (deterministic)
      {
        // No jitter: the five shadow rays are one ray, normalised exactly as the reference does
        // (ray_tracing/directional_light.cpp:20), so that a shadow edge never falls between two roundings.
        f3 const rd = normalize3(to_light, library_math);
#pragma unroll
        for (int j = 0; j < RT_LIGHT_SAMPLES; ++j)
          dir[j] = rd;
      }
      else
      {
        // 15 cells of 8 bits from one Philox block: bytes b[0..15], ball j takes b[3j..3j+2].
        u4 p0 = shadow_words;
        if (lit && L.li != 0u)
          p0 = philox4x32_10(L.depth, 1u + L.li, L.sample, 0u, L.pixel, A.seed);
        uint32_t const wd[4] = {p0.x, p0.y, p0.z, p0.w};
#pragma unroll
        for (int j = 0; j < RT_LIGHT_SAMPLES; ++j)
        {
          uint32_t k[3];
#pragma unroll
          for (int q = 0; q < 3; ++q)
          {
            int const h = 3 * j + q;
            k[q] = (wd[h >> 2] >> (8 * (h & 3))) & 0xFFu;
          }
          // jittered directions need no exact rounding: fused multiply-adds and the hardware rsqrt
          f3 const v = to_light + jitter_ball(env.jitter_table(), k[0], k[1], k[2]);
          float const inv = hw_rsq(fdot3(v, v));
          f3 const rd = v * inv;
          dir[j] = rd;
        }
      }
      rays_on = lit ? ((1u << RT_LIGHT_SAMPLES) - 1u) : 0u;
      dir[RT_RAYS - 1] = L.sdir;
      rays_on |= (in_pass && L.li == 0u) ? (1u << (RT_RAYS - 1)) : 0u;
    }

    float near_t;
    int near_i;
    RT_REGION(cnt, 9);
    uint32_t const occluded = trace_pass<kGrid, kOneLight>(A, env, L.hp, dir, cone, rays_on, near_t, near_i, cnt);

    if (in_pass)
    {
      if (L.li < n_lights)
      {
        // DirectionalLight::TraceLight, ray_tracing/directional_light.cpp:15-35.
        f4 const l1 = env.light(L.li, 1u);
        f3 const lcol = mk3(l1.x, l1.y, l1.z);
        // An occluded sample is tinted by the shaded surface's own material.
        f3 const tint = mix3(L.m_albedo, lcol, 0.75f) * (L.m_refraction * 0.75f);
        // sum of (blocked ? tint : colour) * w_j, gathered as two weights (a colour, not geometry: the last
        // bits may differ from adding sample by sample)
        float w_blocked = 0.0f, w_open = 0.0f;
        // max(0, dot(direction, normal)), directional_light.cpp:24,31 (taken after the trace: five registers fewer
        // across it)
        float w[RT_LIGHT_SAMPLES];
        if (deterministic)
        {
          float const ww = maxf(0.0f, dot3(dir[0], L.hn));  // the five rays are one ray
#pragma unroll
          for (int j = 0; j < RT_LIGHT_SAMPLES; ++j)
            w[j] = ww;
        }
        else
        {
#pragma unroll
          for (int j = 0; j < RT_LIGHT_SAMPLES; ++j)
            w[j] = hw_max(0.0f, fdot3(dir[j], L.hn));
        }
#pragma unroll
        for (int j = 0; j < RT_LIGHT_SAMPLES; ++j)
        {
          bool const blocked = ((occluded >> j) & 1u) != 0u;
          float const wj = w[j];
          w_blocked += blocked ? wj : 0.0f;
          w_open += blocked ? 0.0f : wj;
        }
        cnt.hits += (uint32_t)__builtin_popcount(occluded);
        float const fifth = 1.0f / (float)RT_LIGHT_SAMPLES;
        L.diffuse = L.diffuse + (tint * (w_blocked * fifth) + lcol * (w_open * fifth));
        cnt.queries += RT_LIGHT_SAMPLES;
      }
      if (L.li == 0)
      {
        L.nx_i = near_i;
        L.nx_t = near_t;
      }
      L.li += 1;
      if (L.li >= n_lights)
      {
        // l = attenuation + diffuse + specular (demos/demo_frame_cpu.cpp:66); every material's
        // attenuation is its albedo (ray_tracing/default_materials.cpp:14,26,40).
        path_push<kForward>(L, stack, stride, (L.m_albedo + L.diffuse) + L.spec, L.e);
        cnt.shaded += 1;
        L.shading = false;
        bool ended = true;
        if (L.depth < A.max_depth)
        {
          // The reference traces the scattered ray only while the depth limit allows another entry.
          cnt.queries += 1;
          if (L.nx_i >= 0)
          {
            cnt.hits += 1;
            f4 const sp = env.sphere(L.nx_i);
            f3 const p = L.hp + L.sdir * L.nx_t;
            L.hn = normalize3(p - mk3(sp.x, sp.y, sp.z), library_math);
            L.hp = p;
            L.hd = L.sdir;
            L.hidx = L.nx_i;
            L.has_hit = true;
            ended = false;
          }
        }
        if (ended)
          L.sum = L.sum + path_colour<kForward>(L, stack, stride);
      }
    }
    RT_REGION(cnt, 10);
    if constexpr (kOneLight)
    {
      // the pass above finished whatever was being shaded: the state it used is dead, say so
      L.spec = L.diffuse = L.sdir = L.m_albedo = mk3(0.0f, 0.0f, 0.0f);
      L.e = 0.0f;
      L.m_refraction = 0.0f;
      L.nx_t = 0.0f;
      L.nx_i = -1;
      L.li = 0;
      L.shading = false;
    }
  }
}
}  // namespace rtc
