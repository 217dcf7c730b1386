/*
 * rt_oracle.h -- CPU restatement of the rokuz/rt trace loop.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is the checker for the HIP backend: the parity tests, the smoke check and the
 * cpu_baseline leg of the benchmark load it; nothing that ships (librt_hip, the C++ host, the CLI)
 * may include, link or call it.
 *
 * PARITY UNPINNED: the reference holds no tests, golden images or known-answer vectors for this
 * path, its vector maths lives in an un-vendored glm submodule (https://github.com/g-truc/glm,
 * commit not recorded; 0.9.9-era), and it cannot be compiled in this environment.  The functions
 * below restate the reference's own source line by line (each cites the lines it follows) with
 * glm's published GLSL-equivalent definitions for dot/normalize/cross/reflect/refract/mix/clamp.
 * The only intentional departure is the random source: the reference draws glm::ballRand /
 * glm::linearRand from the C library generator; here every draw comes from Philox4x32-10 keyed on
 * (pixel, seed) with counter (bounce, block, sample), so the CPU and GPU see identical streams.
 *
 * All arithmetic is IEEE binary32 with float math functions (sqrtf, powf, expf, tanf), no FMA
 * contraction (build with -ffp-contract=off).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ray_tracing/types.hpp:7-20 */
#define ORC_MATERIAL_MATTE 0u
#define ORC_MATERIAL_METAL 1u
#define ORC_MATERIAL_GLASS 2u
#define ORC_LIGHT_DIRECTIONAL 0u
#define ORC_EPS 1e-5f

/* demos/demo_frame_cpu.cpp:10, ray_tracing/directional_light.cpp:13 */
#define ORC_MAX_DEPTH 10u
#define ORC_LIGHT_SAMPLES 5u

/* flags */
#define ORC_FLAG_DETERMINISTIC 1u /* ballRand == 0, glass branch draw == 0.5 */

/* What DemoFrameCUDA::FindMaterial flattens a Material into (rt_cuda/demo_frame_cuda.cpp:87-111):
 * the values returned by GetType/GetAlbedo/GetRoughness/GetRefraction. */
typedef struct
{
  uint32_t type;
  float albedo[3];
  float roughness;
  float refraction;
} orc_material;

typedef struct
{
  uint32_t type;
  float direction[3];
  float color[3];
} orc_light;

/* ray_tracing/frame.hpp:66-72 */
typedef struct
{
  float position[3];
  float direction[3]; /* already normalised (Frame::SetCameraDirection, frame.cpp:79-82) */
  float fov;
  float znear;
  float zfar;
  float background[3];
} orc_camera;

typedef struct
{
  const float * spheres;           /* sphere_count x {cx, cy, cz, radius} */
  const uint32_t * material_index; /* sphere_count */
  uint32_t sphere_count;
  const orc_material * materials;
  uint32_t material_count;
  const orc_light * lights;
  uint32_t light_count;
} orc_scene;

typedef struct
{
  uint32_t width;
  uint32_t height;
  uint32_t samples_in_row; /* S: S*S samples per pixel */
  uint32_t flags;
  uint32_t seed;
  uint32_t max_depth;   /* 0 -> ORC_MAX_DEPTH; 1 = primary hit + direct light only */
  uint32_t slice_index; /* this renderer owns sample (sx, sy) of pixel p iff ((p * S + sx) * S + sy) % slice_count == slice_index */
  uint32_t slice_count; /* 0 or 1 -> everything */
} orc_params;

typedef struct
{
  uint64_t samples;        /* camera samples traced */
  uint64_t ray_queries;    /* nearest-hit + shadow queries, each against all spheres */
  uint64_t hit_queries;    /* queries that found a hit */
  uint64_t shaded_bounces; /* hits that ran direct light + specular */
  uint64_t sphere_tests;   /* ray_queries * sphere_count */
} orc_stats;

/* ---- building blocks (exported so the tests can pin each one) ---- */

/* Philox4x32-10 (Salmon et al., SC'11; Random123 reference constants). */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* uint32 -> [0,1): top 24 bits * 2^-24. */
float orc_uniform(uint32_t bits);

/* Uniform point in a ball of radius R from three uniforms (distribution of glm::ballRand). */
void orc_ball(float radius, float u1, float u2, float u3, float out[3]);

/* Frame::ForEachRayImpl, ray_tracing/frame.cpp:29-62: direction of sub-sample (sx, sy) of pixel (x, y). */
void orc_primary_direction(const orc_camera * cam, uint32_t width, uint32_t height, uint32_t samples_in_row,
                           uint32_t x, uint32_t y, uint32_t sx, uint32_t sy, float out_dir[3]);

/* Sphere::TraceNearest, ray_tracing/sphere.cpp:63-81.  Returns 1 on hit. */
int orc_sphere_trace_nearest(const float origin[3], const float dir[3], const float center[3], float radius,
                             float tmin, float tmax, float * out_t, float out_pos[3], float out_normal[3]);

/* Hit, ray_tracing/hitable_object.hpp:15-30, with the sphere's index in place of its material pointer. */
typedef struct
{
  float t;
  float position[3];
  float normal[3];
  uint32_t sphere;
} orc_hit;

/* Sphere::Trace, ray_tracing/sphere.cpp:11-61: all intersections in [tmin, tmax], nearest first.  Returns 0..2. */
int orc_sphere_trace(const float origin[3], const float dir[3], const float center[3], float radius, float tmin,
                     float tmax, orc_hit out[2]);

/* TraceInHitableCollection, ray_tracing/hitable_object.hpp:53-69: all hits of all spheres sorted by t.  Returns
 * the total number of hits; the nearest min(total, capacity) are written to out. */
uint32_t orc_trace_all(const orc_scene * scene, const float origin[3], const float dir[3], float tmin, float tmax,
                       orc_hit * out, uint32_t capacity);

/* TraceNearestInHitableCollection, ray_tracing/hitable_object.hpp:71-87.  Returns sphere index or -1. */
int orc_trace_nearest(const orc_scene * scene, const float origin[3], const float dir[3], float tmin, float tmax,
                      float * out_t, float out_pos[3], float out_normal[3]);

/* Matte/Metal/Glass::Scatter, ray_tracing/default_materials.cpp:11-67, with the random draws injected. */
void orc_scatter(const orc_material * mat, const float ray_dir[3], const float hit_normal[3], const float ball[3],
                 float xi, float out_attenuation[3], float out_dir[3], float * out_emissivity);

/* DirectionalLight::TraceLight, ray_tracing/directional_light.cpp:15-35; jitter = 5 ballRand(0.25) vectors. */
void orc_trace_light(const orc_scene * scene, const orc_light * light, const orc_material * mat,
                     const float hit_pos[3], const float hit_normal[3], const float jitter[ORC_LIGHT_SAMPLES][3],
                     float out_color[3], orc_stats * stats);

/* DirectionalLight::GetSpecular, ray_tracing/directional_light.cpp:37-60. */
void orc_get_specular(const orc_light * light, const orc_material * mat, const float ray_dir[3],
                      const float hit_normal[3], float out_color[3]);

/* Reverse accumulation, demos/demo_frame_cpu.cpp:60-75; l = attenuation + diffuse + specular per hit. */
void orc_accumulate(uint32_t count, const float (*l)[3], const float * emissivity, float out_color[3]);

/* DemoFrameCPU::RayTrace, demos/demo_frame_cpu.cpp:22-76, for one camera sample. */
void orc_ray_trace(const orc_scene * scene, const orc_camera * cam, const orc_params * params, uint32_t pixel,
                   uint32_t sample, const float dir[3], float out_color[3], orc_stats * stats);

/* Frame::ForEachRayImpl on MultiThreadedFrame's row interleave (ray_tracing/multithreaded_frame.cpp:33-59),
 * run to true completion of every row.  rgb = width*height*3 floats, row 0 on top.  Returns 0 on success. */
int orc_render(const orc_scene * scene, const orc_camera * cam, const orc_params * params, uint32_t threads,
               float * rgb, orc_stats * stats);

/* The same for the rows [row_begin, row_end) of the frame only (the range of the row loop, ray_tracing/frame.cpp:44):
 * rgb is still the whole frame, of which only those rows are written; stats count those rows. */
int orc_render_rows(const orc_scene * scene, const orc_camera * cam, const orc_params * params, uint32_t threads,
                    uint32_t row_begin, uint32_t row_end, float * rgb, orc_stats * stats);

#ifdef __cplusplus
}
#endif
#endif
