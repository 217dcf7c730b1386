/*
 * rt_hip.h -- C ABI of the MI355X (gfx950) ray-tracing backend for rokuz/rt.
 *
 * This is the layer a ray_tracing::Frame subclass binds to, in place of the reference's
 * namespace ray_tracing_cuda free functions (rt_cuda/rt_cuda.h:16-32).  Plain C: no HIP types, no
 * C++ types, no exceptions; every call returns an int status (RT_HIP_OK == 0) and leaves a message
 * for rt_hip_last_error() on failure.  Nothing is printed.
 *
 * Conventions
 *   - Host images are float[height * width * 3], RGB interleaved, row 0 on top, linear radiance
 *     already divided by samples_in_row^2 -- the memory layout of the reference's ColorBuffer
 *     (ray_tracing/frame.hpp:21) as DemoFrameCUDA hands it over (rt_cuda/demo_frame_cuda.cpp:39,146).
 *   - Device sum buffers are float[3 * height * width]: three planes (R, G, B) of un-normalised
 *     per-pixel sums; rt_hip_resolve_device turns them into the interleaved image.
 *   - The unit of work is one sub-sample column: the samples_in_row samples (sx, 0..S-1) of one pixel
 *     (the inner loop of ray_tracing/frame.cpp:59-66).  A trace can be restricted to a slice of the
 *     sample space: the samples of the frame, numbered ((p * S + sx) * S + sy) in the order the
 *     reference's loops visit them, are dealt out to the slices in turn (sample number % slice_count;
 *     slice_count < 65536), so every slice takes every slice_count-th sample of each column.  The sums
 *     of all slices add up to the whole frame, which is how one frame is sharded over several GPUs
 *     (one process per GPU; the caller reduces the planes).  RT_HIP_FLAG_PIXEL_RUNS deals out runs of
 *     pixels instead.
 *
 * Threads
 *   - The library serves one device per process (the one rt_hip_init selected).  rt_hip_init for another device
 *     releases what the library owns on the old one first; scenes and jobs made on the old device must have been
 *     destroyed / finished by then.
 *   - Scenes are immutable after rt_hip_scene_create and may be traced from several threads and in several jobs at
 *     once (every job and every stream workspace has its own camera table).
 *   - A job belongs to one thread at a time: poll / snapshot / cancel / finish of one job must not overlap.
 *     Different jobs may be started and finished from different threads; they run on the library's trace stream
 *     in the order they were started.
 *   - rt_hip_trace_device may be called from several threads (one caller stream each; a call that has to wait for
 *     the device -- its stream's workspace grows, or changes hands -- holds up calls on that stream only).
 *   - rt_hip_trace_device_static shares one scratch array per GPU: calls that overlap on one GPU race.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_HIP_OK 0
#define RT_HIP_ERR_INVALID 1    /* bad argument */
#define RT_HIP_ERR_DEVICE 2     /* a HIP call failed; see rt_hip_last_error() */
#define RT_HIP_ERR_NOT_READY 3  /* rt_hip_job_poll: the job is still running */
#define RT_HIP_ERR_CANCELLED 4  /* the job was cancelled before it finished */

/* Type tags, ray_tracing/types.hpp:12-20. */
#define RT_HIP_MATERIAL_MATTE 0u
#define RT_HIP_MATERIAL_METAL 1u
#define RT_HIP_MATERIAL_GLASS 2u
#define RT_HIP_LIGHT_DIRECTIONAL 0u

/* Trace flags. */
#define RT_HIP_FLAG_DETERMINISTIC 1u /* no stochastic scattering: ballRand == 0, glass branch draw == 0.5 */
#define RT_HIP_FLAG_GRID 2u          /* rays walk a uniform grid over the spheres instead of testing every sphere
                                        (scenes of 64 spheres or more; ignored for smaller ones).  Same image: the
                                        grid only selects which spheres the exact intersection is run on. */
#define RT_HIP_FLAG_PIPELINED 4u     /* the frame is one of a stream of frames traced on alternating streams
                                        (rt_hip_trace_device), whose first waves take over the compute units this
                                        frame's last waves leave.  A hint: the launch then keeps its last work
                                        reservations at two queue windows instead of one, which serves the stream's
                                        throughput and costs a frame traced on its own a little.  Same image. */

#define RT_HIP_FLAG_LEAVE_ROOM 8u    /* the launch leaves 64 of the chip's block slots (of 1024) free, so that another
                                        kernel -- the RCCL reduce of the previous frame on a multi-GPU driver -- finds
                                        compute units while this frame is traced.  Same image. */

#define RT_HIP_FLAG_PIXEL_RUNS 16u   /* with slice_count > 1: the slices divide the frame by pixels instead of by
                                        samples.  The frame's pixels, in row-major order, form runs of 64; run q belongs
                                        to slice q % slice_count, which traces every sample of its pixels and leaves
                                        the sums of the others zero.  The slices' planes still add up to the whole
                                        frame -- bit for bit, since every pixel's sum comes from one slice -- and a
                                        slice's work is its share of the columns, not every column for a share of its
                                        samples (a 1/8 slice of the 1080p 16x16 frame: 2.4 ms instead of 3.2).  Same
                                        image. */

#define RT_HIP_FLAG_KERNEL_SUMS 32u  /* the trace kernel adds up the pixels itself: every wave keeps the column sums of
                                        the queue windows it works on in LDS and writes 12 bytes per pixel when a window
                                        is done, instead of 12 bytes per sub-sample column for a separate accumulate
                                        pass.  The same additions in the same order: same image, bit for bit.  Needs
                                        samples_in_row in {1, 2, 4, 8, 16, 32, 64} and a scene whose paths need no stack
                                        (ignored otherwise).  The kernel runs 3-5 % longer than with the pass, so this
                                        is for callers short of memory or bandwidth: the workspace shrinks from 12 bytes
                                        per column of the frame to the room of sample sharing (at most 54 MB), and
                                        rt_hip_trace_device_static loses its frame-size limit. */

/* Replaces CudaMaterial (rt_cuda/types_cuda.h): what Material::GetType/GetAlbedo/GetRoughness/
 * GetRefraction return (ray_tracing/material.hpp:20-26). */
typedef struct
{
  uint32_t type;
  float albedo[3];
  float roughness;
  float refraction;
} rt_hip_material;

/* Replaces CudaLight (rt_cuda/types_cuda.h): DirectionalLight::GetDirection/GetColor
 * (ray_tracing/directional_light.hpp:20-21). */
typedef struct
{
  uint32_t type;
  float direction[3];
  float color[3];
} rt_hip_light;

/* The camera/background arguments of ray_tracing_cuda::RayTrace (rt_cuda/rt_cuda.h:21-22);
 * values as Frame holds them (ray_tracing/frame.hpp:66-72).  direction must be normalised. */
typedef struct
{
  float position[3];
  float direction[3];
  float fov;
  float znear;
  float zfar;
  float background[3];
} rt_hip_camera;

typedef struct
{
  uint32_t width;
  uint32_t height;
  uint32_t samples_in_row; /* S: S*S samples per pixel, regular grid (ray_tracing/frame.cpp:56-66) */
  uint32_t flags;          /* RT_HIP_FLAG_* */
  uint32_t seed;           /* Philox4x32-10 key word: key = (pixel, seed), counter = (bounce, block, sample, 0) */
  uint32_t max_depth;      /* 0 -> 10 (demos/demo_frame_cpu.cpp:10); 1 = primary hit + direct light only */
  uint32_t slice_index;    /* see "Conventions" */
  uint32_t slice_count;    /* 0 or 1 -> the whole sample space */
} rt_hip_trace_params;

/* Work counters of one trace, counted on the device. */
typedef struct
{
  uint64_t samples;        /* camera samples traced */
  uint64_t ray_queries;    /* nearest-hit + shadow queries the algorithm asks for, each against all spheres */
  uint64_t hit_queries;    /* queries that found a hit */
  uint64_t shaded_bounces; /* hits that ran direct light + specular */
  uint64_t sphere_tests;   /* filter tests executed against every sphere of the scene: per shaded bounce the cone that
                              holds the light's five shadow rays and the scattered ray; primary rays once per sample,
                              or once per column for scenes of at most 64 spheres; 0 on the grid path */
  uint64_t exact_tests;    /* exact intersections run: the candidates the filter (or the grid walk) let through */
  uint64_t lane_trips;     /* trips round the kernel's main loop, summed over lanes (busy or idle) */
  uint64_t lane_rounds;    /* primary-ray rounds inside those trips, summed over lanes (busy or idle) */
  uint64_t cone_tests;     /* spheres a shadow cone touched; each was then intersected with the shadow rays still unblocked */
  float kernel_ms;         /* device time of the trace kernel, HIP events on the launch stream; 0 if not measured */
  uint32_t reserved;
} rt_hip_stats;

typedef struct rt_hip_scene rt_hip_scene;
typedef struct rt_hip_job rt_hip_job;

/* Replaces ray_tracing_cuda::Initialize (rt_cuda/rt_cuda.h:16, rt_cuda/rt_cuda.cu:71-96): selects
 * the device for this process (one process per GPU).  device < 0 keeps the current device. */
int rt_hip_init(int device);

/* Releases everything the library still owns on the device. */
void rt_hip_shutdown(void);

/* Message of the last failing call on this thread ("" if none). */
const char * rt_hip_last_error(void);

/* Device facts used for the FP32 vector roofline: peak = cus * 4 SIMDs * 32 lanes * 2 FLOP * clock. */
int rt_hip_device_info(int * compute_units, int * clock_khz, char * name, size_t name_len);

/* Uploads a flattened scene once; replaces the per-trace cudaMalloc+cudaMemcpy of spheres, materials and
 * lights in ray_tracing_cuda::RayTrace (rt_cuda/rt_cuda.cu:186-223).  centers_radii = sphere_count x
 * {cx, cy, cz, radius} (CudaSphere's centre and radius), material_index = sphere_count indices into
 * materials; centres and radii must be finite (RT_HIP_ERR_INVALID otherwise).  The caller keeps its arrays. */
int rt_hip_scene_create(const float * centers_radii, const uint32_t * material_index, uint32_t sphere_count,
                        const rt_hip_material * materials, uint32_t material_count,
                        const rt_hip_light * lights, uint32_t light_count, rt_hip_scene ** out_scene);
void rt_hip_scene_destroy(rt_hip_scene * scene);

/* Replaces ray_tracing_cuda::RayTrace (rt_cuda/rt_cuda.h:19-23): enqueues one frame on one of the library's own
 * non-blocking streams and returns immediately with a job handle (instead of a cudaEvent_t).  No host
 * worker thread and no realtimeHandler callback are needed: progress is pulled with rt_hip_job_snapshot.
 * Successive jobs take two streams in turn (every job has buffers of its own), so a caller that keeps jobs in
 * flight -- starts frame k + 2 before it finishes frame k -- gets the end of one trace, the beginning of the next and
 * the download of the one before side by side (a stream of 1080p 16x16 frames, each downloaded: 17.6 ms per frame
 * with three in flight, 18.2 with two). */
int rt_hip_trace_async(const rt_hip_scene * scene, const rt_hip_camera * camera,
                       const rt_hip_trace_params * params, rt_hip_job ** out_job);

/* Replaces ray_tracing_cuda::InProgress (rt_cuda/rt_cuda.h:26): RT_HIP_OK when the frame is complete,
 * RT_HIP_ERR_NOT_READY while it runs. */
int rt_hip_job_poll(rt_hip_job * job);

/* Replaces ray_tracing_cuda::CopyOutputToBuffer (rt_cuda/rt_cuda.h:29): copies what has been rendered so
 * far (unfinished pixels are black) into rgb without stalling the trace.  The first snapshot of a job allocates its
 * image buffers (two device-synchronising calls, about 2 x 12 bytes per pixel on the device and 12 in pinned host
 * memory); they are released when the job is finished.  A frame too short to be watched while it renders (fewer
 * than 1.5 G samples, about a tenth of a second) may be traced in a way that leaves nothing to look at before its end:
 * a snapshot of such a job waits for the frame and returns the finished image. */
int rt_hip_job_snapshot(rt_hip_job * job, float * rgb);

/* Replaces ray_tracing_cuda::FinishRayTrace (rt_cuda/rt_cuda.h:32): waits for the frame, writes the final
 * image to rgb (may be NULL), fills stats (may be NULL) and destroys the job.  The image is already resolved on
 * the device when the frame completes; the download runs on a copy stream of its own, so with a second job
 * started before this call it overlaps that job's trace (fastest into pinned host memory). */
int rt_hip_job_finish(rt_hip_job * job, float * rgb, rt_hip_stats * stats);

/* Replaces the needInterrupt flag returned by the reference's realtimeHandler
 * (rt_cuda/demo_frame_cuda.cpp:60-65, rt_cuda/rt_cuda.cu:271-275): asks the running kernel to stop
 * handing out pixels.  The job must still be finished with rt_hip_job_finish. */
int rt_hip_job_cancel(rt_hip_job * job);

/* Device-resident form of the same frame, for callers that own device memory and a stream (multi-GPU
 * drivers, benchmarks): traces the requested slice into d_sum_planes (float[3*h*w], device memory, fully
 * overwritten) on `stream` (a hipStream_t passed as void*; NULL = the default stream), asynchronously.
 * d_stats (device memory, 128 bytes, may be NULL) receives the nine uint64 counters of rt_hip_stats for
 * this call, in declaration order (the kernel also keeps its queue word there, so calls in flight at the same
 * time need different buffers).  Calls on one stream run in order; calls on different streams may overlap --
 * the library keeps a grow-only workspace per stream, eight of them (a ninth stream takes over the least
 * recently used one after a device-wide wait) -- which lets the start of one frame fill the compute units
 * that the end of the previous frame leaves idle. */
int rt_hip_trace_device(const rt_hip_scene * scene, const rt_hip_camera * camera,
                        const rt_hip_trace_params * params, float * d_sum_planes, void * d_stats,
                        void * stream);

/* rt_hip_trace_device without any allocation.  No counterpart in the reference: it is for callers that cannot
 * call hipMalloc -- code under HIP graph capture, tight serving loops, harnesses that forbid allocation.  Same
 * arguments, same output (bit for bit) as rt_hip_trace_device, but the scratch the kernels need (counters, camera
 * table, column sums, sample-sharing room) is a 64 MiB file-scope device array of the library instead of a
 * workspace.  The scene may have at most 16384 spheres.  A frame whose column sums fit the array (12 bytes per
 * sub-sample column: width * height * samples_in_row up to 5 570 538, e.g. 1024 x 768 at 2 x 2 ... 7 x 7 samples per
 * pixel) is traced as rt_hip_trace_device traces it; a larger one is traced with RT_HIP_FLAG_KERNEL_SUMS where that
 * flag applies (samples_in_row 1, 2, 4, 8, 16, 32 or 64: any frame size, e.g. 1920 x 1080 at 16 x 16) and refused
 * with RT_HIP_ERR_INVALID and a message naming the limit where it does not.  Every call on a GPU uses that GPU's one
 * array, so calls that overlap on one GPU (on different streams or from different threads) race; calls on one stream
 * are ordered and safe.  Calls no hipMalloc / hipFree / hipHostMalloc and nothing that waits for the device, so it may
 * be captured into a HIP graph (once the scene has been traced once outside a capture). */
int rt_hip_trace_device_static(const rt_hip_scene * scene, const rt_hip_camera * camera,
                               const rt_hip_trace_params * params, float * d_sum_planes, void * d_stats,
                               void * stream);

/* The accumulate/resolve pass: d_rgb[p*3+c] = d_sum_planes[c*h*w + p] * (1 / S^2)
 * (ray_tracing/frame.cpp:43,67), planar sums -> interleaved image, on `stream`. */
int rt_hip_resolve_device(const float * d_sum_planes, float * d_rgb, uint32_t width, uint32_t height,
                          uint32_t samples_in_row, void * stream);

/* No counterpart in the reference: a self-check of the device arithmetic.  The exact intersections and the
 * normalisations of the trace kernel take correctly rounded square roots and quotients from sequences shorter
 * than the compiler's general ones; this runs both on the device the library was initialised on -- every float
 * for sqrt and for 1 / sqrt, one operand pair per float pattern for the quotient and for the near root of
 * Sphere::TraceNearest (ray_tracing/sphere.cpp:72-74) -- and reports the number of differing results of each of the
 * four comparisons (all zero on a device the library supports).  About 0.1 s. */
int rt_hip_check_exact_math(uint64_t * mismatches4);

/* No counterpart in the reference: a self-check of the device's random number generator.  Runs the kernels'
 * Philox4x32-10 on the device over the three known-answer inputs of the Random123 distribution (counter and key all
 * zeros; all ones; digits of pi) -- words12 receives the three blocks of four words -- and over 2^20 further counters
 * and keys, folded into one word (run_device); run_host receives the same fold computed on the host by the routine as
 * published.  The device builds each 32 x 32 -> 64 bit product with one instruction through inline assembly;
 * this is where that form is held against the known answers directly. */
int rt_hip_check_philox(uint32_t * words12, uint32_t * run_device, uint32_t * run_host);

/* Average device time (ms) of the trace kernel over `launches` back-to-back launches of the given frame on
 * the library's stream, measured with HIP events on that stream; used by the benchmark's roofline line. */
int rt_hip_time_trace_kernel(const rt_hip_scene * scene, const rt_hip_camera * camera,
                             const rt_hip_trace_params * params, uint32_t launches, float * out_avg_ms,
                             rt_hip_stats * out_stats);

/* No counterpart in the reference: how the library would trace a frame -- which kernel, how many blocks, how the
 * pixel sums are taken -- without tracing it.  For tests and benchmarks, which want to know that the path they mean to
 * measure is the one that runs; the answer is that of rt_hip_trace_device with a workspace of its own choosing. */
typedef struct
{
  uint32_t blocks;        /* persistent blocks of 256 threads in the launch */
  uint32_t lds_bytes;     /* dynamic LDS per block */
  uint32_t batch_items;   /* queue positions (sub-sample columns) a wave reserves at a time ... */
  uint32_t batch_min;     /* ... and near the end of the queue */
  uint32_t tiled;         /* the scene is staged through LDS in tiles (else: resident in LDS, or walked on the grid) */
  uint32_t grid;          /* rays walk the uniform grid (RT_HIP_FLAG_GRID on a scene that has one) */
  uint32_t forward;       /* path colours accumulated forward (else the reference's reverse loop over a stack) */
  uint32_t one_light;     /* kernel specialised for one light */
  uint32_t library_math;  /* kernel built with the compiler's IEEE square root and division */
  uint32_t column_masks;  /* primary rays filtered once per sub-sample column */
  uint32_t sample_sharing;/* idle lanes take samples of busy lanes' columns at the end of the queue */
  uint32_t kernel_sums;   /* pixel sums taken inside the kernel's loop (RT_HIP_FLAG_KERNEL_SUMS) */
  uint32_t wave_sums;     /* pixel sums taken by the kernel's waves as they run out of work; neither: the fold and
                             accumulate passes follow the kernel */
  uint32_t reserved[3];
} rt_hip_launch_info;
int rt_hip_describe_launch(const rt_hip_scene * scene, const rt_hip_camera * camera,
                           const rt_hip_trace_params * params, rt_hip_launch_info * out);

#ifdef __cplusplus
}
#endif
#endif
