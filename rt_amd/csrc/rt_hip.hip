// rt_hip.hip -- gfx950 kernels and the C ABI of the MI355X ray-tracing backend (include/rt_hip.h).
//
// Kernels
//   trace_kernel<false, *, *>  persistent waves; the whole scene (<= kResidentSpheres spheres) sits in LDS for
//                        the life of the block; every lane runs the path state machine of rt_core.h and
//                        takes sub-sample columns from one frame-wide atomic counter, a reservation of 64 or
//                        more per atomic, fewer towards the end.  Waves never wait for each other.
//   trace_kernel<true, *, *>   the same loop for larger scenes: the block re-stages the spheres through LDS in
//                        tiles of kTileSpheres inside every ray pass, so its waves stay in step.
//   trace_kernel<*, true, *>   paths accumulated forward in registers; <*, false, *> keeps the reference's
//                        reverse loop over a 10-entry stack per lane in LDS (40 KB per block).
//   trace_kernel<*, *, true>   scenes with one light: no shading state carried from trip to trip.
//   trace_kernel_grid    the same loop with rays walking the scene's uniform grid (RT_HIP_FLAG_GRID).
//   trace_kernel<*, true, *, false, true>  the kForward kernels once more, adding up the pixels themselves: a wave
//                        keeps the column sums of its queue windows in LDS (PixelRing) and writes 12 bytes per pixel
//                        when a window is done (RT_HIP_FLAG_KERNEL_SUMS; an option: these kernels run 3-5 % longer).
//   trace_kernel<false, *, *, *, false, true>  the resident kernels once more, with the passes below moved to the end of
//                        each wave's life: a wave notes its queue reservations in LDS (one word each; the loop is
//                        otherwise untouched), and when it has no more work it completes its shared columns and adds
//                        up the pixels of its own columns from the column sums it wrote (wave_sums).  What launches
//                        that fill the chip use: no fold or accumulate launch between two frames of a stream.
//   trace_kernel<true, true, *>  also hand the block's primary rays to its first lanes before a round over the
//                        staged tiles (BlockEnv::gather_rays): a wave without a ray skips the round's arithmetic.
//   fold_kernel          completes the sums of columns whose last samples were traced by other lanes of the
//                        owner's wave (sample sharing at the end of the queue).
//   trace_kernel<*, *, false, true>  the kernels for any number of lights built with the compiler's general IEEE
//                        square root and division (scenes outside the bounds of the short sequences).
//   colsum_kernel        accumulate pass: per pixel, its column sums -> planar pixel sums (a lane per pixel).
//   colsum_lds_kernel    the same for the common sample counts: whole-line loads, transposed through LDS.
//   resolve_kernel       planar un-normalised sums -> interleaved RGB * 1/S^2 (ray_tracing/frame.cpp:43,67),
//                        16 bytes per lane per access.  These are the HBM-bound passes.
//   zero_kernel          clears a job's column sums (what a snapshot may read before the trace has written it).
//   exact_math_check_kernel  self-check of the short correctly rounded sequences (rt_hip_check_exact_math).
//   philox_check_kernel  the random number generator on known-answer inputs (rt_hip_check_philox).
//
//   camtab_kernel        per-launch set-up: table of the primary-ray filter (one entry per sphere), counters reset.
//
// LDS per block: the resident scene (56 B per sphere, 48 B per material and light) and 8 KB of prepared
// columns, or one 20 KB staging tile for scenes above kResidentSpheres spheres; with RT_HIP_FLAG_KERNEL_SUMS also the
// waves' rings (ring_bytes_per_wave: 0.4 KB + 768 B per window a wave may keep unfinished, as many as fit beside
// the rest without costing a block per compute unit); the kernels whose waves take the pixel sums at their end: 8 KB
// of reservation logs in front of everything else (kLogBlockBytes); the tiled forward kernels: 7 KB for the exchange
// of primary rays (RayExchange).
#include "rt_core.h"
#include "rt_hip.h"
#include "rt_prep.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <vector>

namespace
{
using namespace rtc;

// Tuning knobs an experiment build may override (-D...); the values here are the measured best.
#ifndef RT_WAVE_UNROLL
#define RT_WAVE_UNROLL 4
#endif
#ifndef RT_BLOCK_UNROLL
#define RT_BLOCK_UNROLL 4
#endif
#ifndef RT_TILED_WAVES
#define RT_TILED_WAVES 4
#endif
constexpr int kBlock = 256;               // 4 waves
constexpr int kWavesPerSimd = 3;          // blocks per CU the register budget is sized for (path stacks in LDS)
// Without the stacks four waves per SIMD fit, 128 registers each (7 % faster than three).  That fills the
// register file, so the launch of a job leaves kSnapshotReserveBlocks block slots of the chip free: the
// snapshot kernels of a running job run there instead of waiting for the frame to end.  (Measured: with 32
// free slots, one per shader engine of the 8 XCDs, a snapshot runs beside the trace; with 16 it waits.)
constexpr int kWavesPerSimdForward = 4;
constexpr int kSnapshotReserveBlocks = 64;
// ... but only for frames long enough for anyone to look at them while they render (the reference's viewer
// asks every 0.25 s, app.cpp:64-71): below this many samples (about 0.1 s of tracing) a job takes the whole chip
// and a snapshot waits for the frame's end.
constexpr unsigned long long kSnapshotRoomSamples = 1500000000ull;
constexpr uint32_t kResidentMaterials = 256;  // resident scenes also keep their materials and lights in LDS
constexpr uint32_t kResidentLights = 16;
constexpr uint32_t kColumnMaskSpheres = 64;   // column masks: two 32-bit words per lane
constexpr uint32_t kGridMinSpheres = 64;      // smaller scenes get no grid: testing every sphere is faster
constexpr uint32_t kResidentSpheres = 256; // scenes up to this size live in LDS whole (48 B per sphere)
constexpr uint32_t kTileSpheres = 1024;   // larger scenes: 20 KB of LDS per staged tile of filter records and cone terms
constexpr uint32_t kTileSpheresStack = 768;  // ... 15 KB where the path stacks (40 KB) share the block's 64 KB
constexpr uint32_t kStatsWords = 9;
constexpr uint32_t kStatsWordsAll = kStatsWords + 2;  // the counters, the queue word, TraceArgs::ring_errors

// ---- rt_hip_trace_device_static: the one entry point that allocates nothing ----
// Its scratch (counters, camera table, column sums, sample-sharing room) is this file-scope array instead of a
// workspace from hipMalloc.  Every GPU the library runs on holds its own copy, at its own address, until the
// process ends; host code never uses the array's name as an address, only the per-device pointer that
// hipGetSymbolAddress returns (resolve_static_scratch).  Layout:
//   [0, 256)                         counters + queue word, for callers that pass no d_stats
//   [256, 256 + 16 * kStaticMaxSpheres)  camera table
//   [kStaticPartialOffset, end)      column sums (12 bytes per column of the frame), then, if it still fits,
//                                    the records and colours of sample sharing
constexpr size_t kStaticScratchBytes = (size_t)64u << 20;  // 64 MiB per GPU
constexpr uint32_t kStaticMaxSpheres = 16384;
constexpr size_t kStaticCamtabOffset = 256;
constexpr size_t kStaticPartialOffset = kStaticCamtabOffset + sizeof(f4) * (size_t)kStaticMaxSpheres;
constexpr size_t kStaticPartialBytes = kStaticScratchBytes - kStaticPartialOffset;
static_assert(kStaticPartialOffset % 256u == 0u, "the column sums start on a 256-byte boundary");
constexpr int kMaxDevices = 64;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

// ---------------------------------------------------------------- device side

// The wave's index in its block as a scalar (threadIdx.x / 64 is the same in every lane, which the compiler cannot
// know: taken from the first lane, what is computed from it lives in scalar registers).
__device__ __forceinline__ uint32_t wave_in_block() { return (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64u)); }
// The lane's index, opaque to the compiler: what is computed from it is computed where it is used, not once ahead
// of the kernel's main loop and kept in a register across it.
__device__ __forceinline__ uint32_t lane_here()
{
  uint32_t lane = __lane_id();
  asm volatile("" : "+v"(lane));
  return lane;
}


// Work items come from one frame-wide counter, but a wave only touches it once per A.batch_items items (a
// multiple of kItemBatch chosen by the host: a single counter word sustains well under a hundred reservations
// per microsecond, far fewer than the waves would ask for item by item, or even batch by batch when columns
// are short).  The wave works through its reservation in windows of kItemBatch, serving its lanes from the
// window with a ballot and a prefix count.  Position k of the queue is item k (every slice visits every column).
#if defined(RT_PROFILE_REGIONS)
// Per wave of the resident kernel: constant-rate clock (100 MHz) at entry, after staging, after the loop, and when the
// wave found the queue used up.
constexpr uint32_t kTimelineWaves = 8192;
__device__ unsigned long long g_wave_times[4 * kTimelineWaves];
__device__ __forceinline__ void timeline_mark(uint32_t k)
{
  uint32_t const wave = blockIdx.x * (kBlock / 64u) + threadIdx.x / 64u;
  if (lane_here() == 0u && wave < kTimelineWaves)
    g_wave_times[4u * wave + k] = wall_clock64();
}
#endif
constexpr uint32_t kItemBatch = 64;
constexpr uint32_t kRunPixels = 64;  // slices by pixels: pixels per run (its items: a multiple of kItemBatch)
constexpr uint32_t kCancelPoll = 8;
constexpr uint32_t kGuide = 16;

// Pixel sums inside the trace kernel (TraceArgs::ring_slots).  Every wave keeps, in LDS, the column sums of the
// windows it has opened and not yet finished: up to `slots` windows of kItemBatch columns, oldest (tail) to newest
// (head), each with a header (columns of the window that are items of the queue, first pixel).  A lane that finishes
// a column parks its sum in the window's slot.  Windows are finished in the order they were opened: the oldest one
// is complete once all its columns have been handed out and no lane of the wave holds one of them any more -- one
// ballot over Lane::slot, which names the window in bits 6..8 -- and then the wave adds each pixel's columns in
// increasing sx and writes the pixel sums.  LDS operations of one wave are carried out in program order, so a sum
// written in an earlier instruction (by any lane) is there when a later one reads it; the wave-level fences only
// keep the compiler from moving the accesses.
//
// With sample sharing (the end of the queue) a lane that traces a sample of another lane's column holds that
// column's window too (its slot names it), and the colour goes through memory (TraceArgs::shared) as it does for
// the fold pass.  The owner leaves `meta[its lane]` = where the column belongs and the first sample that was handed
// out; when the window is complete, the owner's lane adds the shared colours to its own part in sample order --
// the additions the owner would have made itself -- before the pixels are summed.
constexpr uint32_t kRingHeaderWords = 2;
constexpr uint32_t kRingMaxSlots = 8;
constexpr uint32_t kRingMinSlots = 2;
constexpr uint32_t kRingStallLimit = 1u << 16;  // rounds a wave may wait for a slot before it gives the frame up (a
                                                // column of 64 samples of ten bounces ends within some 1500)
constexpr uint32_t kRingMetaValid = 0x80000000u;

// Number of lanes below this one whose bit is set in `mask`: two instructions, and no per-lane mask to keep.
__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask)
{
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Window a lane's slot names (both forms of Lane::slot in a kernel that takes the pixel sums itself: slot * 64 +
// column for a column from the queue; RT_SHARED_SLOT | owner lane << 16 | slot << 6 | place for a shared sample).
__device__ __forceinline__ uint32_t ring_window_of(uint32_t lane_slot) { return (lane_slot >> 6) & 7u; }

// LDS of one wave's ring: [meta 64][headers kRingMaxSlots x kRingHeaderWords][state kRingStateWords][sums slots x
// kItemBatch x 3].  Whatever the ring needs only now and then lives in the state words, not in scalar registers (the
// kernel has none to spare: every one more is one more spilled, and the spills cost a vector register): the queue
// position behind the newest window's last column, the count of refused openings, and the arguments of the launch
// that only the end of a window reads.
constexpr uint32_t kRingStateWords = 12;
constexpr uint32_t kRingStalls = 1, kRingPlanes = 2, kRingPixels = 4, kRingShared = 5, kRingShareDepth = 7,
                   kRingErrors = 8;
__host__ __device__ constexpr size_t ring_bytes_per_wave(uint32_t slots)
{
  return slots == 0u ? 0u : sizeof(uint32_t) * (64u + kRingMaxSlots * kRingHeaderWords + kRingStateWords) + (size_t)slots * 3u * 64u * sizeof(float);
}

struct PixelRing
{
  uint32_t * words = nullptr;   // LDS: the wave's region (all offsets below are constants)
  // One scalar: bits 0..7 the slot of the newest window (the one being handed out), 8..15 the number of unfinished
  // windows (the slots head - open + 1 ... head, modulo the slot count), 16..23 the slot count, 24..31 log2(s_row).
  uint32_t packed = 0;

  __device__ __forceinline__ uint32_t head() const { return packed & 0xFFu; }
  __device__ __forceinline__ uint32_t open() const { return (packed >> 8) & 0xFFu; }
  __device__ __forceinline__ uint32_t slots() const { return (packed >> 16) & 0xFFu; }
  __device__ __forceinline__ uint32_t s_shift() const { return packed >> 24; }
  __device__ __forceinline__ uint32_t * meta() const { return words; }
  __device__ __forceinline__ uint32_t * hdr() const { return words + 64u; }
  __device__ __forceinline__ uint32_t * state() const { return words + 64u + kRingMaxSlots * kRingHeaderWords; }
  __device__ __forceinline__ float * sums() const { return reinterpret_cast<float *>(words + 64u + kRingMaxSlots * kRingHeaderWords + kRingStateWords); }
  __device__ __forceinline__ uint32_t state_word(uint32_t k) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)state()[k]); }
  template <class T>
  __device__ __forceinline__ T * state_pointer(uint32_t k) const
  {
    return reinterpret_cast<T *>((unsigned long long)state()[k] | ((unsigned long long)state()[k + 1u] << 32));
  }
  __device__ __forceinline__ uint32_t tail() const
  {
    uint32_t const t = head() + 1u + slots() - open();  // in [1, 2 slots)
    return t >= slots() ? t - slots() : t;
  }

  __device__ __forceinline__ void init(const TraceArgs & A, unsigned char * wave_lds)
  {
    words = reinterpret_cast<uint32_t *>(wave_lds);
    uint32_t const lane = __lane_id();
    meta()[lane] = 0u;
    if (lane == 0u)
    {
      unsigned long long const planes = reinterpret_cast<unsigned long long>(A.planes);
      unsigned long long const shared = reinterpret_cast<unsigned long long>(A.shared);
      unsigned long long const errors = reinterpret_cast<unsigned long long>(A.ring_errors);
      state()[kRingStalls] = 0u;
      state()[kRingPlanes] = (uint32_t)planes;
      state()[kRingPlanes + 1u] = (uint32_t)(planes >> 32);
      state()[kRingPixels] = A.n_pixels;
      state()[kRingShared] = (uint32_t)shared;
      state()[kRingShared + 1u] = (uint32_t)(shared >> 32);
      state()[kRingShareDepth] = A.share_depth;
      state()[kRingErrors] = (uint32_t)errors;
      state()[kRingErrors + 1u] = (uint32_t)(errors >> 32);
    }
    packed = (A.ring_slots - 1u) | (A.ring_slots << 16) | (A.s_shift << 24);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __device__ __forceinline__ void raise_error() const { atomicAdd(state_pointer<unsigned long long>(kRingErrors), 1ull); }
  // May the next window be opened?  Not while every slot holds an unfinished window.
  __device__ __forceinline__ bool can_open() const { return open() < slots(); }
  // Counts a refused opening; true when the wave has been refused kRingStallLimit times in a row.
  __device__ __forceinline__ bool refused()
  {
    uint32_t const n = state_word(kRingStalls) + 1u;
    if (lane_here() == 0u)
      state()[kRingStalls] = n;
    return n >= kRingStallLimit;
  }
  // The window of `valid` columns whose first item is `first_item` takes the next slot.
  __device__ __forceinline__ void opened(uint32_t valid, uint32_t first_item)
  {
    uint32_t const h = head() + 1u == slots() ? 0u : head() + 1u;
    packed = ((packed & 0xFFFFFF00u) | h) + 0x100u;
    if (lane_here() == 0u)
    {
      hdr()[h * kRingHeaderWords + 0u] = valid;
      hdr()[h * kRingHeaderWords + 1u] = first_item >> s_shift();
      state()[kRingStalls] = 0u;
    }
  }

  // The oldest window is complete: shared samples join their columns, the pixels are summed and written, the slot
  // is free.  `end_of_queue`: samples may have been shared (only then).  s_row, slice_count: TraceArgs'.
  __device__ __forceinline__ void complete_tail(uint32_t s_row, uint32_t slice_count, uint32_t lane, bool end_of_queue)
  {
    uint32_t const w = tail();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (end_of_queue)
    {
      // Lane l looks after the column lane l owned: its own part, then the shared colours in increasing sy
      // (rt_core.h fold_split does the same after a kernel that leaves column sums).
      uint32_t const m = meta()[lane];
      if ((m & kRingMetaValid) != 0u && ring_window_of(m) == w)
      {
        uint32_t const wc = m & 0x1FFu;
        float * const s = sums() + 3u * wc;
        f3 sum = mk3(s[0], s[1], s[2]);
        uint32_t const wave_first = blockIdx.x * blockDim.x + wave_in_block() * 64u;
        size_t const base = (size_t)(wave_first + lane) * state()[kRingShareDepth];
        const float * const shared = state_pointer<const float>(kRingShared);
        for (uint32_t sy = (m >> 16) & 0x7FFFu; sy < s_row; sy += slice_count)
        {
          // written by other lanes of this wave: read past the vector cache, which may hold an older copy of the line
          const float * const v = shared + 3u * (base + (s_row - 1u - sy));
          float const x = __hip_atomic_load(v + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          float const y = __hip_atomic_load(v + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          float const z = __hip_atomic_load(v + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          sum = sum + mk3(x, y, z);
        }
        s[0] = sum.x;
        s[1] = sum.y;
        s[2] = sum.z;
        meta()[lane] = 0u;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    uint32_t const valid = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr()[w * kRingHeaderWords + 0u]);
    uint32_t const first_pixel = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr()[w * kRingHeaderWords + 1u]);
    if (lane < (valid >> s_shift()))
    {
      // the accumulate pass of this pixel: 0 + column 0 + column 1 + ... as colsum_kernel adds them
      const float * c = sums() + 3u * (w * 64u + (lane << s_shift()));
      float r = 0.0f, g = 0.0f, b = 0.0f;
      for (uint32_t i = 0; i < s_row; ++i, c += 3)
      {
        r += c[0];
        g += c[1];
        b += c[2];
      }
      size_t const n_pixels = state()[kRingPixels];
      float * const planes = state_pointer<float>(kRingPlanes);
      size_t const px = (size_t)first_pixel + lane;
      if (px < n_pixels)
      {
        planes[px] = r;
        planes[n_pixels + px] = g;
        planes[2u * n_pixels + px] = b;
      }
      else
        raise_error();  // (never: a window's pixels are pixels of the frame)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    packed -= 0x100u;
  }

  // Lanes with `finishing` have a complete column or shared sample in L.sum: it goes to its place in the ring (a
  // shared sample's colour to memory).  No window is finished here: that happens when the wave opens its next
  // window (retire) and when it has no more work (drain), not in every round.
  __device__ __forceinline__ void park(const TraceArgs & A, const Lane & L, bool finishing) const
  {
    if (!finishing)
      return;
    if ((L.slot & RT_SHARED_SLOT) == 0u)
    {
      float * const s = sums() + 3u * (L.slot & 0x1FFu);
      s[0] = L.sum.x;
      s[1] = L.sum.y;
      s[2] = L.sum.z;
      if (L.sy_end < A.s_row)  // samples of this column were handed to other lanes: this is the owner's part
        meta()[lane_here()] = kRingMetaValid | (L.sy_end << 16) | (L.slot & 0x1FFu);
    }
    else
    {
      uint32_t const owner = (L.slot >> 16) & 63u;
      uint32_t const place = L.slot & 63u;
      uint32_t const wave_first = blockIdx.x * blockDim.x + wave_in_block() * 64u;
      uint32_t const depth = state()[kRingShareDepth];
      if (place < depth)
        store3(state_pointer<float>(kRingShared) + 3u * ((size_t)(wave_first + owner) * depth + place), L.sum);
      else
        raise_error();  // (never: the places of a region are share_depth)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the colour is in memory before its window is finished
    }
  }

  // Finishes the oldest windows that no lane holds a column (or a shared sample) of any more.  Called when every
  // open window has been handed out completely: when the wave is about to open the next one.  holds, lane_slot:
  // whether the lane holds a column or shared sample, and its Lane::slot.
  __device__ __forceinline__ void retire(const TraceArgs & A, bool holds, uint32_t lane_slot, bool end_of_queue)
  {
    while (open() != 0u && __ballot(holds && ring_window_of(lane_slot) == tail()) == 0ull)
      complete_tail(A.s_row, A.slice_count, lane_here(), end_of_queue);
  }

  // The wave has no more work: every column it was handed is done, so every window still open is complete.
  __device__ __forceinline__ void drain(const TraceArgs & A)
  {
    while (open() != 0u)
      complete_tail(A.s_row, A.slice_count, lane_here(), true);
  }
};

// Pixel sums at the end of a wave's life (kernels with kLog; the resident layout only).  The kernel leaves column sums
// as usual, and every wave notes its queue reservations in LDS -- one word each, nothing else happens inside the loop.
// When the wave has no more work, every column it was handed is in memory: it completes the columns whose last
// samples other lanes of it traced (the fold pass), then goes through its reservations window by window, adds each
// pixel's columns in increasing sx and writes the pixel sums (the accumulate pass) -- the same additions in the same
// order as the two passes, by the wave that wrote the columns, with no launch that has to find block slots between
// two frames of a stream.  Per wave, at the start of the block's dynamic LDS: kLogEntries reservation words
// (queue position | windows - 1; positions are multiples of the 64-item window), then two words per lane for the one
// column a lane may leave with shared samples (its slot and the first shared sample; a lane owns at most one such
// column per launch, because they only arise once the wave's queue is used up and no lane gets another column).
// A wave whose log is full takes no more reservations; the host sizes reservations so that this stays theory, and
// checks that the logs of a launch can hold every reservation the queue can be cut into.
//
// What the end of a wave needs of the launch's arguments is parked in LDS when the block starts (EndArgs, in front of
// the logs) and read back from there: an argument that is used after the main loop would otherwise sit in a scalar
// register across it, and the loop has none to spare.
constexpr uint32_t kLogEntries = 384;
constexpr uint32_t kLogWords = kLogEntries + 2u * 64u;
constexpr uint32_t kLogMaxWindows = 64;   // windows of a reservation (6 bits of the word)
constexpr uint32_t kLogMaxQuantum = 8;    // windows of the smallest reservation that holds whole pixels (EndArgs::quantum)
struct EndArgs
{
  float * partial;
  float * planes;
  const float * shared;
  uint32_t n_pixels, n_items, queue_items, run_items, run_stride, run_first, s_row, slice_count, share_depth;
  uint32_t log_entries;  // reservations a wave may note (<= kLogEntries; fewer in tests of what happens when a log is full)
  uint32_t quantum;      // every reservation is a multiple of this many queue positions and starts at one: the least
                         // common multiple of the 64-position window and s_row, so that it holds whole pixels
  uint32_t pad[3];
};
static_assert(sizeof(EndArgs) == 80, "EndArgs is read back word by word");
constexpr size_t kLogBlockBytes = sizeof(EndArgs) + sizeof(uint32_t) * kLogWords * (kBlock / 64);
__device__ __forceinline__ uint32_t log_entries()
{
  extern __shared__ __align__(16) unsigned char lds_raw[];
  return reinterpret_cast<const EndArgs *>(lds_raw)->log_entries;
}
__device__ __forceinline__ uint32_t log_quantum()
{
  extern __shared__ __align__(16) unsigned char lds_raw[];
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const EndArgs *>(lds_raw)->quantum);
}
__device__ __forceinline__ uint32_t * wave_log()
{
  extern __shared__ __align__(16) unsigned char lds_raw[];
  // (worked out where it is used, every time: a pointer computed once ahead of the main loop would live across it)
  uint32_t wave = threadIdx.x / 64u;
  asm volatile("" : "+v"(wave));
  return reinterpret_cast<uint32_t *>(lds_raw + sizeof(EndArgs)) + (uint32_t)__builtin_amdgcn_readfirstlane((int)wave) * kLogWords;
}

struct ItemPool
{
  uint32_t next = 0, end = 0;  // the current window (queue positions), wave-uniform
  uint32_t delta = 0;          // item of a position of the current window - that position (a window lies in one run)
  uint32_t reserved_end = 0;   // end of the wave's reservation
  uint32_t refills = 0;
  uint32_t last_size = 0;      // size of the wave's previous reservation
  bool exhausted = false;      // the last reservation lay past the end of the queue: nothing more will come

  // Opens the next window: the rest of the reservation, or a new reservation.  kLog: the wave notes its reservations
  // (wave_log) and stops taking them when the log is full.
  template <bool kLog>
  __device__ __forceinline__ void refill(const TraceArgs & A)
  {
    if (end == reserved_end)
    {
      // Reservations shrink towards the end of the queue (down to batch_min), so that the waves run dry within
      // a short reservation of each other: a wave takes 1 / (kGuide * waves) of what it believes is left, a
      // fraction small enough that a reservation in the most expensive part of the frame (several times the
      // average cost per sample) is still worked off before the others have emptied the queue.  It does
      // not read the counter for that (every wave reading the word the others add to costs more than it gains):
      // it reckons that since its own last reservation every wave made one more of the same size.
      uint32_t size = A.batch_items;
      if (A.batch_min < A.batch_items)
      {
        unsigned long long const seen = (unsigned long long)reserved_end + (unsigned long long)A.guide_waves * last_size;
        unsigned long long const left = seen < A.queue_items ? A.queue_items - seen : 0ull;
        // (whole windows; kernels with a log: whole pixels too, see EndArgs::quantum)
        unsigned long long quantum = kItemBatch;
        if constexpr (kLog)
          quantum = log_quantum();
        unsigned long long const want = left / ((unsigned long long)kGuide * A.guide_waves) / quantum * quantum;
        size = (uint32_t)min(max(want, (unsigned long long)A.batch_min), (unsigned long long)A.batch_items);
      }
      last_size = size;
      uint32_t base = 0;
      if (lane_here() == 0u)
      {
        // A cancelled job stops at a reservation: the wave is sent past the end of the queue.  The flag may
        // live in host memory, a slow read, so only every kCancelPoll-th reservation looks at it.
        bool cancelled = false;
        if constexpr (kLog)
          cancelled = refills >= log_entries();  // the log is full: the wave is sent past the end of the queue too
        if (!cancelled && (refills % kCancelPoll) == 0u)
          cancelled = __hip_atomic_load(A.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
        base = cancelled ? 0xF0000000u : atomicAdd(A.queue_counter, size);
        if (kLog && refills < kLogEntries)
          wave_log()[refills] = base | (size / kItemBatch - 1u);
      }
      refills += 1u;
      end = __builtin_amdgcn_readfirstlane(base);
      reserved_end = end + size;
      exhausted = end >= A.queue_items;
#if defined(RT_PROFILE_REGIONS)
      if (exhausted)
        timeline_mark(3);
#endif
    }
    next = end;
    end = next + kItemBatch;
    delta = queue_item(A, next) - next;
  }

  // The same with the wave's pixel sums in mind: false when the ring has no free slot for the window yet (older
  // windows have unfinished columns; the lanes that asked ask again in a later round).  A wave that is refused
  // kRingStallLimit times in a row -- which no frame does: a busy slot means a lane of this wave is still tracing
  // one of its columns -- gives up: it reports the queue used up, so that the launch ends, and raises
  // TraceArgs::ring_errors.
  // L, info: the lane's state, and what the call that asks for the window has handed the lane so far (a column
  // handed out a moment ago is not in L yet).
  template <bool kFused, bool kLog>
  __device__ __forceinline__ bool open(const TraceArgs & A, PixelRing & ring, const Lane & L, const ItemInfo & info)
  {
    if constexpr (kFused)
    {
      bool const just_got = info.pos != RT_INVALID;
      ring.retire(A, just_got || (L.alive && !L.need_item), just_got ? info.slot : L.slot, exhausted);
    }
    if (kFused && !ring.can_open())
    {
      if (!ring.refused())
        return false;
      if (lane_here() == 0u)
        ring.raise_error();
      exhausted = true;
      next = end;
      return true;
    }
    refill<kLog>(A);
    if (kFused && next < A.queue_items)
      ring.opened(min(kItemBatch, A.queue_items - next), next + delta);
    return true;
  }

  // Hands the next queue positions to the lanes that need one: info.pos (the item) and info.slot.
  template <bool kFused, bool kLog = false>
  __device__ __forceinline__ void fetch(const TraceArgs & A, PixelRing & ring, const Lane & L, bool need, ItemInfo & info)
  {
    for (;;)
    {
      unsigned long long const mask = __ballot(need);
      if (mask == 0ull)
        break;
      if (next == end)
      {
        if (exhausted)
          break;
        if (!open<kFused, kLog>(A, ring, L, info))
        {
          info.stalled = true;
          break;
        }
        if (next == end)
          break;  // the wave gave up (see open)
      }
      uint32_t const avail = end - next;
      uint32_t const wanted = (uint32_t)__popcll(mask);
      uint32_t const rank = lanes_below(mask);
      if (need && rank < avail)
      {
        if (next + rank < A.queue_items)
        {
          info.pos = next + rank + delta;
          info.slot = kFused ? ring.head() * kItemBatch + (next + rank - (end - kItemBatch)) : info.pos;
        }
        need = false;
      }
      next += wanted < avail ? wanted : avail;
    }
  }
};

__device__ __forceinline__ uint32_t lane_value(uint32_t v, uint32_t lane)
{
  return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane);
}
__device__ __forceinline__ float lane_value(float v, uint32_t lane)
{
  return __uint_as_float(lane_value(__float_as_uint(v), lane));
}

// Sample sharing inside a wave, once the queue is used up (TraceArgs::shared): the idle lanes are dealt out
// to the lanes that still have samples to start, an even share each, taking samples from the end of the
// owner's column.  Everything here is wave-wide and needs no memory traffic: the owner's ray set-up travels
// through lane reads.  The region of a lane is its index in the launch, so no two owners share one.
template <bool kFused>
__device__ __forceinline__ void share_samples(const TraceArgs & A, Lane & L)
{
  if (A.share_depth == 0u)
    return;
  unsigned long long const idle = __ballot(!L.alive);
  if (idle == 0ull)
    return;
  uint32_t const spare = spare_samples(A, L);
  unsigned long long donors = __ballot(spare != 0u);
  if (donors == 0ull)
    return;
  uint32_t const lane = lane_here();
  uint32_t const n_idle = (uint32_t)__popcll(idle);
  uint32_t const per = max(1u, n_idle / (uint32_t)__popcll(donors));
  uint32_t const my_rank = lanes_below(idle);
  bool const was_idle = !L.alive;
  uint32_t const wave_first = blockIdx.x * blockDim.x + wave_in_block() * 64u;
  uint32_t taken = 0u;
  while (donors != 0ull && taken < n_idle)
  {
    uint32_t const d = (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_ctzll(donors));
    donors &= donors - 1ull;
    uint32_t const give = min(min(lane_value(spare, d), per), n_idle - taken);
    uint32_t const d_end = lane_value(L.sy_end, d), d_sy = lane_value(L.sy, d);
    uint32_t const d_slot = kFused ? lane_value(L.slot, d) : 0u;
    uint32_t const d_pixel = lane_value(L.pixel, d), d_sx = kFused ? (d_slot & (A.s_row - 1u)) : lane_value(L.sx, d);
    uint32_t const d_cm0 = lane_value(L.cm0, d), d_cm1 = lane_value(L.cm1, d);
    float const d_sdx = lane_value(L.sdx, d), d_dyp = lane_value(L.dyp, d);
    if (lane == d)
    {
      L.sy_end = spare_sample_sy(A, d_sy, d_end, give - 1u);  // the owner keeps what lies below
    }
    if (was_idle && my_rank >= taken && my_rank < taken + give)
    {
      uint32_t const sy = spare_sample_sy(A, d_sy, d_end, my_rank - taken);
      take_shared_sample(A, L, (wave_first + d) * A.share_depth, d_pixel, d_sx, d_sdx, d_dyp, d_cm0, d_cm1, sy);
      if constexpr (kFused)  // the owner's lane, sx and window of its column, the colour's place in the owner's region
        L.slot = RT_SHARED_SLOT | (d << 16) | (d_sx << 9) | (ring_window_of(d_slot) << 6) | (A.s_row - 1u - sy);
    }
    taken += give;
  }
}

// The owner of a column some of whose samples were shared out leaves a record for the fold pass.
__device__ __forceinline__ void append_split(const TraceArgs & A, uint32_t slot, uint32_t first_shared_sy)
{
  uint32_t const at = atomicAdd(A.split_count, 1u);
  u4 rec;
  rec.x = slot;
  rec.y = first_shared_sy;
  rec.z = (blockIdx.x * blockDim.x + threadIdx.x) * A.share_depth;
  rec.w = 0u;
  A.splits[at] = rec;
}

// The whole scene is resident in LDS; waves never wait for each other.
template <bool kLibMath, bool kFused, bool kLog = false>
struct WaveEnv
{
  static constexpr bool kLibraryMath = kLibMath;
  static constexpr bool kCompactsRays = false;
  static constexpr uint32_t kFilterUnroll = RT_WAVE_UNROLL;
  static constexpr uint32_t kPrimaryUnroll = 2;  // scenes of 65 to 256 spheres; smaller ones filter per column
  const TraceArgs & A;
  const f4 * filt;       // LDS: n_padded filter records
  const f4 * cam;        // LDS: n_padded camera-table records
  const f4 * exact;      // LDS: n_padded exact records
  const f4 * mats;       // LDS: 2 * n_materials
  const f4 * lights;     // LDS: RT_LIGHT_RECORDS * n_lights
  const uint32_t * smat; // LDS: n_padded material indices
  const float * aux;     // LDS: n_padded cone terms (aux_cone for the one-light kernels, else aux_rw)
  const float * jitter;  // LDS: RT_JITTER_FLOATS
  ColumnEntry * batch;   // LDS: this wave's kItemBatch prepared columns (column masks only)
  ItemPool pool;
  PixelRing ring;

  __device__ __forceinline__ bool any(bool p) const { return __any(p) != 0; }
  __device__ __forceinline__ int count(bool p) const { return (int)__popcll(__ballot(p)); }
  __device__ __forceinline__ bool any_wave(bool p) const { return __any(p) != 0; }
  __device__ __forceinline__ bool endgame() const { return pool.exhausted; }
  __device__ __forceinline__ void share(Lane & L) const { share_samples<kFused>(A, L); }
  __device__ __forceinline__ void record_split(uint32_t slot, uint32_t sy) const
  {
    if constexpr (kLog)
    {
      // the wave completes the column itself when it has no more work (wave_sums)
      uint32_t * const mine = wave_log() + kLogEntries + 2u * lane_here();
      mine[0] = slot;
      mine[1] = sy;
    }
    else
      append_split(A, slot, sy);
  }
  // (kFused: a window holds whole pixels and starts at one, so a column's place in its window gives sx; a shared
  // sample carries the owner's sx in bits 9..14 of its slot)
  __device__ __forceinline__ uint32_t column_sx(const Lane & L) const
  {
    if constexpr (kFused)
      return (L.slot & RT_SHARED_SLOT) != 0u ? ((L.slot >> 9) & 63u) : (L.slot & (A.s_row - 1u));
    else
      return L.sx;
  }
  __device__ __forceinline__ void finish_columns(Lane & L, bool finishing)
  {
    if constexpr (kFused)
      ring.park(A, L, finishing);
    else if (finishing)
      store_column(A, L, *this);
  }
  __device__ __forceinline__ f4 sphere(int i) const { return exact[i]; }
  __device__ __forceinline__ uint32_t sphere_material(int i) const { return smat[i]; }
  __device__ __forceinline__ f4 material(uint32_t m, uint32_t k) const { return mats[RT_MAT_RECORDS * m + k]; }
  __device__ __forceinline__ f4 light(uint32_t l, uint32_t k) const { return lights[RT_LIGHT_RECORDS * l + k]; }
  __device__ __forceinline__ const float * jitter_table() const { return jitter; }
  __device__ __forceinline__ uint32_t tiles() const { return 1u; }
  __device__ __forceinline__ uint32_t tile_base(uint32_t) const { return 0u; }
  __device__ __forceinline__ uint32_t load_tile(uint32_t, bool) const { return A.n_padded; }
  __device__ __forceinline__ const f4 * tile_data(bool primary) const { return primary ? cam : filt; }
  __device__ __forceinline__ const float * tile_aux() const { return aux; }

  // With column masks the wave prepares a whole window when it opens one: lane l sets up the column at queue
  // position base + l (ray set-up and candidate masks, one filter pass per column instead of one per sample)
  // and parks it in the wave's LDS table, from where the lanes pick their columns up.
  __device__ __forceinline__ void fetch_item(const Lane & L, bool need, ItemInfo & info)
  {
    info.pos = RT_INVALID;
    info.slot = RT_INVALID;
    info.stalled = false;
    info.prepared = false;
    info.filter_tests = 0u;
    if (A.column_masks == 0u)
    {
      pool.template fetch<kFused, kLog>(A, ring, L, need, info);
      return;
    }
    uint32_t const lane = lane_here();
    for (;;)
    {
      unsigned long long const mask = __ballot(need);
      if (mask == 0ull)
        break;
      if (pool.next == pool.end)
      {
        if (pool.exhausted)
          break;
        if (!pool.template open<kFused, kLog>(A, ring, L, info))
        {
          info.stalled = true;
          break;
        }
        if (pool.next == pool.end)
          break;  // the wave gave up (see ItemPool::open)
        // (a window past the end of the queue gets invalid entries: lanes that ask later must not find the
        // previous window's columns)
        ColumnEntry e;
        e.pixel = RT_INVALID;
        if (pool.next + lane < A.queue_items)
          e = prepare_column(A, cam, pool.next + lane + pool.delta);
        if (pool.next < A.queue_items)
          info.filter_tests += min(kItemBatch, A.queue_items - pool.next);
        batch[lane] = e;
        __builtin_amdgcn_wave_barrier();
      }
      uint32_t const avail = pool.end - pool.next;
      uint32_t const wanted = (uint32_t)__popcll(mask);
      uint32_t const rank = lanes_below(mask);
      if (need && rank < avail)
      {
        uint32_t const column = pool.next + rank - (pool.end - kItemBatch);
        ColumnEntry const e = batch[column];
        if (e.pixel != RT_INVALID)
        {
          info.pos = pool.next + rank + pool.delta;
          info.slot = kFused ? ring.head() * kItemBatch + column : info.pos;
          info.prepared = true;
          info.sdx = e.sdx;
          info.dyp = e.dyp;
          info.cm0 = e.cm0;
          info.cm1 = e.cm1;
          info.pixel = e.pixel;
          info.sx = e.sx;
          info.sy0 = e.sy0;
        }
        need = false;
      }
      pool.next += wanted < avail ? wanted : avail;
    }
  }
};

// Larger scenes: the block re-stages the spheres (or the camera table) through one LDS tile inside every
// pass, so all of its waves take the same trips and every loop decision is block-wide.
// LDS through which a block's lanes hand their primary rays to its first lanes (BlockEnv::gather_rays): per lane a
// ray (direction and its squared length), the lane it came from, and the hit that goes back; a count per wave.
struct RayExchange
{
  f4 rays[kBlock];
  uint32_t homes[kBlock];
  float hit_t[kBlock];
  int hit_i[kBlock];
  uint32_t counts[kBlock / 64];
};
static_assert(sizeof(RayExchange) % 16u == 0u, "what follows it in LDS stays 16-byte aligned");

template <bool kLibMath, bool kFused, bool kCompact = false>
struct BlockEnv
{
  static constexpr bool kLibraryMath = kLibMath;
  static constexpr bool kCompactsRays = kCompact;
  static constexpr uint32_t kFilterUnroll = RT_BLOCK_UNROLL;  // almost all of this path's time is the filter
#ifndef RT_BLOCK_PRIMARY_UNROLL
#define RT_BLOCK_PRIMARY_UNROLL 4
#endif
  static constexpr uint32_t kPrimaryUnroll = RT_BLOCK_PRIMARY_UNROLL;
  const TraceArgs & A;
  f4 * tile;              // LDS: kTileSpheres entries (kTileSpheresStack in the kernels with path stacks)
  float * tile_cone;      // LDS: as many cone terms, staged with the filter records
  const float * aux_src;  // where those come from: A.aux_cone for the one-light kernel, else A.aux_rw
  const float * jitter;   // LDS: RT_JITTER_FLOATS
  ItemPool pool;
  PixelRing ring;
  RayExchange * exchange = nullptr;  // LDS (kCompact)

  // The block's lanes that hold a ray (`mine`) hand it to the block's first lanes, in lane order; afterwards lane j
  // holds ray j (mine = j < number of rays) and the return value is the lane the ray came from.
  __device__ __forceinline__ uint32_t gather_rays(bool & mine, f3 & dir, float & a) const
  {
    unsigned long long const ballot = __ballot(mine);
    uint32_t const wave = wave_in_block();
    if (lane_here() == 0u)
      exchange->counts[wave] = (uint32_t)__popcll(ballot);
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kBlock / 64u; ++w)
    {
      uint32_t const c = exchange->counts[w];
      before += w < wave ? c : 0u;
      total += c;
    }
    if (mine)
    {
      uint32_t const slot = before + lanes_below(ballot);
      f4 r;
      r.x = dir.x;
      r.y = dir.y;
      r.z = dir.z;
      r.w = a;
      exchange->rays[slot] = r;
      exchange->homes[slot] = threadIdx.x;
    }
    __syncthreads();
    mine = threadIdx.x < total;
    uint32_t home = 0;
    if (mine)
    {
      f4 const r = exchange->rays[threadIdx.x];
      dir = mk3(r.x, r.y, r.z);
      a = r.w;
      home = exchange->homes[threadIdx.x];
    }
    return home;
  }
  // ... and the hits go back to the lanes the rays came from (`active`: this lane had handed one in).
  __device__ __forceinline__ void return_hits(bool mine, uint32_t home, bool active, float & t, int & i) const
  {
    if (mine)
    {
      exchange->hit_t[home] = t;
      exchange->hit_i[home] = i;
    }
    __syncthreads();
    if (active)
    {
      t = exchange->hit_t[threadIdx.x];
      i = exchange->hit_i[threadIdx.x];
    }
    else
    {
      t = 0.0f;
      i = -1;
    }
  }

  __device__ __forceinline__ bool any(bool p) const { return __syncthreads_or(p) != 0; }
  __device__ __forceinline__ int count(bool p) const { return __syncthreads_count(p); }
  __device__ __forceinline__ bool any_wave(bool p) const { return __any(p) != 0; }
  __device__ __forceinline__ bool endgame() const { return pool.exhausted; }
  __device__ __forceinline__ void share(Lane & L) const { share_samples<kFused>(A, L); }
  __device__ __forceinline__ void record_split(uint32_t slot, uint32_t sy) const { append_split(A, slot, sy); }
  // (kFused: a window holds whole pixels and starts at one, so a column's place in its window gives sx; a shared
  // sample carries the owner's sx in bits 9..14 of its slot)
  __device__ __forceinline__ uint32_t column_sx(const Lane & L) const
  {
    if constexpr (kFused)
      return (L.slot & RT_SHARED_SLOT) != 0u ? ((L.slot >> 9) & 63u) : (L.slot & (A.s_row - 1u));
    else
      return L.sx;
  }
  __device__ __forceinline__ void finish_columns(Lane & L, bool finishing)
  {
    if constexpr (kFused)
      ring.park(A, L, finishing);
    else if (finishing)
      store_column(A, L, *this);
  }
  __device__ __forceinline__ void fetch_item(const Lane & L, bool need, ItemInfo & info)
  {
    info.pos = RT_INVALID;
    info.slot = RT_INVALID;
    info.stalled = false;
    info.prepared = false;
    info.filter_tests = 0u;
    pool.template fetch<kFused>(A, ring, L, need, info);
  }
  __device__ __forceinline__ f4 sphere(int i) const { return A.spheres[i]; }
  __device__ __forceinline__ uint32_t sphere_material(int i) const { return A.sphere_mat[i]; }
  __device__ __forceinline__ f4 material(uint32_t m, uint32_t k) const { return A.materials[RT_MAT_RECORDS * m + k]; }
  __device__ __forceinline__ f4 light(uint32_t l, uint32_t k) const { return A.lights[RT_LIGHT_RECORDS * l + k]; }
  __device__ __forceinline__ const float * jitter_table() const { return jitter; }
  __device__ __forceinline__ uint32_t tiles() const { return (A.n_padded + A.tile_size - 1u) / A.tile_size; }
  __device__ __forceinline__ uint32_t tile_base(uint32_t k) const { return k * A.tile_size; }
  __device__ __forceinline__ uint32_t load_tile(uint32_t k, bool primary) const
  {
    uint32_t const base = k * A.tile_size;
    uint32_t const count = min(A.tile_size, A.n_padded - base);
    const f4 * src = (primary ? A.camtab : A.filter) + base;
    __syncthreads();  // everyone is done with the previous tile
    for (uint32_t i = threadIdx.x; i < count; i += kBlock)
      tile[i] = src[i];
    if (!primary)
      for (uint32_t i = threadIdx.x; i < count; i += kBlock)
        tile_cone[i] = aux_src[base + i];
    __syncthreads();
    return count;
  }
  __device__ __forceinline__ const f4 * tile_data(bool) const { return tile; }
  __device__ __forceinline__ const float * tile_aux() const { return tile_cone; }
};

// Grid walks: every ray finds its own spheres, read from global memory (the L2 holds the scene); waves never
// wait for each other.  The tile functions are unused.
template <bool kLibMath, bool kFused>
struct GridEnv
{
  static constexpr bool kLibraryMath = kLibMath;
  static constexpr bool kCompactsRays = false;
  static constexpr uint32_t kFilterUnroll = 2;
  static constexpr uint32_t kPrimaryUnroll = 2;
  const TraceArgs & A;
  const float * jitter;  // LDS: RT_JITTER_FLOATS
  ItemPool pool;
  PixelRing ring;

  __device__ __forceinline__ bool any(bool p) const { return __any(p) != 0; }
  __device__ __forceinline__ int count(bool p) const { return (int)__popcll(__ballot(p)); }
  __device__ __forceinline__ bool any_wave(bool p) const { return __any(p) != 0; }
  __device__ __forceinline__ bool endgame() const { return pool.exhausted; }
  __device__ __forceinline__ void share(Lane & L) const { share_samples<kFused>(A, L); }
  __device__ __forceinline__ void record_split(uint32_t slot, uint32_t sy) const { append_split(A, slot, sy); }
  // (kFused: a window holds whole pixels and starts at one, so a column's place in its window gives sx; a shared
  // sample carries the owner's sx in bits 9..14 of its slot)
  __device__ __forceinline__ uint32_t column_sx(const Lane & L) const
  {
    if constexpr (kFused)
      return (L.slot & RT_SHARED_SLOT) != 0u ? ((L.slot >> 9) & 63u) : (L.slot & (A.s_row - 1u));
    else
      return L.sx;
  }
  __device__ __forceinline__ void finish_columns(Lane & L, bool finishing)
  {
    if constexpr (kFused)
      ring.park(A, L, finishing);
    else if (finishing)
      store_column(A, L, *this);
  }
  __device__ __forceinline__ void fetch_item(const Lane & L, bool need, ItemInfo & info)
  {
    info.pos = RT_INVALID;
    info.slot = RT_INVALID;
    info.stalled = false;
    info.prepared = false;
    info.filter_tests = 0u;
    pool.template fetch<kFused>(A, ring, L, need, info);
  }
  __device__ __forceinline__ f4 sphere(int i) const { return A.spheres[i]; }
  __device__ __forceinline__ uint32_t sphere_material(int i) const { return A.sphere_mat[i]; }
  __device__ __forceinline__ f4 material(uint32_t m, uint32_t k) const { return A.materials[RT_MAT_RECORDS * m + k]; }
  __device__ __forceinline__ f4 light(uint32_t l, uint32_t k) const { return A.lights[RT_LIGHT_RECORDS * l + k]; }
  __device__ __forceinline__ const float * jitter_table() const { return jitter; }
  __device__ __forceinline__ uint32_t tiles() const { return 0u; }
  __device__ __forceinline__ uint32_t tile_base(uint32_t) const { return 0u; }
  __device__ __forceinline__ uint32_t load_tile(uint32_t, bool) const { return 0u; }
  __device__ __forceinline__ const f4 * tile_data(bool) const { return nullptr; }
  __device__ __forceinline__ const float * tile_aux() const { return nullptr; }
};

__device__ __forceinline__ unsigned long long wave_sum(uint32_t lane_value)
{
  unsigned long long v = lane_value;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    v += __shfl_down(v, off);
  return v;
}

#if defined(RT_PROFILE_REGIONS)
__device__ unsigned long long g_region_ticks[RT_REGION_COUNT];
#endif

// Work counters: one 64-bit atomic per wave and counter.
__device__ __forceinline__ void flush_counters(const TraceArgs & A, LaneCounters & cnt)
{
  unsigned long long const c0 = wave_sum(cnt.samples);
  unsigned long long const c1 = wave_sum(cnt.queries);
  unsigned long long const c2 = wave_sum(cnt.hits);
  unsigned long long const c3 = wave_sum(cnt.shaded);
  unsigned long long const c4 = (wave_sum(cnt.filter_tests) + cnt.prepared) * A.n_spheres;
  unsigned long long const c8 = wave_sum(cnt.cone_tests);
  unsigned long long const c5 = wave_sum(cnt.exact_tests);
  unsigned long long const c6 = wave_sum(cnt.trips);
  unsigned long long const c7 = wave_sum(cnt.rounds);
  if (A.stats != nullptr && lane_here() == 0u)
  {
    atomicAdd(&A.stats[0], c0);
    atomicAdd(&A.stats[1], c1);
    atomicAdd(&A.stats[2], c2);
    atomicAdd(&A.stats[3], c3);
    atomicAdd(&A.stats[4], c4);
    atomicAdd(&A.stats[5], c5);
    atomicAdd(&A.stats[6], c6);
    atomicAdd(&A.stats[7], c7);
    atomicAdd(&A.stats[8], c8);
  }
#if defined(RT_PROFILE_REGIONS)
  RT_REGION(cnt, 4);
  if (lane_here() == 0u)
    for (int k = 0; k < RT_REGION_COUNT; ++k)
      atomicAdd(&g_region_ticks[k], cnt.region[k]);
#endif
}

// Slices by runs of pixels, kernels that take the pixel sums at the end of their waves: the sums of the pixels other
// slices trace are zero; the launch's waves share those runs out among themselves (64 lanes, one run of one plane).
__device__ __forceinline__ void zero_other_runs(const TraceArgs & A)
{
  if (A.run_items == 0u)
    return;
  uint32_t const lane = __lane_id();
  uint32_t const run_pixels = A.run_items / A.s_row;
  uint32_t const n_runs = (A.n_pixels + run_pixels - 1u) / run_pixels;
  uint32_t const waves = gridDim.x * (kBlock / 64u);
  for (uint32_t k = blockIdx.x * (kBlock / 64u) + wave_in_block(); k < 3u * n_runs; k += waves)
  {
    uint32_t const plane = k / n_runs, run = k - plane * n_runs;
    if (run % A.run_stride == A.run_first)
      continue;
    for (uint32_t i = lane; i < run_pixels; i += 64u)
    {
      uint32_t const px = run * run_pixels + i;
      if (px < A.n_pixels)
        A.planes[(size_t)plane * A.n_pixels + px] = 0.0f;
    }
  }
}

// The block's EndArgs, written by one lane before the block's first barrier and read back (as scalars) after the loop.
__device__ __forceinline__ void park_end_args(const TraceArgs & A)
{
  extern __shared__ __align__(16) unsigned char lds_raw[];
  EndArgs * const e = reinterpret_cast<EndArgs *>(lds_raw);
  e->partial = A.partial;
  e->planes = A.planes;
  e->shared = A.shared;
  e->n_pixels = A.n_pixels;
  e->n_items = A.n_items;
  e->queue_items = A.queue_items;
  e->run_items = A.run_items;
  e->run_stride = A.run_stride;
  e->run_first = A.run_first;
  e->s_row = A.s_row;
  e->slice_count = A.slice_count;
  e->share_depth = A.share_depth;
  e->log_entries = min(A.log_entries, kLogEntries);
  e->quantum = A.log_quantum;
  e->pad[0] = e->pad[1] = e->pad[2] = 0u;
}
__device__ __forceinline__ EndArgs read_end_args()
{
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const uint32_t * const w = reinterpret_cast<const uint32_t *>(lds_raw);
  uint32_t v[sizeof(EndArgs) / 4u];
#pragma unroll
  for (uint32_t k = 0; k < sizeof(EndArgs) / 4u; ++k)
    v[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)w[k]);
  EndArgs e;
  __builtin_memcpy(&e, v, sizeof(e));
  return e;
}
// rt_core.h queue_item for EndArgs
__device__ __forceinline__ uint32_t end_queue_item(const EndArgs & E, uint32_t pos)
{
  if (E.run_items == 0u)
    return pos;
  uint32_t const run = pos / E.run_items;
  return (run * E.run_stride + E.run_first) * E.run_items + (pos - run * E.run_items);
}

// The end of a wave in the kernels with kLog (see kLogEntries): fold pass and accumulate pass over the wave's own
// columns.  log: the wave's log; entries: reservations it made; stage: LDS of the wave, room for one chunk of column
// sums.  Every reservation is a whole number of `quantum` positions (EndArgs::quantum) and starts at one; a quantum
// holds whole pixels and lies in one run of pixels, so its columns are consecutive items.  It is worked off in chunks
// of whole pixels, at most kChunkColumns columns each: the wave lists its chunks (position, columns) kListChunks at
// a time, reads kGroupChunks chunks per round (up to six loads per lane and chunk, 64 consecutive floats per load
// instruction, all in flight together: the wave pays the memory's latency once per group, and it has the registers
// now), parks one chunk at a time in LDS and lets lane (pixel, plane) add the pixel's columns.
constexpr uint32_t kChunkColumns = 128;  // 1536 B of column sums: fits the wave's prepared-column table
constexpr uint32_t kChunkLoads = 3u * kChunkColumns / 64u;
constexpr uint32_t kGroupChunks = 4;
constexpr uint32_t kListChunks = 48;     // two words each, in the 128 words of the split table
static_assert(kListChunks % kGroupChunks == 0u && 2u * kListChunks + 3u <= 128u, "");
__device__ __forceinline__ float load_written(const float * p)
{
  // written earlier by this wave: read past the vector cache, which may hold an older copy of the line
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wave_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Pixel sums of the chunk in `stage` (`columns` column sums from queue position `pos` on): the accumulate pass of
// each pixel and plane, 0 + column 0 + column 1 + ... as colsum_kernel adds them.  kS: s_row (0: any).
template <uint32_t kS>
__device__ __forceinline__ void sum_chunk(const EndArgs & E, const float * stage, uint32_t pos, uint32_t columns, uint32_t lane)
{
  uint32_t const s_row = kS != 0u ? kS : E.s_row;
  uint32_t const chains = 3u * (columns / s_row);  // sums of the chunk: a pixel's red, green, blue
  size_t const first_pixel = (size_t)(end_queue_item(E, pos) / s_row);
  for (uint32_t c = lane; c < chains; c += 64u)
  {
    uint32_t const pixel = c / 3u, plane = c - 3u * pixel;
    const float * v = stage + 3u * pixel * s_row + plane;
    float sum = 0.0f;
    if constexpr (kS != 0u)
    {
#pragma unroll 16
      for (uint32_t i = 0; i < kS; ++i)
        sum += v[3u * i];
    }
    else
      for (uint32_t i = 0; i < s_row; ++i)
        sum += v[3u * i];
    size_t const px = first_pixel + pixel;
    if (px < E.n_pixels)
      E.planes[(size_t)plane * E.n_pixels + px] = sum;
  }
}
// The listed chunks (list[2 k] = position, list[2 k + 1] = columns), kGroupChunks at a time.
template <uint32_t kS>
__device__ __forceinline__ void sum_chunks(const EndArgs & E, const uint32_t * list, uint32_t n, float * stage, uint32_t lane)
{
  for (uint32_t g = 0; g < n; g += kGroupChunks)
  {
    float v[kGroupChunks][kChunkLoads];
#pragma unroll
    for (uint32_t k = 0; k < kGroupChunks; ++k)
    {
#pragma unroll
      for (uint32_t j = 0; j < kChunkLoads; ++j)
        v[k][j] = 0.0f;
      if (g + k < n)
      {
        uint32_t const pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)list[2u * (g + k)]);
        uint32_t const columns = (uint32_t)__builtin_amdgcn_readfirstlane((int)list[2u * (g + k) + 1u]);
        if (pos < E.queue_items && columns <= kChunkColumns && columns <= E.queue_items - pos)
        {
          const float * const src = E.partial + 3u * (size_t)end_queue_item(E, pos);
#pragma unroll
          for (uint32_t j = 0; j < kChunkLoads; ++j)
            if (lane + 64u * j < 3u * columns)
              v[k][j] = load_written(src + lane + 64u * j);
        }
      }
    }
#pragma unroll
    for (uint32_t k = 0; k < kGroupChunks; ++k)
    {
      if (g + k < n)
      {
        uint32_t const pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)list[2u * (g + k)]);
        uint32_t const columns = (uint32_t)__builtin_amdgcn_readfirstlane((int)list[2u * (g + k) + 1u]);
#pragma unroll
        for (uint32_t j = 0; j < kChunkLoads; ++j)
          stage[lane + 64u * j] = v[k][j];
        wave_fence();
        if (pos < E.queue_items && columns <= kChunkColumns && columns <= E.queue_items - pos)
          sum_chunk<kS>(E, stage, pos, columns, lane);
        wave_fence();
      }
    }
  }
}
__device__ __forceinline__ void wave_sums(uint32_t * log, uint32_t entries, float * stage)
{
  EndArgs const E = read_end_args();
  uint32_t const lane = __lane_id();
  // every column sum and shared colour of this wave is in memory
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  {
    uint32_t const slot = log[kLogEntries + 2u * lane], first = log[kLogEntries + 2u * lane + 1u];
    if (slot != RT_INVALID && slot < E.n_items)
    {
      // rt_core.h fold_split: the owner's part, then the shared colours in increasing sy
      float * const c = E.partial + 3u * (size_t)slot;
      f3 sum = mk3(load_written(c), load_written(c + 1), load_written(c + 2));
      size_t const base = (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * E.share_depth;
      for (uint32_t sy = first; sy < E.s_row; sy += E.slice_count)
      {
        const float * const v = E.shared + 3u * (base + (E.s_row - 1u - sy));
        sum = sum + mk3(load_written(v), load_written(v + 1), load_written(v + 2));
      }
      store3(c, sum);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  wave_fence();
  uint32_t * const list = log + kLogEntries;  // the split table has done its work
  // whole pixels per chunk; a quantum is a whole number of pixels too, so chunks never reach across one
  uint32_t const chunk = kChunkColumns / E.s_row * E.s_row;
  uint32_t const quantum = E.quantum != 0u ? E.quantum : kItemBatch;
  uint32_t e = 0, off = 0;                    // the next columns to list: entry, position in its reservation
  for (;;)
  {
    if (lane == 0u)
    {
      uint32_t n = 0;
      while (e < entries && n < kListChunks)
      {
        uint32_t const rec = log[e];
        uint32_t const first_pos = rec & ~(kItemBatch - 1u), length = ((rec & (kItemBatch - 1u)) + 1u) * kItemBatch;
        while (off < length && n < kListChunks)
        {
          uint32_t const pos = first_pos + off;
          if (pos >= E.queue_items || chunk == 0u)
          {
            off = length;
            break;
          }
          uint32_t const columns = min(min(chunk, quantum - off % quantum), E.queue_items - pos);
          list[2u * n] = pos;
          list[2u * n + 1u] = columns;
          n += 1u;
          off += columns;
        }
        if (off >= length)
        {
          e += 1u;
          off = 0u;
        }
      }
      list[2u * kListChunks] = n;
      list[2u * kListChunks + 1u] = e;
      list[2u * kListChunks + 2u] = off;
    }
    wave_fence();
    uint32_t const n = (uint32_t)__builtin_amdgcn_readfirstlane((int)list[2u * kListChunks]);
    e = (uint32_t)__builtin_amdgcn_readfirstlane((int)list[2u * kListChunks + 1u]);
    off = (uint32_t)__builtin_amdgcn_readfirstlane((int)list[2u * kListChunks + 2u]);
    if (n == 0u)
      break;
    switch (E.s_row)
    {
    case 1: sum_chunks<1>(E, list, n, stage, lane); break;
    case 2: sum_chunks<2>(E, list, n, stage, lane); break;
    case 4: sum_chunks<4>(E, list, n, stage, lane); break;
    case 8: sum_chunks<8>(E, list, n, stage, lane); break;
    case 10: sum_chunks<10>(E, list, n, stage, lane); break;
    case 16: sum_chunks<16>(E, list, n, stage, lane); break;
    case 32: sum_chunks<32>(E, list, n, stage, lane); break;
    case 64: sum_chunks<64>(E, list, n, stage, lane); break;
    default: sum_chunks<0>(E, list, n, stage, lane); break;
    }
    wave_fence();
  }
}

// LDS layout: [path stacks, unless kForward][jitter table][resident: filter, camera table, exact records, materials, lights,
// prepared-column tables | tiled: one staging tile].
// kLibMath: see rt_core.h Env::kLibraryMath (only the kernels for any number of lights come in that form).
// kFused: the kernel takes the pixel sums itself (PixelRing; only the kForward kernels with the short exact sequences
// come in that form).
// kLog: the waves take the pixel sums when they have no more work (wave_sums; resident layout only, not with kFused).
template <bool kTiled, bool kForward, bool kOneLight, bool kLibMath = false, bool kFused = false, bool kLog = false>
__global__ __launch_bounds__(kBlock, (kForward && !kTiled) ? kWavesPerSimdForward : (kTiled && kForward ? RT_TILED_WAVES : kWavesPerSimd)) void trace_kernel(TraceArgs A)
{
  static_assert(!kLog || (!kTiled && !kFused), "the reservation log belongs to the resident layout");
  extern __shared__ __align__(16) unsigned char lds_raw[];
  f4 * stack = reinterpret_cast<f4 *>(lds_raw + (kLog ? kLogBlockBytes : 0u));  // [RT_MAX_DEPTH][kBlock]
  f4 * sph = stack + (kForward ? 0u : RT_MAX_DEPTH * kBlock);      // resident scene, or the staging tile
  LaneCounters cnt = {};
#if defined(RT_PROFILE_REGIONS)
  cnt.last_tick = __builtin_amdgcn_s_memtime();
  timeline_mark(0);
#endif
  // both layouts start with the jitter table
  float * jitter = reinterpret_cast<float *>(sph);
  for (uint32_t i = threadIdx.x; i < RT_JITTER_FLOATS; i += kBlock)
    jitter[i] = A.jitter_tab[i];
  sph += RT_JITTER_FLOATS / 4u;
  if constexpr (kTiled)
  {
    constexpr uint32_t kTile = kForward ? kTileSpheres : kTileSpheresStack;
    float * const cone = reinterpret_cast<float *>(sph + kTile);
    // (the kernels without path stacks have the LDS for it: the lanes' primary rays are handed to the block's first lanes)
    constexpr bool kCompact = kForward;
    BlockEnv<kLibMath, kFused, kCompact> env{A, sph, cone, kOneLight ? A.aux_cone : A.aux_rw, jitter, ItemPool(), PixelRing()};
    unsigned char * after = reinterpret_cast<unsigned char *>(cone + kTile);
    if constexpr (kCompact)
    {
      env.exchange = reinterpret_cast<RayExchange *>(after);
      after += sizeof(RayExchange);
    }
    if constexpr (kFused)
      env.ring.init(A, after + wave_in_block() * ring_bytes_per_wave(A.ring_slots));
    __syncthreads();
    trace_loop<kForward, false, kOneLight>(A, env, stack + threadIdx.x, (uint32_t)kBlock, cnt);
    if constexpr (kFused)
      env.ring.drain(A);
  }
  else
  {
    f4 * cam = sph + A.n_padded;
    f4 * exact = cam + A.n_padded;
    f4 * mats = exact + A.n_padded;
    f4 * lights = mats + RT_MAT_RECORDS * A.n_materials;
    uint32_t * smat = reinterpret_cast<uint32_t *>(lights + RT_LIGHT_RECORDS * A.n_lights);
    float * aux = reinterpret_cast<float *>(smat + ((A.n_padded + 3u) & ~3u));
    ColumnEntry * batch = reinterpret_cast<ColumnEntry *>(aux + ((A.n_padded + 3u) & ~3u));
    for (uint32_t i = threadIdx.x; i < A.n_padded; i += kBlock)
    {
      sph[i] = A.filter[i];
      cam[i] = A.camtab[i];
      exact[i] = A.spheres[i];
    }
    for (uint32_t i = threadIdx.x; i < RT_MAT_RECORDS * A.n_materials; i += kBlock)
      mats[i] = A.materials[i];
    for (uint32_t i = threadIdx.x; i < RT_LIGHT_RECORDS * A.n_lights; i += kBlock)
      lights[i] = A.lights[i];
    for (uint32_t i = threadIdx.x; i < A.n_padded; i += kBlock)
    {
      smat[i] = A.sphere_mat[i];
      aux[i] = kOneLight ? A.aux_cone[i] : A.aux_rw[i];
    }
    if constexpr (kLog)
    {
      wave_log()[kLogEntries + 2u * __lane_id()] = RT_INVALID;  // no column with shared samples yet
      if (threadIdx.x == 0u)
        park_end_args(A);
      zero_other_runs(A);
    }
    __syncthreads();
#if defined(RT_PROFILE_REGIONS)
    timeline_mark(1);
#endif
    // The waves trace at wave priority 1 (a wave starts at 0).  Measured, not derived: set here -- after the tables are
    // staged, in front of the loop -- every frame kind is 0.6-0.8 % faster, alone and in a stream of frames (1080p 16x16
    // in a stream 17.31 -> 17.19 ms, glass 1024x768 32x32 26.44 -> 26.30, 10x10 2.81 -> 2.79), although all the waves
    // of a launch and of the launch it overlaps sit at one priority either way; priority 3 gives the same.  Part of it
    // is that nothing is scheduled across this instruction (the kernel with the reservation logs then spills 71 scalar
    // registers instead of 77; a bare scheduling barrier here gives 0.16 %).  The same instruction at the top of the
    // kernel costs 0.4 %, lowering the priority again for waves that have run out of work costs short frames 1 %, and
    // raising only theirs costs long frames 0.3 %.
    __builtin_amdgcn_s_setprio(1);
    WaveEnv<kLibMath, kFused, kLog> env{A, sph, cam, exact, mats, lights, smat, aux, jitter, batch + wave_in_block() * kItemBatch, ItemPool(), PixelRing()};
    if constexpr (kFused)
      env.ring.init(A, reinterpret_cast<unsigned char *>(batch + (kBlock / 64u) * kItemBatch) + wave_in_block() * ring_bytes_per_wave(A.ring_slots));
    trace_loop<kForward, false, kOneLight>(A, env, stack + threadIdx.x, (uint32_t)kBlock, cnt);
    if constexpr (kFused)
      env.ring.drain(A);
    if constexpr (kLog)
      wave_sums(wave_log(), min(env.pool.refills, min(A.log_entries, kLogEntries)), reinterpret_cast<float *>(env.batch));
#if defined(RT_PROFILE_REGIONS)
    timeline_mark(2);
#endif
  }

  flush_counters(A, cnt);
}

// The same loop with rays walking the scene's uniform grid (RT_HIP_FLAG_GRID).  LDS: [path stacks, unless
// kForward][jitter table].
template <bool kForward, bool kOneLight, bool kLibMath = false, bool kFused = false>
__global__ __launch_bounds__(kBlock, kForward ? kWavesPerSimdForward : kWavesPerSimd) void trace_kernel_grid(TraceArgs A)
{
  extern __shared__ __align__(16) unsigned char lds_raw[];
  f4 * stack = reinterpret_cast<f4 *>(lds_raw);
  float * jitter = reinterpret_cast<float *>(stack + (kForward ? 0u : RT_MAX_DEPTH * kBlock));
  for (uint32_t i = threadIdx.x; i < RT_JITTER_FLOATS; i += kBlock)
    jitter[i] = A.jitter_tab[i];
  __syncthreads();
  LaneCounters cnt = {};
#if defined(RT_PROFILE_REGIONS)
  cnt.last_tick = __builtin_amdgcn_s_memtime();
#endif
  GridEnv<kLibMath, kFused> env{A, jitter, ItemPool(), PixelRing()};
  if constexpr (kFused)
    env.ring.init(A, reinterpret_cast<unsigned char *>(jitter + RT_JITTER_FLOATS) + wave_in_block() * ring_bytes_per_wave(A.ring_slots));
  trace_loop<kForward, true, kOneLight>(A, env, stack + threadIdx.x, (uint32_t)kBlock, cnt);
  if constexpr (kFused)
    env.ring.drain(A);
  flush_counters(A, cnt);
}

// Per-launch set-up in one small kernel: the primary-ray filter table -- per sphere, (centre - camera, filter
// bound) -- and the words a trace starts from zero (its counters and queue word, the split-record count).
// One thread per table entry, at least one block.
__global__ __launch_bounds__(256) void camtab_kernel(const f4 * __restrict__ spheres, const float * __restrict__ aux_rw,
                                                     f4 * __restrict__ camtab, uint32_t n_padded, f3 cam_pos, float col_delta,
                                                     unsigned long long * __restrict__ stats, uint32_t stats_words,
                                                     uint32_t * __restrict__ split_count)
{
  uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < stats_words)
    stats[i] = 0ull;
  if (i == 0u && split_count != nullptr)
    *split_count = 0u;
  if (i >= n_padded)
    return;
  f4 const s = spheres[i];
  float const rw = aux_rw[i];  // the radius widened by the filter margin
  float const dx = s.x - cam_pos.x, dy = s.y - cam_pos.y, dz = s.z - cam_pos.z;
  f4 e;
  e.x = dx;
  e.y = dy;
  e.z = dz;
  // With column masks the bound is lowered by |d| * col_delta (rounded up), see TraceArgs::col_delta.
  float const len = __builtin_sqrtf(dx * dx + dy * dy + dz * dz) * 1.000001f;
  e.w = filter_bound(dx, dy, dz, rw * rw) - len * col_delta;
  camtab[i] = e;
}

// After a trace with sample sharing: completes the column sums whose last samples other lanes traced.  One
// lane per record; the records of a launch are at most one per lane of its grid.
__global__ __launch_bounds__(256) void fold_kernel(float * __restrict__ partial, const float * __restrict__ shared,
                                                   const u4 * __restrict__ splits,
                                                   const uint32_t * __restrict__ split_count, uint32_t s_row,
                                                   uint32_t slice_count)
{
  uint32_t const n = *split_count;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    fold_split(partial, shared, splits[i], s_row, slice_count);
}

// The accumulate pass: per pixel, the column sums are added in increasing sx (the order Frame::ForEachRayImpl
// adds its outer loop, ray_tracing/frame.cpp:56).  partial: the column sums (interleaved RGB), s_row consecutive
// entries per pixel; planes: 3 planes of n floats.  One lane per pixel.
__global__ __launch_bounds__(256) void colsum_kernel(const float * __restrict__ partial, float * __restrict__ planes,
                                                     uint32_t n, uint32_t s_row, uint32_t own_stride, uint32_t own_first)
{
  uint32_t const stride = gridDim.x * blockDim.x;
  bool const wide = (s_row % 4u) == 0u;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride)
  {
    const float * c = partial + 3u * (size_t)p * s_row;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    // (own_stride > 1: slices by runs of kRunPixels pixels; the pixels of other slices' runs have no sums here)
    if (own_stride > 1u && (p / kRunPixels) % own_stride != own_first)
      ;
    else if (wide)
    {
      // a multiple of four columns: 48-byte groups, 16-byte aligned
      const float4 * v = reinterpret_cast<const float4 *>(c);
      for (uint32_t i = 0; i < s_row; i += 4u, v += 3)
      {
        float4 const a0 = v[0], a1 = v[1], a2 = v[2];
        r += a0.x; g += a0.y; b += a0.z;
        r += a0.w; g += a1.x; b += a1.y;
        r += a1.z; g += a1.w; b += a2.x;
        r += a2.y; g += a2.z; b += a2.w;
      }
    }
    else
      for (uint32_t i = 0; i < s_row; ++i, c += 3)
      {
        r += c[0];
        g += c[1];
        b += c[2];
      }
    planes[p] = r;
    planes[n + p] = g;
    planes[2u * (size_t)n + p] = b;
  }
}

// The same pass for the common sample counts, through LDS: a lane of colsum_kernel reads its pixel's 12 * S bytes
// 16 at a time, so one load instruction of a wave touches 64 places 12 * S bytes apart.  Here a wave reads the
// column sums of kPixels consecutive pixels as one contiguous run, 16 bytes per lane (whole lines per instruction),
// parks them in its own LDS region (rows padded to an odd length, so that the pixel-per-lane reads that follow hit
// 64 different banks), and lane p then adds pixel p's columns in increasing sx exactly as colsum_kernel does: same
// sums, bit for bit.
template <uint32_t kS, uint32_t kPixels>
__global__ __launch_bounds__(256) void colsum_lds_kernel(const float * __restrict__ partial, float * __restrict__ planes,
                                                         uint32_t n, uint32_t own_stride, uint32_t own_first)
{
  static_assert(kRunPixels % kPixels == 0u, "a group of pixels lies in one run");
  constexpr uint32_t kRow = 3u * kS;        // floats per pixel
  constexpr uint32_t kPad = kRow | 1u;      // LDS row length: odd
  static_assert((kPixels * kRow) % 4u == 0u, "a wave's run starts on a 16-byte boundary");
  __shared__ float lds[4][kPixels * kPad];
  uint32_t const wave = threadIdx.x / 64u, lane = threadIdx.x % 64u;
  float * const mine = lds[wave];
  uint32_t const groups = (n + kPixels - 1u) / kPixels;
  for (uint32_t group = blockIdx.x * 4u + wave; group < groups; group += gridDim.x * 4u)
  {
    uint32_t const p0 = group * kPixels;
    uint32_t const count = min(kPixels, n - p0);
    if (own_stride > 1u && (p0 / kRunPixels) % own_stride != own_first)
    {
      // another slice's run: zero sums
      if (lane < count)
      {
        uint32_t const p = p0 + lane;
        planes[p] = 0.0f;
        planes[n + p] = 0.0f;
        planes[2u * (size_t)n + p] = 0.0f;
      }
      continue;
    }
    uint32_t const total = count * kRow;
    const float * const src = partial + (size_t)p0 * kRow;
    for (uint32_t g = 4u * lane; g < total; g += 256u)
    {
      float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (g + 3u < total)
      {
        float4 const q = *reinterpret_cast<const float4 *>(src + g);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      }
      else
        for (uint32_t j = 0; g + j < total; ++j)
          v[j] = src[g + j];
#pragma unroll
      for (uint32_t j = 0; j < 4u; ++j)
      {
        uint32_t const e = g + j, pixel = e / kRow;
        if (e < total)
          mine[pixel * kPad + (e - pixel * kRow)] = v[j];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < count)
    {
      const float * c = mine + lane * kPad;
      float r = 0.0f, gsum = 0.0f, b = 0.0f;
#pragma unroll
      for (uint32_t i = 0; i < kS; ++i, c += 3)
      {
        r += c[0];
        gsum += c[1];
        b += c[2];
      }
      uint32_t const p = p0 + lane;
      planes[p] = r;
      planes[n + p] = gsum;
      planes[2u * (size_t)n + p] = b;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// Zeroes n floats (16-byte aligned start): 16 bytes per lane per store.  A job's column sums start from zero so that
// snapshots show unfinished pixels black; the runtime's own fill moves 0.85 TB/s on this part (0.47 ms for the
// 398 MB of a 1080p 16x16 frame), this one five times that.
__global__ __launch_bounds__(256) void zero_kernel(float * __restrict__ p, size_t n)
{
  size_t const quads = n / 4u;
  size_t const stride = (size_t)gridDim.x * blockDim.x;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += stride)
    reinterpret_cast<float4 *>(p)[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  size_t const t = quads * 4u + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n)
    p[t] = 0.0f;
}

// planes: 3 x n floats; rgb: n x 3 floats.  Each lane takes 4 pixels: three 16-byte loads, three 16-byte
// stores (48 contiguous bytes), so both sides stream in full lines.
__global__ __launch_bounds__(256) void resolve_kernel(const float * __restrict__ planes,
                                                      float * __restrict__ rgb, uint32_t n, float inv)
{
  uint32_t const quads = n / 4u;
  uint32_t const stride = gridDim.x * blockDim.x;
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += stride)
  {
    float4 const r = reinterpret_cast<const float4 *>(planes)[q];
    float4 const g = reinterpret_cast<const float4 *>(planes + n)[q];
    float4 const b = reinterpret_cast<const float4 *>(planes + 2u * (size_t)n)[q];
    float4 * out = reinterpret_cast<float4 *>(rgb + 12u * (size_t)q);
    out[0] = make_float4(r.x * inv, g.x * inv, b.x * inv, r.y * inv);
    out[1] = make_float4(g.y * inv, b.y * inv, r.z * inv, g.z * inv);
    out[2] = make_float4(b.z * inv, r.w * inv, g.w * inv, b.w * inv);
  }
  // tail (n not a multiple of 4)
  uint32_t const t = quads * 4u + blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n)
  {
    rgb[3u * (size_t)t + 0] = planes[t] * inv;
    rgb[3u * (size_t)t + 1] = planes[n + t] * inv;
    rgb[3u * (size_t)t + 2] = planes[2u * (size_t)n + t] * inv;
  }
}

// ---------------------------------------------------------------- host side

// The message of the last failing call on this thread, in a fixed buffer: reporting an error allocates nothing.
thread_local char g_error[512] = {0};

int fail(int code, const char * what, hipError_t e = hipSuccess)
{
  if (e != hipSuccess)
    std::snprintf(g_error, sizeof(g_error), "%s: %s", what, hipGetErrorString(e));
  else
    std::snprintf(g_error, sizeof(g_error), "%s", what);
  return code;
}

#define RT_CHECK(call)                                         \
  do                                                           \
  {                                                            \
    hipError_t const e_ = (call);                              \
    if (e_ != hipSuccess)                                      \
      return fail(RT_HIP_ERR_DEVICE, #call, e_);               \
  } while (0)

struct Library
{
  std::mutex mutex;
  bool ready = false;
  int device = 0;
  int compute_units = 0;
  int clock_khz = 0;
  size_t lds_per_block = 65536;  // what one block may ask for without opting in to more
  char name[256] = {0};
  hipStream_t stream = nullptr;       // trace launches of the job API ...
  hipStream_t stream_b = nullptr;     // ... which take turns with this one (a lower priority, so that HIP gives it a
                                      // hardware queue of its own): the first waves of a job's trace take over the
                                      // compute units the last waves of the job before it leave
  unsigned job_turn = 0;              // (under the mutex)
  hipStream_t copy_stream = nullptr;  // snapshots, so they never queue behind a running trace
  // What an rt_hip_trace_device call needs besides its arguments, one set per caller stream so that calls on
  // different streams may overlap: column sums + sample-sharing room, the camera table, counters.  Grow-only.
  struct Workspace
  {
    hipStream_t stream = nullptr;
    bool used = false;
    float * partial = nullptr;
    size_t partial_bytes = 0;
    f4 * camtab = nullptr;
    size_t camtab_entries = 0;
    unsigned long long * stats = nullptr;  // kStatsWordsAll words, for callers that pass no d_stats
    unsigned long long last_use = 0;       // for handing the least recently used set to a ninth stream
    int callers = 0;                       // calls that hold this set right now (under Library::mutex)
    std::mutex use;                        // held while a call grows the set and enqueues on it
  };
  unsigned long long workspace_clock = 0;
  static constexpr int kWorkspaces = 8;
  Workspace workspaces[kWorkspaces];
  // Device address of `scratch` on each device the library has been initialised on (nullptr: not resolved yet).
  unsigned char * static_scratch[kMaxDevices] = {nullptr};
  uint32_t * never_cancel = nullptr;  // a zero word for traces that cannot be cancelled
  float * jitter_tab = nullptr;       // RT_JITTER_FLOATS, see rt_core.h jitter_ball
  rt_hip_job * spare_job = nullptr;   // buffers of the last finished job, see free_job
};
Library g_lib;

}  // namespace

struct rt_hip_scene
{
  f4 * d_spheres = nullptr;  // exact records, n_padded
  f4 * d_filter = nullptr;   // filter records, n_padded
  float * d_aux_rw = nullptr;    // per sphere, the cone term of a shade pass: for any light ...
  float * d_aux_cone = nullptr;  // ... and for light 0 alone (TraceArgs::aux_rw, aux_cone)
  uint32_t * d_sphere_mat = nullptr;
  f4 * d_materials = nullptr;
  f4 * d_lights = nullptr;
  uint32_t n_spheres = 0, n_padded = 0, n_materials = 0, n_lights = 0;
  bool forward_ok = false;  // paths can be accumulated forward (rt_prep.h forward_accumulation_ok)
  ShadowCone cone0;         // shadow cone of light 0 (kernels specialised for one light take it as an argument)
  f3 centre = {0.0f, 0.0f, 0.0f};  // what the filter records are relative to (rt_prep.h flatten_scene)
  bool in_bounds = false;   // the sizes rt_core.h near_root and inv_sqrt_rn ask for: centres within 2^40 of the origin,
                            // radii and light vectors between 2^-40 and 2^40 long, roughness at most 2^40
  // Blocks per compute unit of the trace kernel this scene is traced with, asked of the runtime once per kernel
  // variant ([1]: the grid walk) and LDS size rather than on every launch.
  // ... together with the number of windows a wave may keep unfinished when the pixel sums are taken inside the
  // kernel (TraceArgs::ring_slots): as many as fit the LDS the blocks of a compute unit share without costing a
  // block; 0: none fit.  Filled under the mutex: one scene may be traced from several threads at once.
  struct Occupancy
  {
    size_t lds = ~(size_t)0;  // of the launch without a ring
    bool tiled = false;
    int per_cu = 0;
    uint32_t ring_slots = 0;
    bool log_fits = false;    // the waves' reservation logs fit beside the rest without costing a block per compute unit
  };
  mutable Occupancy occupancy[4];
  mutable std::mutex occupancy_mutex;
  // uniform grid for RT_HIP_FLAG_GRID traces (scenes of at least kGridMinSpheres spheres)
  bool has_grid = false;
  uint32_t * d_grid_start = nullptr;
  uint32_t * d_grid_items = nullptr;
  uint32_t * d_grid_big = nullptr;
  uint32_t grid_big_count = 0;
  uint32_t grid_dim[3] = {1, 1, 1};
  float grid_org[3] = {0, 0, 0};
  float grid_cell = 1.0f;
};

struct rt_hip_job
{
  rt_hip_trace_params params;
  size_t partial_size = 0;  // bytes behind d_partial
  size_t pixels = 0;        // pixels behind d_planes / d_rgb
  float * d_partial = nullptr;  // the slice's column sums, RGB interleaved, then sample-sharing room (trace_buffer_bytes)
  float * d_planes = nullptr;
  float * d_rgb = nullptr;  // the finished image: the accumulate and resolve passes follow the trace on its stream
  f4 * d_camtab = nullptr;  // this frame's camera table
  size_t camtab_entries = 0;
  unsigned long long * d_stats = nullptr;  // kStatsWords counters, the queue counter and the error word behind them
  hipEvent_t start = nullptr, stop = nullptr;  // directly around the trace kernel
  hipEvent_t done = nullptr;                   // after the resolve pass: d_rgb is complete
  // Snapshots of the running frame have an image of their own (the passes that produce d_rgb may run at the same
  // time) and a pinned staging buffer for the download; all three are made by the first snapshot.
  float * d_snap_planes = nullptr;
  float * d_snap_rgb = nullptr;
  float * h_snap = nullptr;
  uint32_t * h_cancel = nullptr;  // pinned host word the kernel polls; written by rt_hip_job_cancel
  bool cancelled = false;
  bool fused = false;  // the trace kernel took the pixel sums itself: d_partial holds no column sums
  bool snapshots_wait = false;  // a short frame whose column sums were not cleared: a snapshot is the finished image
};

namespace
{
// RT_HIP_FORCE_TILE=n (at most kTileSpheres) makes every scene take the staged path with tiles of n records;
// RT_HIP_FORCE_STACK (any value) makes every scene created afterwards keep its paths on the LDS stack.  Both
// are testing aids for exercising those paths on the small demo scenes.
uint32_t forced_tile()
{
  const char * v = std::getenv("RT_HIP_FORCE_TILE");
  if (v == nullptr)
    return 0u;
  uint32_t const n = (uint32_t)std::strtoul(v, nullptr, 10);
  return std::min(std::max(n, 1u), kTileSpheres);
}

// RT_HIP_NO_CONES (any value): scenes created while it is set intersect their shadow rays with every sphere instead
// of testing the cone that holds them first; for comparisons (same frames, same query counters).
bool cones_disabled() { return std::getenv("RT_HIP_NO_CONES") != nullptr; }

// RT_HIP_NO_SHARING (any value) switches sample sharing at the end of the queue off, for comparisons.
bool share_disabled() { return std::getenv("RT_HIP_NO_SHARING") != nullptr; }
// Slices by runs of pixels (RT_HIP_FLAG_PIXEL_RUNS) instead of by samples.
bool pixel_runs(const rt_hip_trace_params * p) { return (p->flags & RT_HIP_FLAG_PIXEL_RUNS) != 0u && p->slice_count > 1u; }

bool uses_grid(const rt_hip_scene * sc, uint32_t flags) { return (flags & RT_HIP_FLAG_GRID) != 0u && sc->has_grid; }

// LDS of the resident layout: path stacks (scenes whose paths cannot be accumulated forward), the jitter table,
// three records, a material index and a cone term per sphere, the materials and lights, one prepared-column table per wave.
size_t resident_lds_bytes(const rt_hip_scene * sc)
{
  return (sc->forward_ok ? 0u : sizeof(f4) * (size_t)(RT_MAX_DEPTH * kBlock)) + sizeof(float) * RT_JITTER_FLOATS +
         sizeof(f4) * (size_t)(3u * sc->n_padded + RT_MAT_RECORDS * sc->n_materials + RT_LIGHT_RECORDS * sc->n_lights) +
         (sizeof(uint32_t) + sizeof(float)) * (size_t)((sc->n_padded + 3u) & ~3u) + sizeof(ColumnEntry) * (size_t)(kBlock / 64) * kItemBatch;
}

// A scene is staged through LDS in tiles when it is too large to live there whole: too many records for the
// resident layout's limits, or (with the path stacks of a scene that needs them) more bytes than a block may have.
bool is_tiled(const rt_hip_scene * sc)
{
  return sc->n_padded > kResidentSpheres || sc->n_materials > kResidentMaterials || sc->n_lights > kResidentLights ||
         resident_lds_bytes(sc) > g_lib.lds_per_block || forced_tile() != 0u;
}
uint32_t tile_size(const rt_hip_scene * sc)
{
  uint32_t const most = sc->forward_ok ? kTileSpheres : kTileSpheresStack;
  return forced_tile() ? std::min(forced_tile(), most) : most;
}
bool uses_column_masks(const rt_hip_scene * sc)
{
  return !is_tiled(sc) && sc->n_padded > 0u && sc->n_padded <= kColumnMaskSpheres;
}

// Path stacks (scenes whose paths cannot be accumulated forward) + either the resident scene (filter, camera
// table, exact records, materials, lights, one prepared-column table per wave) or one staging tile.
// ring_slots: windows per wave whose column sums the kernel keeps (0: the column sums go to memory).
size_t lds_bytes(const rt_hip_scene * sc, uint32_t flags, uint32_t ring_slots)
{
  size_t bytes = (sc->forward_ok ? 0u : sizeof(f4) * (size_t)(RT_MAX_DEPTH * kBlock)) + sizeof(float) * RT_JITTER_FLOATS;
  if (uses_grid(sc, flags))
    ;
  else if (is_tiled(sc))
    bytes += (sizeof(f4) + sizeof(float)) * (size_t)(sc->forward_ok ? kTileSpheres : kTileSpheresStack) +
             (sc->forward_ok ? sizeof(RayExchange) : 0u);
  else
    bytes = resident_lds_bytes(sc);
  bytes += (size_t)(kBlock / 64) * ring_bytes_per_wave(ring_slots);
  return std::max<size_t>(bytes, 16u);
}

// The kernel for a scene: resident / tiled / grid walk, stack or forward accumulation; the forward kernels of
// the resident and the grid path also come specialised for scenes with exactly one light.
// wave_sums: the kernel whose waves take the pixel sums when they run out of work (resident layout, not with `fused`).
template <class Launch>
auto with_trace_kernel(const rt_hip_scene * sc, uint32_t flags, bool library_math, bool fused, bool wave_sums, Launch && go)
{
  bool const one = sc->n_lights == 1u;
  if (library_math)
  {
    // scenes or cameras outside the bounds of the short exact sequences: the general kernels, built with the
    // compiler's IEEE square root and division throughout
    if (uses_grid(sc, flags))
      return sc->forward_ok ? go(trace_kernel_grid<true, false, true>) : go(trace_kernel_grid<false, false, true>);
    if (is_tiled(sc))
      return sc->forward_ok ? go(trace_kernel<true, true, false, true>) : go(trace_kernel<true, false, false, true>);
    if (wave_sums)
      return sc->forward_ok ? go(trace_kernel<false, true, false, true, false, true>) : go(trace_kernel<false, false, false, true, false, true>);
    return sc->forward_ok ? go(trace_kernel<false, true, false, true>) : go(trace_kernel<false, false, false, true>);
  }
  if (uses_grid(sc, flags))
  {
    if (!sc->forward_ok)
      return go(trace_kernel_grid<false, false>);
    if (fused)
      return one ? go(trace_kernel_grid<true, true, false, true>) : go(trace_kernel_grid<true, false, false, true>);
    return one ? go(trace_kernel_grid<true, true>) : go(trace_kernel_grid<true, false>);
  }
  if (is_tiled(sc))
  {
    if (!sc->forward_ok)
      return go(trace_kernel<true, false, false>);
    if (fused)
      return one ? go(trace_kernel<true, true, true, false, true>) : go(trace_kernel<true, true, false, false, true>);
    return one ? go(trace_kernel<true, true, true>) : go(trace_kernel<true, true, false>);
  }
  if (!sc->forward_ok)
    return wave_sums ? go(trace_kernel<false, false, false, false, false, true>) : go(trace_kernel<false, false, false>);
  if (fused)
    return one ? go(trace_kernel<false, true, true, false, true>) : go(trace_kernel<false, true, false, false, true>);
  if (wave_sums)
    return one ? go(trace_kernel<false, true, true, false, false, true>) : go(trace_kernel<false, true, false, false, false, true>);
  return one ? go(trace_kernel<false, true, true>) : go(trace_kernel<false, true, false>);
}

// Which scenes have kernels that take the pixel sums themselves.
bool has_fused_kernel(const rt_hip_scene * sc, bool library_math) { return sc->forward_ok && !library_math; }

// The short exact sequences need the scene and the camera within 2^40 of the origin and a near plane between 2^-20
// and 2^40 (rt_core.h near_root, inv_sqrt_rn); otherwise, and with RT_HIP_LIBRARY_MATH set (any value, for
// comparisons), a frame is traced by the kernels built with the compiler's sequences.
bool needs_library_math(const rt_hip_scene * sc, const rt_hip_camera * camera)
{
  bool in_bounds = sc->in_bounds && camera->znear >= 0x1p-20f && camera->znear <= 0x1p40f;
  for (int k = 0; k < 3; ++k)
    in_bounds = in_bounds && std::fabs(camera->position[k]) <= 0x1p40f;
  return !in_bounds || std::getenv("RT_HIP_LIBRARY_MATH") != nullptr;
}

// Persistent grid: as many blocks as the chip holds at once (more would only queue behind them), but no
// more lanes than work items -- and fewer for small frames.  A lane traces its samples one after the other, a path
// of ten bounces is ten trips round the loop, and a trip takes as long as the waves that share the SIMD make it
// (13 us with four of them, 4 us for a wave on its own).  Where a full grid leaves a lane only a few dozen
// samples, the frame takes as long as its longest chain of trips, not as long as its work: fewer, faster waves
// win.  kSamplesPerLane is the measured break-even (41 spheres: 1024x768 2x2 0.72 -> 0.58 ms, 720x540 2x2 0.63 ->
// 0.41 ms, 1920x1080 2x2 1.04 -> 1.00 ms); never below one wave per SIMD.  Tiled scenes keep the full grid: their
// blocks move in step and the samples cost a hundred times as much.
constexpr unsigned long long kSamplesPerLane = 36;
// Blocks per compute unit of the trace kernel for this scene, and the ring it can afford (rt_hip_scene::Occupancy).
// RT_HIP_RING_SLOTS=n: at most n slots (experiments).
rt_hip_scene::Occupancy launch_shape(const rt_hip_scene * sc, uint32_t flags, bool library_math)
{
  size_t const lds = lds_bytes(sc, flags, 0u);
  uint32_t most = kRingMaxSlots;
  if (const char * v = std::getenv("RT_HIP_RING_SLOTS"))
    most = std::min<uint32_t>(most, (uint32_t)std::strtoul(v, nullptr, 10));
  std::lock_guard<std::mutex> lock(sc->occupancy_mutex);
  rt_hip_scene::Occupancy & cached = sc->occupancy[(uses_grid(sc, flags) ? 1 : 0) + (library_math ? 2 : 0)];
  if (cached.lds != lds || cached.tiled != is_tiled(sc) || cached.per_cu < 1)
  {
    auto blocks_with = [&](size_t bytes, bool fused, bool wave_sums = false) {
      int asked = 0;
      hipError_t const e = with_trace_kernel(sc, flags, library_math, fused, wave_sums, [&](auto kernel) {
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(&asked, kernel, kBlock, bytes);
      });
      return (e != hipSuccess || asked < 1) ? 0 : asked;
    };
    cached.lds = lds;
    cached.tiled = is_tiled(sc);
    cached.per_cu = std::max(blocks_with(lds, false), 1);
    cached.log_fits = !cached.tiled && !uses_grid(sc, flags) && lds + kLogBlockBytes <= g_lib.lds_per_block &&
                      blocks_with(lds + kLogBlockBytes, false, true) >= cached.per_cu;
    cached.ring_slots = 0u;
    for (uint32_t slots = kRingMaxSlots; has_fused_kernel(sc, library_math) && slots >= kRingMinSlots; --slots)
    {
      size_t const with_ring = lds_bytes(sc, flags, slots);
      if (with_ring <= g_lib.lds_per_block && blocks_with(with_ring, true) >= cached.per_cu)
      {
        cached.ring_slots = slots;
        break;
      }
    }
  }
  rt_hip_scene::Occupancy shape = cached;
  shape.ring_slots = std::min(shape.ring_slots, most);
  if (shape.ring_slots < kRingMinSlots)
    shape.ring_slots = 0u;
  return shape;
}

// The pixel sums of a frame are taken inside the trace kernel when the caller asks for it (RT_HIP_FLAG_KERNEL_SUMS; or
// RT_HIP_KERNEL_SUMS in the environment, any value, for comparisons), a window of the work queue holds whole pixels
// (samples_in_row divides kItemBatch) and the scene's launch has room for a ring.  Otherwise the kernel leaves column
// sums and the accumulate pass adds them: the faster of the two on this part (the kernels with the ring run 3-5 %
// longer, more than the pass they save), but 12 bytes of workspace per sub-sample column of the frame.
uint32_t frame_ring_slots(const rt_hip_scene * sc, const rt_hip_trace_params * p, bool library_math)
{
  uint32_t const s = p->samples_in_row;
  if ((p->flags & RT_HIP_FLAG_KERNEL_SUMS) == 0u && std::getenv("RT_HIP_KERNEL_SUMS") == nullptr)
    return 0u;
  if (s == 0u || s > kItemBatch || (kItemBatch % s) != 0u)
    return 0u;
  return launch_shape(sc, p->flags, library_math).ring_slots;
}

int persistent_blocks(const rt_hip_scene * sc, uint32_t flags, bool library_math, unsigned long long items,
                      unsigned long long samples, bool leave_room)
{
  int const per_cu = launch_shape(sc, flags, library_math).per_cu;
  long long blocks = (long long)per_cu * std::max(g_lib.compute_units, 1);
  if (leave_room && per_cu >= kWavesPerSimdForward)
    blocks -= kSnapshotReserveBlocks;
  long long const needed = ((long long)items + kBlock - 1) / kBlock;
  blocks = std::max(1ll, std::min(blocks, needed));
  if (!is_tiled(sc) || uses_grid(sc, flags))
  {
    long long const one_wave_per_simd = std::max(g_lib.compute_units, 1);
    long long const wanted = (long long)(samples / (kSamplesPerLane * kBlock));
    blocks = std::min(blocks, std::max(wanted, one_wave_per_simd));
  }
  if (const char * v = std::getenv("RT_HIP_MAX_BLOCKS"))  // experiments: cap on the grid
    blocks = std::max(1ll, std::min(blocks, (long long)std::strtol(v, nullptr, 10)));
  return (int)blocks;
}

// Items per queue reservation: about 1024 samples' worth (16 columns of 64 ... one column of 1024 samples),
// but few enough that every wave of a full grid still makes several reservations (load balance at the end of
// the frame), and always a multiple of the window.
// (`cost` scales the sample counts down for scenes whose samples cost more: without the grid every ray is tested
// against every sphere, so a sample against 10 000 spheres takes a hundred and fifty times what one against the
// 64 spheres these sizes were tuned on does, and a reservation sized in samples alone would keep a wave busy for
// tens of milliseconds after the queue has run dry.)
// `items`: positions in the launch's queue; `per_column`: samples the launch traces of a column.
uint32_t batch_items(unsigned long long items, unsigned long long per_column, int compute_units, unsigned long long cost)
{
  unsigned long long const waves = (unsigned long long)std::max(compute_units, 1) * 4u * kWavesPerSimd;
  unsigned long long k = (1024u / cost + kItemBatch * per_column - 1u) / (kItemBatch * per_column);
  // ... and never fewer than four windows where a sample costs little: every reservation is an atomic on one word
  // for the whole chip, and with one window per reservation (1080p 16x16: half a million of them in 19 ms) the
  // waves queue for it; four windows in a row also keep a wave on neighbouring pixels (measured: 19.2 -> 18.0 ms)
  if (cost == 1u)
    k = std::max<unsigned long long>(k, 4u);
  unsigned long long const balance = items / (waves * kItemBatch * 4u);
  k = std::max<unsigned long long>(1u, std::min(k, balance));
  return (uint32_t)(k * kItemBatch);
}

// Smallest reservation near the end of the queue: about 256 samples' worth (the counter word stays well below
// its limit even if every reservation were that small), at least one window.  A frame in a stream of frames
// (`pipelined`, RT_HIP_FLAG_PIPELINED) takes at least two where a sample costs little: the next frame's waves fill the
// chip as this frame's leave, so the end of the launch costs little, and single windows cost more in atomics and in
// neighbouring pixels lost than they gain (1/8 of the 1080p 16x16 frame in a stream: 2.42 ms against 2.49).  A frame
// traced on its own ends when its last wave does, and single windows end it sooner where a window is 1024 samples or
// more (16x16: 17.9 -> 17.3 ms, 1/8 of it 3.6 -> 3.0, 32x32 at 1024x768 26.5 -> 26.4; 10x10: 3.03 -> 3.14, so not there).
uint32_t batch_min(unsigned long long per_column, uint32_t batch, unsigned long long cost, bool pipelined)
{
  unsigned long long k = std::max<unsigned long long>(1u, (256u / cost + kItemBatch * per_column - 1u) / (kItemBatch * per_column));
  if (cost == 1u && (pipelined || kItemBatch * per_column < 1024u))
    k = std::max<unsigned long long>(k, 2u);
  return (uint32_t)std::min<unsigned long long>(k * kItemBatch, batch);
}

// The end of a launch: a wave that finds the queue used up spreads the samples its busy lanes have not started
// over its idle lanes (up to kShareDepth samples at the end of a column), so it finishes when its work does,
// not when its slowest column does.  That needs room behind the column sums: a 16-byte header with the record
// count, then one record and share_depth colours per lane of the grid.
constexpr uint32_t kShareDepth = 64;
uint32_t share_depth(const rt_hip_trace_params * p) { return std::min(p->samples_in_row, kShareDepth); }
size_t share_bytes_for(size_t lanes, uint32_t depth) { return 16u + lanes * (sizeof(u4) + 3u * sizeof(float) * depth); }
unsigned long long frame_items(const rt_hip_trace_params * p)
{
  return (unsigned long long)p->width * p->height * p->samples_in_row;
}
// Enough for any grid this frame can be traced with.
size_t share_bytes(const rt_hip_trace_params * p)
{
  unsigned long long const chip = (unsigned long long)std::max(g_lib.compute_units, 1) * 4u * kWavesPerSimdForward * 64u;
  size_t const lanes = (size_t)std::min(chip, (frame_items(p) + kBlock - 1u) / kBlock * kBlock);
  return share_bytes_for(lanes, share_depth(p));
}

// Column sums: 12 bytes per column of the frame, whatever the slice (every slice visits every column).
size_t partial_bytes(const rt_hip_trace_params * p) { return 3 * sizeof(float) * (size_t)frame_items(p); }

// A trace's work buffer: the column sums -- unless the kernel takes the pixel sums itself (`fused`) -- then
// (256-byte aligned) the room sample sharing needs.
size_t share_offset(const rt_hip_trace_params * p, bool fused) { return fused ? 0u : ((partial_bytes(p) + 255u) & ~(size_t)255u); }
size_t trace_buffer_bytes(const rt_hip_trace_params * p, bool fused) { return share_offset(p, fused) + share_bytes(p); }

int validate(const rt_hip_scene * scene, const rt_hip_camera * camera, const rt_hip_trace_params * p)
{
  if (!g_lib.ready)
    return fail(RT_HIP_ERR_INVALID, "rt_hip_init has not been called");
  if (scene == nullptr || camera == nullptr || p == nullptr)
    return fail(RT_HIP_ERR_INVALID, "null argument");
  if (p->width == 0 || p->height == 0 || p->samples_in_row == 0)
    return fail(RT_HIP_ERR_INVALID, "width, height and samples_in_row must be positive");
  if ((unsigned long long)p->width * p->height > 0x40000000ull)
    return fail(RT_HIP_ERR_INVALID, "frame larger than 2^30 pixels");
  if (p->samples_in_row > 4096u)
    return fail(RT_HIP_ERR_INVALID, "samples_in_row larger than 4096");
  if ((unsigned long long)p->width * p->height * p->samples_in_row >= 0xF0000000ull)
    return fail(RT_HIP_ERR_INVALID, "width * height * samples_in_row must stay below 0xF0000000");
  if (p->slice_count > 1u && p->slice_index >= p->slice_count)
    return fail(RT_HIP_ERR_INVALID, "slice_index must be below slice_count");
  if (p->slice_count > 65535u)
    return fail(RT_HIP_ERR_INVALID, "slice_count must stay below 65536");
  return RT_HIP_OK;
}

int launch_colsum(const float * d_partial, float * d_planes, const rt_hip_trace_params * p, hipStream_t stream,
                  bool beside_trace = false);

// What a trace writes to, besides the caller's stream.
struct TraceTarget
{
  float * work = nullptr;       // column sums (frames whose pixel sums the kernel does not take itself), then the room
  size_t work_bytes = 0;        // of sample sharing: trace_buffer_bytes; sharing is on when its room fits
  float * planes = nullptr;     // 3 x pixels: the slice's pixel sums
  f4 * camtab = nullptr;        // scene->n_padded entries; belongs to the caller's job or workspace, so one scene
                                // carries any number of traces at a time
  unsigned long long * stats = nullptr;  // kStatsWordsAll words: counters, queue word, error word
  bool want_stats = false;
  const uint32_t * cancel = nullptr;
  bool clear_first = false;     // a job: what snapshots read (pixel sums, or column sums) starts from zero
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;  // recorded directly around the trace kernel (and its fold pass)
  bool fused = false;           // out: the kernel took the pixel sums itself
  bool wave_sums = false;       // out: the kernel left column sums and its waves added them up as they ended
  rt_hip_launch_info * describe = nullptr;  // not null: nothing is enqueued, the launch is described here instead
  bool column_sums_cleared = false;  // out: what a snapshot reads (clear_first) was cleared
};

// Enqueues, on `stream`: the set-up kernel (camera table for this frame's camera, counters reset), the trace kernel and
// what completes the slice's pixel sums in t.planes -- nothing when the kernel takes them itself, else the fold pass of
// sample sharing and the accumulate pass over the column sums.
int launch_trace(const rt_hip_scene * scene, const rt_hip_camera * camera, const rt_hip_trace_params * p,
                 TraceTarget & t, hipStream_t stream)
{
  TraceArgs a;
  std::memset(&a, 0, sizeof(a));
  prepare_frame(*camera, *p, a);
  a.spheres = scene->d_spheres;
  a.filter = scene->d_filter;
  a.aux_rw = scene->d_aux_rw;
  a.aux_cone = scene->d_aux_cone;
  a.centre = scene->centre;
  a.camtab = t.camtab;
  a.sphere_mat = scene->d_sphere_mat;
  a.materials = scene->d_materials;
  a.lights = scene->d_lights;
  a.jitter_tab = g_lib.jitter_tab;
  a.n_spheres = scene->n_spheres;
  a.n_padded = scene->n_padded;
  a.tile_size = tile_size(scene);
  a.n_materials = scene->n_materials;
  a.n_lights = scene->n_lights;
  a.cone_axis = scene->cone0.axis;
  a.cone_cos = scene->cone0.cos_m;
  a.cone_sin = scene->cone0.sin_m;
  a.cone_inv_cos = scene->cone0.cos_m > 0.0f ? 1.0f / scene->cone0.cos_m : 1.0f;
  a.partial = t.work;
  a.planes = t.planes;
  a.n_pixels = p->width * p->height;
  a.n_items = p->width * p->height * p->samples_in_row;
  a.queue_items = a.n_items;
  if (pixel_runs(p))
  {
    // this slice's runs of kRunPixels pixels, every sample of them (include/rt_hip.h RT_HIP_FLAG_PIXEL_RUNS)
    set_pixel_runs(a, kRunPixels);
  }
  unsigned long long const per_column = std::max<uint32_t>(1u, a.s_row / a.slice_count);  // samples of a column traced here
  a.stats = t.want_stats ? t.stats : nullptr;
  a.queue_counter = reinterpret_cast<uint32_t *>(t.stats + kStatsWords);
  a.ring_errors = t.stats + kStatsWords + 1u;
  a.cancel = t.cancel ? t.cancel : g_lib.never_cancel;
  unsigned long long const cost = uses_grid(scene, p->flags) ? 1u : std::max<unsigned long long>(1u, scene->n_spheres / 64u);
  a.batch_items = batch_items(a.queue_items, per_column, g_lib.compute_units, cost);
  if (const char * v = std::getenv("RT_HIP_BATCH_ITEMS"))  // experiments: reservation size in items (a multiple of 64)
    a.batch_items = std::max<uint32_t>(kItemBatch, (uint32_t)std::strtoul(v, nullptr, 10) / kItemBatch * kItemBatch);
  // (RT_HIP_FLAG_PIPELINED once kept the reservations of a frame in a stream of frames large to the end; since
  // reservations are four windows, shrinking them at the end is better there too: 17.64 -> 17.50 ms per 1080p 16x16
  // frame, 2.93 -> 2.62 ms per 1/8 of one, 337 -> 329 ms per 3840x2160 32x32 frame.  What the flag still decides is
  // how small the last reservations get.)
  a.batch_min = batch_min(per_column, a.batch_items, cost, (p->flags & RT_HIP_FLAG_PIPELINED) != 0u);
  if (const char * v = std::getenv("RT_HIP_BATCH_MIN"))  // experiments: smallest reservation in items
    a.batch_min = std::min(a.batch_items, std::max<uint32_t>(kItemBatch, (uint32_t)std::strtoul(v, nullptr, 10) / kItemBatch * kItemBatch));

  if (uses_grid(scene, p->flags))
  {
    a.grid_start = scene->d_grid_start;
    a.grid_items = scene->d_grid_items;
    a.grid_big = scene->d_grid_big;
    a.grid_big_count = scene->grid_big_count;
    for (int k = 0; k < 3; ++k)
      a.grid_dim[k] = scene->grid_dim[k];
    a.grid_org = mk3(scene->grid_org[0], scene->grid_org[1], scene->grid_org[2]);
    a.grid_cell = scene->grid_cell;
    a.grid_inv_cell = 1.0f / scene->grid_cell;
  }
  bool const library_math = needs_library_math(scene, camera);
  a.ring_slots = frame_ring_slots(scene, p, library_math);
  t.fused = a.ring_slots != 0u;
  if (t.fused)
    for (a.s_shift = 0u; (1u << a.s_shift) < a.s_row; ++a.s_shift)
      ;
  else if (t.work_bytes < partial_bytes(p))
    return fail(RT_HIP_ERR_INVALID, "internal: work buffer smaller than the frame's column sums");
  a.column_masks = (uses_column_masks(scene) && !uses_grid(scene, p->flags)) ? 1u : 0u;
  a.col_delta = a.column_masks ? column_delta(a) : 0.0f;
  bool const leave_room = (p->flags & RT_HIP_FLAG_LEAVE_ROOM) != 0u ||
                          (t.cancel != nullptr && (unsigned long long)a.queue_items * per_column >= kSnapshotRoomSamples);
  int const blocks = persistent_blocks(scene, p->flags, library_math, std::max<uint32_t>(a.queue_items, 1u),
                                       (unsigned long long)a.queue_items * per_column, leave_room);
  a.guide_waves = (uint32_t)blocks * (kBlock / 64u);
  // Pixel sums by the waves as they end (kLogEntries): the resident kernels, when a window holds whole pixels and the
  // waves' logs can hold the launch's reservations (rt_prep.h plan_wave_logs sizes them).
  // Only launches that fill the chip: beside a smaller grid (small frames, persistent_blocks) the passes find free
  // block slots and cost a stream of frames nothing, while the waves' own sums lengthen every wave a little.
  // RT_HIP_NO_WAVE_SUMS (any value): off, for comparisons.  RT_HIP_LOG_ENTRIES=n (tests): the waves' logs take n
  // reservations, and the launch goes ahead whatever the grid.
  t.wave_sums = false;
  uint32_t log_entries = kLogEntries;
  bool forced = false;
  if (const char * v = std::getenv("RT_HIP_LOG_ENTRIES"))
  {
    log_entries = std::min<uint32_t>(kLogEntries, std::max<uint32_t>(2u, (uint32_t)std::strtoul(v, nullptr, 10)));
    forced = true;
  }
  rt_hip_scene::Occupancy const shape = launch_shape(scene, p->flags, library_math);
  bool const full_grid = blocks >= shape.per_cu * std::max(g_lib.compute_units, 1) - kSnapshotReserveBlocks;
  bool const have_work = t.work != nullptr || t.describe != nullptr;  // (a description assumes a workspace of work_bytes)
  // A reservation has to hold whole pixels: its size is a multiple of the least common multiple of the window and
  // s_row (one window for 1, 2, 4 ... 64 samples per row, five for 10 or 20, three for 12), which must stay small,
  // and a chunk of kChunkColumns columns has to hold at least one pixel.
  // Where that makes the smallest reservation larger than it would be, only frames in a stream take the path: the
  // end of a frame traced on its own gets longer by more than the passes cost (1024x768 10x10, reservations of five
  // windows instead of two at the end: 3.47 ms against 3.09 alone, 2.80 against 2.96 in a stream).
  uint32_t const quantum_windows = a.s_row / std::gcd(a.s_row, kItemBatch);
  bool const coarser = quantum_windows * kItemBatch > a.batch_min;
  if (!t.fused && have_work && a.s_row <= kChunkColumns && quantum_windows <= kLogMaxQuantum && a.queue_items != 0u &&
      (!coarser || (p->flags & RT_HIP_FLAG_PIPELINED) != 0u || forced) &&
      shape.log_fits && (full_grid || forced) && std::getenv("RT_HIP_NO_WAVE_SUMS") == nullptr)
  {
    LogPlan const plan = plan_wave_logs(a.queue_items, a.guide_waves, a.batch_items, a.batch_min, log_entries, forced,
                                        kItemBatch, kLogMaxWindows, quantum_windows);
    if (plan.ok)
    {
      a.batch_items = plan.batch_items;
      a.batch_min = plan.batch_min;
      a.log_entries = log_entries;
      a.log_quantum = quantum_windows * kItemBatch;
      t.wave_sums = true;
    }
  }
  size_t const lds = lds_bytes(scene, p->flags, a.ring_slots) + (t.wave_sums ? kLogBlockBytes : 0u);
  // Sample sharing at the end of the queue, when the caller brought room for this grid's records and colours.
  size_t const lanes = (size_t)blocks * kBlock;
  size_t const share_at = share_offset(p, t.fused);
  if (have_work && t.work_bytes >= share_at + share_bytes_for(lanes, share_depth(p)) && !share_disabled())
  {
    a.share_depth = share_depth(p);
    if (t.work != nullptr)
    {
      unsigned char * const base = reinterpret_cast<unsigned char *>(t.work) + share_at;
      a.split_count = reinterpret_cast<uint32_t *>(base);
      a.splits = reinterpret_cast<u4 *>(base + 16);
      a.shared = reinterpret_cast<float *>(base + 16 + lanes * sizeof(u4));
      a.shared_off = (size_t)(a.shared - t.work);
    }
  }
  if (t.describe != nullptr)
  {
    rt_hip_launch_info & d = *t.describe;
    std::memset(&d, 0, sizeof(d));
    d.blocks = (uint32_t)blocks;
    d.lds_bytes = (uint32_t)lds;
    d.batch_items = a.batch_items;
    d.batch_min = a.batch_min;
    d.grid = uses_grid(scene, p->flags) ? 1u : 0u;
    d.tiled = (!d.grid && is_tiled(scene)) ? 1u : 0u;
    d.forward = scene->forward_ok ? 1u : 0u;
    d.library_math = library_math ? 1u : 0u;
    d.one_light = (scene->n_lights == 1u && scene->forward_ok && !library_math) ? 1u : 0u;
    d.column_masks = a.column_masks;
    d.sample_sharing = a.share_depth != 0u ? 1u : 0u;
    d.kernel_sums = t.fused ? 1u : 0u;
    d.wave_sums = t.wave_sums ? 1u : 0u;
    return RT_HIP_OK;
  }
  // What must start from zero: the sums of the pixels this slice does not trace (slices by runs of pixels: the kernel
  // writes its own pixels only) and whatever a snapshot may read before the kernel has written it.
  if (t.fused && (t.clear_first || pixel_runs(p)))
    RT_CHECK(hipMemsetAsync(t.planes, 0, 3 * sizeof(float) * (size_t)a.n_pixels, stream));
  auto zero_floats = [&](float * at, size_t floats) {
    int const fill_blocks = (int)std::min<size_t>((floats / 4u + 255u) / 256u + 1u, 256u * 8u);
    hipLaunchKernelGGL(zero_kernel, dim3(fill_blocks), dim3(256), 0, stream, at, floats);
  };
  if (!t.fused && t.clear_first)
  {
    // A frame too short for anyone to look at while it renders (kSnapshotRoomSamples) whose pixel sums the kernel's
    // waves take keeps its column sums uncleared: 398 MB for a 1080p 16x16 frame, a fill that would sit in front of
    // the trace on its stream and find block slots only as the previous job's trace leaves them.  A snapshot of such
    // a job waits for the frame (a tenth of a second at most) and is the finished image.
    bool const short_job = (unsigned long long)a.queue_items * per_column < kSnapshotRoomSamples;
    t.column_sums_cleared = !(t.wave_sums && short_job);
    if (t.column_sums_cleared)
    {
      zero_floats(t.work, partial_bytes(p) / sizeof(float));
      RT_CHECK(hipGetLastError());
    }
    // (a job may be cancelled: the pixels of windows no wave reserved keep the zero sums the accumulate pass would
    // have given them)
    if (t.wave_sums)
    {
      zero_floats(t.planes, 3u * (size_t)a.n_pixels);
      RT_CHECK(hipGetLastError());
    }
  }
  {
    int const tab_blocks = std::max(1, (int)((scene->n_padded + 255u) / 256u));
    hipLaunchKernelGGL(camtab_kernel, dim3(tab_blocks), dim3(256), 0, stream, scene->d_spheres, scene->d_aux_rw, t.camtab,
                       scene->n_padded, a.cam_pos, a.col_delta, t.stats, kStatsWordsAll, a.split_count);
    RT_CHECK(hipGetLastError());
  }
  if (t.ev_start)
    RT_CHECK(hipEventRecord(t.ev_start, stream));
  (void)with_trace_kernel(scene, p->flags, library_math, t.fused, t.wave_sums, [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kBlock), lds, stream, a);
    return hipSuccess;
  });
  RT_CHECK(hipGetLastError());
  if (!t.fused && !t.wave_sums && a.share_depth != 0u)
  {
    hipLaunchKernelGGL(fold_kernel, dim3(std::min(blocks, 256)), dim3(256), 0, stream, t.work, a.shared, a.splits,
                       a.split_count, a.s_row, a.slice_count);
    RT_CHECK(hipGetLastError());
  }
  if (t.ev_stop)
    RT_CHECK(hipEventRecord(t.ev_stop, stream));
  if (!t.fused && !t.wave_sums)
    return launch_colsum(t.work, t.planes, p, stream);
  return RT_HIP_OK;
}

// beside_trace: the pass is a snapshot's and has to run in the block slots a running trace kernel leaves free.
// Measured on a 3840x2160 job: the lane-per-pixel kernel does (12 ms per snapshot), the LDS kernel, 50 KB a block,
// waits for the frame's end.
int launch_colsum(const float * d_partial, float * d_planes, const rt_hip_trace_params * p, hipStream_t stream,
                  bool beside_trace)
{
  uint32_t const n = p->width * p->height;
  uint32_t const own_stride = pixel_runs(p) ? p->slice_count : 1u;
  uint32_t const own_first = pixel_runs(p) ? p->slice_index % p->slice_count : 0u;
  // the sample counts of the reference's demos and of the benchmark go through LDS (whole-line loads)
  auto through_lds = [&](auto kernel, uint32_t pixels_per_wave) {
    uint32_t const groups = (n + pixels_per_wave - 1u) / pixels_per_wave;
    int const blocks = (int)std::min<uint32_t>((groups + 3u) / 4u, 256u * 3u);
    hipLaunchKernelGGL(kernel, dim3(std::max(blocks, 1)), dim3(256), 0, stream, d_partial, d_planes, n, own_stride, own_first);
  };
  switch ((beside_trace || std::getenv("RT_HIP_PLAIN_COLSUM")) ? 0u : p->samples_in_row)
  {
  case 2: through_lds(colsum_lds_kernel<2, 64>, 64u); break;
  case 4: through_lds(colsum_lds_kernel<4, 64>, 64u); break;
  case 8: through_lds(colsum_lds_kernel<8, 64>, 64u); break;
  case 10: through_lds(colsum_lds_kernel<10, 64>, 64u); break;
  case 16: through_lds(colsum_lds_kernel<16, 64>, 64u); break;
  case 32: through_lds(colsum_lds_kernel<32, 32>, 32u); break;
  default:
  {
    int const blocks = (int)std::min<uint32_t>((n + 255u) / 256u, 256u * 8u);
    hipLaunchKernelGGL(colsum_kernel, dim3(blocks), dim3(256), 0, stream, d_partial, d_planes, n, p->samples_in_row, own_stride,
                       own_first);
  }
  }
  RT_CHECK(hipGetLastError());
  return RT_HIP_OK;
}

int launch_resolve(const float * d_planes, float * d_rgb, uint32_t width, uint32_t height, uint32_t s_row,
                   hipStream_t stream)
{
  uint32_t const n = width * height;
  float const inv = 1.0f / static_cast<float>(s_row * s_row);  // frame.cpp:43
  uint32_t const quads = std::max(n / 4u, 1u);
  int const blocks = (int)std::min<uint32_t>((quads + 255u) / 256u, 256u * 8u);
  hipLaunchKernelGGL(resolve_kernel, dim3(blocks), dim3(256), 0, stream, d_planes, d_rgb, n, inv);
  RT_CHECK(hipGetLastError());
  return RT_HIP_OK;
}

void read_stats(const unsigned long long * h, rt_hip_stats * out)
{
  out->samples = h[0];
  out->ray_queries = h[1];
  out->hit_queries = h[2];
  out->shaded_bounces = h[3];
  out->sphere_tests = h[4];
  out->exact_tests = h[5];
  out->lane_trips = h[6];
  out->lane_rounds = h[7];
  out->cone_tests = h[8];
}

void destroy_job(rt_hip_job * job)
{
  if (job == nullptr)
    return;
  if (job->d_partial) (void)hipFree(job->d_partial);
  if (job->d_planes) (void)hipFree(job->d_planes);
  if (job->d_rgb) (void)hipFree(job->d_rgb);
  if (job->d_camtab) (void)hipFree(job->d_camtab);
  if (job->d_snap_planes) (void)hipFree(job->d_snap_planes);
  if (job->d_snap_rgb) (void)hipFree(job->d_snap_rgb);
  if (job->h_snap) (void)hipHostFree(job->h_snap);
  if (job->done) (void)hipEventDestroy(job->done);
  if (job->d_stats) (void)hipFree(job->d_stats);
  if (job->h_cancel) (void)hipHostFree(job->h_cancel);
  if (job->start) (void)hipEventDestroy(job->start);
  if (job->stop) (void)hipEventDestroy(job->stop);
  delete job;
}

// A finished job's buffers are kept for the next frame of the same size (a viewer renders the same frame
// size over and over; allocating 400 MB and a pinned word per frame would cost more than a small frame).
void free_job(rt_hip_job * job)
{
  if (job == nullptr)
    return;
  // what only a snapshot needed goes back at once: a parked job holds the buffers every frame needs, no more
  if (job->d_snap_planes) (void)hipFree(job->d_snap_planes);
  if (job->d_snap_rgb) (void)hipFree(job->d_snap_rgb);
  if (job->h_snap) (void)hipHostFree(job->h_snap);
  job->d_snap_planes = job->d_snap_rgb = nullptr;
  job->h_snap = nullptr;
  rt_hip_job * evicted = job;
  {
    std::lock_guard<std::mutex> lock(g_lib.mutex);
    if (g_lib.ready && job->d_partial && job->d_planes && job->d_rgb && job->d_camtab && job->d_stats &&
        job->h_cancel && job->start && job->stop && job->done)
    {
      evicted = g_lib.spare_job;
      g_lib.spare_job = job;
    }
  }
  destroy_job(evicted);
}

// A job object with buffers for `params`: the spare one if it fits, else a fresh one.
rt_hip_job * acquire_job(const rt_hip_trace_params * params, size_t need, size_t camtab_entries, hipError_t & e)
{
  size_t const pixels = (size_t)params->width * params->height;
  camtab_entries = std::max<size_t>(camtab_entries, 1);
  rt_hip_job * job = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_lib.mutex);
    if (g_lib.spare_job != nullptr && g_lib.spare_job->partial_size >= need && g_lib.spare_job->pixels == pixels &&
        g_lib.spare_job->camtab_entries >= camtab_entries)
    {
      job = g_lib.spare_job;
      g_lib.spare_job = nullptr;
    }
  }
  e = hipSuccess;
  if (job != nullptr)
  {
    job->cancelled = false;
    job->params = *params;
    *job->h_cancel = 0u;
    return job;
  }
  job = new (std::nothrow) rt_hip_job;
  if (job == nullptr)
  {
    e = hipErrorOutOfMemory;
    return nullptr;
  }
  job->params = *params;
  job->partial_size = need;
  job->pixels = pixels;
  job->camtab_entries = camtab_entries;
  e = hipMalloc((void **)&job->d_partial, std::max<size_t>(need, 16));
  if (e == hipSuccess) e = hipMalloc((void **)&job->d_camtab, camtab_entries * sizeof(f4));
  if (e == hipSuccess) e = hipEventCreate(&job->done);
  if (e == hipSuccess) e = hipMalloc((void **)&job->d_planes, 3 * pixels * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void **)&job->d_rgb, 3 * pixels * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void **)&job->d_stats, sizeof(unsigned long long) * kStatsWordsAll);
  if (e == hipSuccess) e = hipEventCreate(&job->start);
  if (e == hipSuccess) e = hipEventCreate(&job->stop);
  if (e == hipSuccess) e = hipHostMalloc((void **)&job->h_cancel, 64, hipHostMallocMapped);
  if (e == hipSuccess)
    *job->h_cancel = 0u;
  else
  {
    destroy_job(job);
    job = nullptr;
  }
  return job;
}
}  // namespace

extern "C" {

// Everything the library owns on g_lib.device (that device must be current; g_lib.mutex held): streams, the spare
// job, the per-stream workspaces, the cancel word and the jitter table.  Scenes and running jobs belong to the
// caller.
static void release_device_state()
{
  if (g_lib.stream) (void)hipStreamSynchronize(g_lib.stream);
  if (g_lib.stream_b) (void)hipStreamSynchronize(g_lib.stream_b);
  if (g_lib.copy_stream) (void)hipStreamSynchronize(g_lib.copy_stream);
  if (g_lib.stream) (void)hipStreamDestroy(g_lib.stream);
  if (g_lib.stream_b) (void)hipStreamDestroy(g_lib.stream_b);
  if (g_lib.copy_stream) (void)hipStreamDestroy(g_lib.copy_stream);
  g_lib.stream = nullptr;
  g_lib.stream_b = nullptr;
  g_lib.copy_stream = nullptr;
  destroy_job(g_lib.spare_job);
  g_lib.spare_job = nullptr;
  for (Library::Workspace & w : g_lib.workspaces)
  {
    if (w.partial) (void)hipFree(w.partial);
    if (w.camtab) (void)hipFree(w.camtab);
    if (w.stats) (void)hipFree(w.stats);
    w.stream = nullptr;
    w.used = false;
    w.partial = nullptr;
    w.partial_bytes = 0;
    w.camtab = nullptr;
    w.camtab_entries = 0;
    w.stats = nullptr;
    w.last_use = 0;
  }
  if (g_lib.never_cancel) (void)hipFree(g_lib.never_cancel);
  if (g_lib.jitter_tab) (void)hipFree(g_lib.jitter_tab);
  g_lib.never_cancel = nullptr;
  g_lib.jitter_tab = nullptr;
  g_lib.ready = false;
}

// The address of the file-scope scratch array on the current device, once per device.
static int resolve_static_scratch(int device)
{
  if (device < 0 || device >= kMaxDevices)
    return RT_HIP_OK;  // rt_hip_trace_device_static reports the missing address when it is asked for
  if (g_lib.static_scratch[device] != nullptr)
    return RT_HIP_OK;
  void * ptr = nullptr;
  RT_CHECK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
  g_lib.static_scratch[device] = static_cast<unsigned char *>(ptr);
  return RT_HIP_OK;
}

int rt_hip_init(int device)
{
  std::lock_guard<std::mutex> lock(g_lib.mutex);
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(RT_HIP_ERR_DEVICE, "no HIP device available", e);
  if (device >= count)
    return fail(RT_HIP_ERR_INVALID, "device index out of range");
  int target = device;
  if (target < 0)
    RT_CHECK(hipGetDevice(&target));
  if (g_lib.ready && g_lib.device == target)
  {
    RT_CHECK(hipSetDevice(target));
    return RT_HIP_OK;
  }
  if (g_lib.ready)
  {
    // Another device: what was allocated on the old one is released there, not carried over (kernels on the new
    // device could not read it).
    RT_CHECK(hipSetDevice(g_lib.device));
    release_device_state();
  }
  if (g_lib.stream || g_lib.stream_b || g_lib.copy_stream || g_lib.never_cancel || g_lib.jitter_tab)
  {
    // an earlier rt_hip_init failed half way
    RT_CHECK(hipSetDevice(g_lib.device));
    release_device_state();
  }
  RT_CHECK(hipSetDevice(target));
  hipDeviceProp_t prop;
  RT_CHECK(hipGetDeviceProperties(&prop, target));
  g_lib.device = target;
  g_lib.compute_units = prop.multiProcessorCount;
  g_lib.clock_khz = prop.clockRate;
  // 64 KB is what a block gets without raising the kernel's dynamic-LDS attribute; the tiled layout with path
  // stacks (61 KB with tiles of kTileSpheresStack records) fits it, and a resident scene that would not is traced
  // through tiles instead.
  g_lib.lds_per_block = std::min<size_t>(prop.sharedMemPerBlock ? prop.sharedMemPerBlock : 65536u, 65536u);
  std::snprintf(g_lib.name, sizeof(g_lib.name), "%s (%s)", prop.name, prop.gcnArchName);
  RT_CHECK(hipStreamCreateWithFlags(&g_lib.stream, hipStreamNonBlocking));
  {
    int least = 0, greatest = 0;
    RT_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    // (no priorities on this device: the second stream is an ordinary one, and jobs simply take turns)
    if (hipStreamCreateWithPriority(&g_lib.stream_b, hipStreamNonBlocking, greatest) != hipSuccess)
      RT_CHECK(hipStreamCreateWithFlags(&g_lib.stream_b, hipStreamNonBlocking));
  }
  RT_CHECK(hipStreamCreateWithFlags(&g_lib.copy_stream, hipStreamNonBlocking));
  RT_CHECK(hipMalloc((void **)&g_lib.never_cancel, 64));
  RT_CHECK(hipMemset(g_lib.never_cancel, 0, 64));
  {
    float tab[RT_JITTER_FLOATS];
    build_jitter_table(tab);
    RT_CHECK(hipMalloc((void **)&g_lib.jitter_tab, sizeof(tab)));
    RT_CHECK(hipMemcpy(g_lib.jitter_tab, tab, sizeof(tab), hipMemcpyHostToDevice));
  }
  // (without the address only rt_hip_trace_device_static is lost, and it says so itself)
  (void)resolve_static_scratch(target);
  g_lib.ready = true;
  return RT_HIP_OK;
}

void rt_hip_shutdown(void)
{
  std::lock_guard<std::mutex> lock(g_lib.mutex);
  if (!g_lib.ready)
    return;
  (void)hipSetDevice(g_lib.device);
  release_device_state();
}

const char * rt_hip_last_error(void) { return g_error; }

int rt_hip_device_info(int * compute_units, int * clock_khz, char * name, size_t name_len)
{
  if (!g_lib.ready)
    return fail(RT_HIP_ERR_INVALID, "rt_hip_init has not been called");
  if (compute_units) *compute_units = g_lib.compute_units;
  if (clock_khz) *clock_khz = g_lib.clock_khz;
  if (name && name_len)
    std::snprintf(name, name_len, "%s", g_lib.name);
  return RT_HIP_OK;
}

int rt_hip_scene_create(const float * centers_radii, const uint32_t * material_index, uint32_t sphere_count,
                        const rt_hip_material * materials, uint32_t material_count,
                        const rt_hip_light * lights, uint32_t light_count, rt_hip_scene ** out_scene)
{
  if (out_scene == nullptr)
    return fail(RT_HIP_ERR_INVALID, "null argument");
  *out_scene = nullptr;
  if (!g_lib.ready)
    return fail(RT_HIP_ERR_INVALID, "rt_hip_init has not been called");
  if ((sphere_count && (!centers_radii || !material_index)) || (material_count && !materials) ||
      (light_count && !lights))
    return fail(RT_HIP_ERR_INVALID, "null array with a non-zero count");
  if (sphere_count > (1u << 24))
    return fail(RT_HIP_ERR_INVALID, "more than 2^24 spheres");
  for (uint32_t i = 0; i < sphere_count; ++i)
    if (material_index[i] >= material_count)
      return fail(RT_HIP_ERR_INVALID, "material index out of range");
  // The reference's range test lets a NaN distance through (ray_tracing/sphere.cpp:75), so a sphere with a NaN
  // in it "hits" wherever nothing nearer does; that is refused here rather than reproduced.
  for (size_t i = 0; i < (size_t)sphere_count * 4u; ++i)
    if (!std::isfinite(centers_radii[i]))
      return fail(RT_HIP_ERR_INVALID, "non-finite sphere centre or radius");
  for (uint32_t i = 0; i < material_count; ++i)
    if (materials[i].type > RT_HIP_MATERIAL_GLASS)
      return fail(RT_HIP_ERR_INVALID, "unknown material type");
  for (uint32_t i = 0; i < light_count; ++i)
    if (lights[i].type != RT_HIP_LIGHT_DIRECTIONAL)
      return fail(RT_HIP_ERR_INVALID, "unknown light type");

  FlatScene fs;
  flatten_scene(centers_radii, material_index, sphere_count, materials, material_count, lights, light_count, fs,
                !cones_disabled());

  rt_hip_scene * s = new (std::nothrow) rt_hip_scene;
  if (s == nullptr)
    return fail(RT_HIP_ERR_DEVICE, "out of host memory");
  s->n_spheres = sphere_count;
  s->n_padded = static_cast<uint32_t>(fs.spheres.size());
  s->n_materials = material_count;
  s->n_lights = light_count;
  s->centre = fs.centre;
  s->cone0 = shadow_cone(light_count ? lights[0].direction : nullptr, light_count != 0u && !cones_disabled());
  s->in_bounds = true;
  for (size_t k = 0; k < 4u * (size_t)sphere_count; ++k)
    s->in_bounds = s->in_bounds && std::fabs(centers_radii[k]) <= 0x1p40f && (k % 4u != 3u || std::fabs(centers_radii[k]) >= 0x1p-40f);
  for (uint32_t k = 0; k < material_count; ++k)
    s->in_bounds = s->in_bounds && std::fabs(materials[k].roughness) <= 0x1p40f;
  for (uint32_t k = 0; k < light_count; ++k)
  {
    double const len = std::sqrt((double)lights[k].direction[0] * lights[k].direction[0] + (double)lights[k].direction[1] * lights[k].direction[1] +
                                 (double)lights[k].direction[2] * lights[k].direction[2]);
    s->in_bounds = s->in_bounds && len >= 0x1p-40 && len <= 0x1p40;
  }
  s->forward_ok = forward_accumulation_ok(materials, material_count) && std::getenv("RT_HIP_FORCE_STACK") == nullptr;
  auto upload = [](void ** dst, const void * src, size_t bytes) -> hipError_t {
    // Never a zero-byte allocation: keep the pointer valid for empty arrays.
    hipError_t e = hipMalloc(dst, std::max<size_t>(bytes, 16));
    if (e == hipSuccess && bytes)
      e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    return e;
  };
  hipError_t e = upload((void **)&s->d_spheres, fs.spheres.data(), fs.spheres.size() * sizeof(f4));
  if (e == hipSuccess)
    e = upload((void **)&s->d_filter, fs.filter.data(), fs.filter.size() * sizeof(f4));
  if (e == hipSuccess)
    e = upload((void **)&s->d_aux_rw, fs.aux_rw.data(), fs.aux_rw.size() * sizeof(float));
  if (e == hipSuccess)
    e = upload((void **)&s->d_aux_cone, fs.aux_cone.data(), fs.aux_cone.size() * sizeof(float));
  if (e == hipSuccess)
    e = upload((void **)&s->d_sphere_mat, fs.sphere_mat.data(), fs.sphere_mat.size() * sizeof(uint32_t));
  if (e == hipSuccess)
    e = upload((void **)&s->d_materials, fs.materials.data(), fs.materials.size() * sizeof(f4));
  if (e == hipSuccess)
    e = upload((void **)&s->d_lights, fs.lights.data(), fs.lights.size() * sizeof(f4));
  if (e == hipSuccess && sphere_count >= kGridMinSpheres)
  {
    GridHost g;
    build_grid(centers_radii, sphere_count, g);
    if (g.valid)
    {
      e = upload((void **)&s->d_grid_start, g.start.data(), g.start.size() * sizeof(uint32_t));
      if (e == hipSuccess)
        e = upload((void **)&s->d_grid_items, g.items.data(), g.items.size() * sizeof(uint32_t));
      if (e == hipSuccess)
        e = upload((void **)&s->d_grid_big, g.big.data(), g.big.size() * sizeof(uint32_t));
      s->grid_big_count = static_cast<uint32_t>(g.big.size());
      for (int k = 0; k < 3; ++k)
      {
        s->grid_dim[k] = g.dim[k];
        s->grid_org[k] = g.org[k];
      }
      s->grid_cell = g.cell;
      s->has_grid = (e == hipSuccess);
    }
  }
  if (e != hipSuccess)
  {
    rt_hip_scene_destroy(s);
    return fail(RT_HIP_ERR_DEVICE, "scene upload", e);
  }
  *out_scene = s;
  return RT_HIP_OK;
}

void rt_hip_scene_destroy(rt_hip_scene * scene)
{
  if (scene == nullptr)
    return;
  if (scene->d_spheres) (void)hipFree(scene->d_spheres);
  if (scene->d_filter) (void)hipFree(scene->d_filter);
  if (scene->d_aux_rw) (void)hipFree(scene->d_aux_rw);
  if (scene->d_aux_cone) (void)hipFree(scene->d_aux_cone);
  if (scene->d_sphere_mat) (void)hipFree(scene->d_sphere_mat);
  if (scene->d_materials) (void)hipFree(scene->d_materials);
  if (scene->d_lights) (void)hipFree(scene->d_lights);
  if (scene->d_grid_start) (void)hipFree(scene->d_grid_start);
  if (scene->d_grid_items) (void)hipFree(scene->d_grid_items);
  if (scene->d_grid_big) (void)hipFree(scene->d_grid_big);
  delete scene;
}

int rt_hip_trace_async(const rt_hip_scene * scene, const rt_hip_camera * camera,
                       const rt_hip_trace_params * params, rt_hip_job ** out_job)
{
  if (out_job == nullptr)
    return fail(RT_HIP_ERR_INVALID, "null argument");
  *out_job = nullptr;
  int rc = validate(scene, camera, params);
  if (rc != RT_HIP_OK)
    return rc;

  hipError_t e = hipSuccess;
  bool const fused = frame_ring_slots(scene, params, needs_library_math(scene, camera)) != 0u;
  rt_hip_job * job = acquire_job(params, trace_buffer_bytes(params, fused), scene->n_padded, e);
  if (job == nullptr)
    return fail(RT_HIP_ERR_DEVICE, "job set-up", e);
  uint32_t * d_cancel = nullptr;
  e = hipHostGetDevicePointer((void **)&d_cancel, job->h_cancel, 0);
  if (e != hipSuccess)
  {
    free_job(job);
    return fail(RT_HIP_ERR_DEVICE, "job set-up", e);
  }
  TraceTarget t;
  t.work = job->d_partial;
  t.work_bytes = job->partial_size;
  t.planes = job->d_planes;
  t.camtab = job->d_camtab;
  t.stats = job->d_stats;
  t.want_stats = true;
  t.cancel = d_cancel;
  t.clear_first = true;  // unfinished pixels read as black in snapshots
  t.ev_start = job->start;
  t.ev_stop = job->stop;
  // The pixel sums and the resolve pass follow the trace on its stream, so the image is complete when `done` fires
  // and rt_hip_job_finish only has the download left -- on the copy stream, where it runs beside whatever frame
  // the trace streams have started next.  Successive jobs take turns on two streams (every job has buffers of its
  // own): a caller that starts the next frame before it finishes this one gets the end of one trace and the
  // beginning of the next side by side.
  hipStream_t stream = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_lib.mutex);
    stream = (g_lib.job_turn++ & 1u) != 0u ? g_lib.stream_b : g_lib.stream;
  }
  rc = launch_trace(scene, camera, params, t, stream);
  job->fused = t.fused;
  job->snapshots_wait = !t.fused && !t.column_sums_cleared;
  if (rc == RT_HIP_OK)
    rc = launch_resolve(job->d_planes, job->d_rgb, params->width, params->height, params->samples_in_row, stream);
  if (rc == RT_HIP_OK && (e = hipEventRecord(job->done, stream)) != hipSuccess)
    rc = fail(RT_HIP_ERR_DEVICE, "hipEventRecord", e);
  if (rc != RT_HIP_OK)
  {
    (void)hipStreamSynchronize(stream);
    free_job(job);
    return rc;
  }
  *out_job = job;
  return RT_HIP_OK;
}

int rt_hip_job_poll(rt_hip_job * job)
{
  if (job == nullptr)
    return fail(RT_HIP_ERR_INVALID, "null job");
  hipError_t const e = hipEventQuery(job->done);
  if (e == hipSuccess)
    return RT_HIP_OK;
  if (e == hipErrorNotReady)
    return RT_HIP_ERR_NOT_READY;
  return fail(RT_HIP_ERR_DEVICE, "hipEventQuery", e);
}

int rt_hip_job_snapshot(rt_hip_job * job, float * rgb)
{
  if (job == nullptr || rgb == nullptr)
    return fail(RT_HIP_ERR_INVALID, "null argument");
  size_t const pixels = (size_t)job->params.width * job->params.height;
  size_t const bytes = 3 * pixels * sizeof(float);
  if (job->snapshots_wait)
  {
    // a frame of well under a tenth of a second, traced without clearing what a snapshot would read: the snapshot
    // is the finished image
    RT_CHECK(hipEventSynchronize(job->done));
    RT_CHECK(hipMemcpyAsync(rgb, job->d_rgb, bytes, hipMemcpyDeviceToHost, g_lib.copy_stream));
    RT_CHECK(hipStreamSynchronize(g_lib.copy_stream));
    return RT_HIP_OK;
  }
  if (job->h_snap == nullptr)
  {
    // (the first snapshot of a job allocates: device-synchronising calls, made while the trace runs)
    hipError_t e = job->fused ? hipSuccess : hipMalloc((void **)&job->d_snap_planes, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&job->d_snap_rgb, bytes);
    if (e == hipSuccess) e = hipHostMalloc((void **)&job->h_snap, bytes, hipHostMallocDefault);
    if (e != hipSuccess)
    {
      if (job->d_snap_planes) (void)hipFree(job->d_snap_planes);
      if (job->d_snap_rgb) (void)hipFree(job->d_snap_rgb);
      job->d_snap_planes = job->d_snap_rgb = nullptr;
      job->h_snap = nullptr;
      return fail(RT_HIP_ERR_DEVICE, "snapshot buffers", e);
    }
  }
  // Resolve (after the accumulate pass, where the kernel leaves column sums) and download on the copy stream: they
  // run beside the trace kernel and never wait for it.  Pixels (columns) the trace has not finished yet still hold
  // the zeros they were cleared to.
  int rc = RT_HIP_OK;
  const float * sums = job->d_planes;
  if (!job->fused)
  {
    rc = launch_colsum(job->d_partial, job->d_snap_planes, &job->params, g_lib.copy_stream, true);
    sums = job->d_snap_planes;
  }
  if (rc == RT_HIP_OK)
    rc = launch_resolve(sums, job->d_snap_rgb, job->params.width, job->params.height,
                        job->params.samples_in_row, g_lib.copy_stream);
  if (rc != RT_HIP_OK)
    return rc;
  RT_CHECK(hipMemcpyAsync(job->h_snap, job->d_snap_rgb, bytes, hipMemcpyDeviceToHost, g_lib.copy_stream));
  RT_CHECK(hipStreamSynchronize(g_lib.copy_stream));
  std::memcpy(rgb, job->h_snap, bytes);
  return RT_HIP_OK;
}

int rt_hip_job_cancel(rt_hip_job * job)
{
  if (job == nullptr)
    return fail(RT_HIP_ERR_INVALID, "null job");
  job->cancelled = true;
  // The kernel reads this pinned word whenever a wave refills its pool of work items: lanes finish the
  // columns they hold (at most one batch per wave) and stop.  No GPU command is needed to deliver it.
  __atomic_store_n(job->h_cancel, 1u, __ATOMIC_RELEASE);
  return RT_HIP_OK;
}

int rt_hip_job_finish(rt_hip_job * job, float * rgb, rt_hip_stats * stats)
{
  if (job == nullptr)
    return fail(RT_HIP_ERR_INVALID, "null job");
  int rc = RT_HIP_OK;
  hipError_t e = hipEventSynchronize(job->done);
  if (e != hipSuccess)
    rc = fail(RT_HIP_ERR_DEVICE, "hipEventSynchronize", e);
  size_t const pixels = (size_t)job->params.width * job->params.height;
  if (rc == RT_HIP_OK && rgb != nullptr)
  {
    // The image is complete (`done`); its download does not queue behind the next frame's trace.
    e = hipMemcpyAsync(rgb, job->d_rgb, 3 * pixels * sizeof(float), hipMemcpyDeviceToHost, g_lib.copy_stream);
    if (e == hipSuccess)
      e = hipStreamSynchronize(g_lib.copy_stream);
    if (e != hipSuccess)
      rc = fail(RT_HIP_ERR_DEVICE, "image download", e);
  }
  if (rc == RT_HIP_OK && stats != nullptr)
  {
    unsigned long long h[kStatsWordsAll] = {0};
    std::memset(stats, 0, sizeof(*stats));
    e = hipMemcpyAsync(h, job->d_stats, sizeof(h), hipMemcpyDeviceToHost, g_lib.copy_stream);
    if (e == hipSuccess)
      e = hipStreamSynchronize(g_lib.copy_stream);
    if (e == hipSuccess && h[kStatsWords + 1u] != 0ull)
      rc = fail(RT_HIP_ERR_DEVICE, "internal: a wave gave up waiting for a free window slot; the image is incomplete");
    else if (e == hipSuccess)
    {
      read_stats(h, stats);
      float ms = 0.0f;
      if (hipEventElapsedTime(&ms, job->start, job->stop) == hipSuccess)
        stats->kernel_ms = ms;
    }
    else
      rc = fail(RT_HIP_ERR_DEVICE, "stats download", e);
  }
  if (rc == RT_HIP_OK && job->cancelled)
    rc = fail(RT_HIP_ERR_CANCELLED, "the job was cancelled");
  free_job(job);
  return rc;
}

int rt_hip_trace_device(const rt_hip_scene * scene, const rt_hip_camera * camera,
                        const rt_hip_trace_params * params, float * d_sum_planes, void * d_stats,
                        void * stream)
{
  int const rc = validate(scene, camera, params);
  if (rc != RT_HIP_OK)
    return rc;
  if (d_sum_planes == nullptr)
    return fail(RT_HIP_ERR_INVALID, "null output");
  hipStream_t const hs = static_cast<hipStream_t>(stream);
  // This stream's workspace: calls on one stream are ordered and share it, calls on different streams overlap.
  // The library's lock is held only while the set is picked and marked as held; growing it and enqueueing on it
  // happen under the set's own lock, so a call that has to wait for the device (a set that grows or changes
  // hands) holds up no call on another stream.
  Library::Workspace * ws = nullptr;
  bool takeover = false;
  {
    std::lock_guard<std::mutex> lock(g_lib.mutex);
    for (Library::Workspace & w : g_lib.workspaces)
      if (w.used && w.stream == hs)
        ws = &w;
    for (Library::Workspace & w : g_lib.workspaces)
      if (ws == nullptr && !w.used)
      {
        w.used = true;
        w.stream = hs;
        ws = &w;
      }
    if (ws == nullptr)
    {
      // every set belongs to some stream: the least recently used one that no call holds changes hands once the
      // device is idle (its old stream may no longer exist, so there is nothing narrower to wait for)
      for (Library::Workspace & w : g_lib.workspaces)
        if (w.callers == 0 && (ws == nullptr || w.last_use < ws->last_use))
          ws = &w;
      if (ws == nullptr)
        return fail(RT_HIP_ERR_INVALID, "rt_hip_trace_device: more than eight calls on different streams at the same time");
      ws->stream = hs;
      takeover = true;
    }
    ws->last_use = ++g_lib.workspace_clock;
    ws->callers += 1;
  }
  struct Release
  {
    Library::Workspace * ws;
    ~Release()
    {
      std::lock_guard<std::mutex> lock(g_lib.mutex);
      ws->callers -= 1;
    }
  } release{ws};
  std::lock_guard<std::mutex> use(ws->use);
  if (takeover)
    RT_CHECK(hipDeviceSynchronize());
  bool const fused = frame_ring_slots(scene, params, needs_library_math(scene, camera)) != 0u;
  size_t const need = trace_buffer_bytes(params, fused);
  if (ws->partial_bytes < need)
  {
    if (ws->partial != nullptr)
    {
      RT_CHECK(hipStreamSynchronize(hs));
      (void)hipFree(ws->partial);
      ws->partial = nullptr;
      ws->partial_bytes = 0;
    }
    RT_CHECK(hipMalloc((void **)&ws->partial, need));
    ws->partial_bytes = need;
  }
  if (ws->camtab_entries < std::max<size_t>(scene->n_padded, 1))
  {
    if (ws->camtab != nullptr)
    {
      RT_CHECK(hipStreamSynchronize(hs));
      (void)hipFree(ws->camtab);
      ws->camtab = nullptr;
      ws->camtab_entries = 0;
    }
    RT_CHECK(hipMalloc((void **)&ws->camtab, sizeof(f4) * std::max<size_t>(scene->n_padded, 1)));
    ws->camtab_entries = std::max<size_t>(scene->n_padded, 1);
  }
  // Without a caller buffer the counters (and the queue word behind them) live in the workspace.
  unsigned long long * st = static_cast<unsigned long long *>(d_stats);
  if (st == nullptr)
  {
    if (ws->stats == nullptr)
      RT_CHECK(hipMalloc((void **)&ws->stats, sizeof(unsigned long long) * kStatsWordsAll));
    st = ws->stats;
  }
  TraceTarget t;
  t.work = ws->partial;
  t.work_bytes = ws->partial_bytes;
  t.planes = d_sum_planes;
  t.camtab = ws->camtab;
  t.stats = st;
  t.want_stats = d_stats != nullptr;
  return launch_trace(scene, camera, params, t, hs);
}

int rt_hip_trace_device_static(const rt_hip_scene * scene, const rt_hip_camera * camera,
                               const rt_hip_trace_params * params, float * d_sum_planes, void * d_stats,
                               void * stream)
{
  int const rc = validate(scene, camera, params);
  if (rc != RT_HIP_OK)
    return rc;
  if (d_sum_planes == nullptr)
    return fail(RT_HIP_ERR_INVALID, "null output");
  unsigned char * const base =
      (g_lib.device >= 0 && g_lib.device < kMaxDevices) ? g_lib.static_scratch[g_lib.device] : nullptr;
  if (base == nullptr)
    return fail(RT_HIP_ERR_DEVICE, "the static scratch array has no address on this device");
  if (scene->n_padded > kStaticMaxSpheres)
    return fail(RT_HIP_ERR_INVALID, "rt_hip_trace_device_static takes scenes of at most 16384 spheres");
  // A frame whose pixel sums the kernel takes itself keeps no column sums, so it is not limited by the array: frames
  // whose column sums do not fit are traced that way where they can be.
  rt_hip_trace_params with_sums = *params;
  bool fused = frame_ring_slots(scene, params, needs_library_math(scene, camera)) != 0u;
  if (!fused && partial_bytes(params) > kStaticPartialBytes)
  {
    with_sums.flags |= RT_HIP_FLAG_KERNEL_SUMS;
    fused = frame_ring_slots(scene, &with_sums, needs_library_math(scene, camera)) != 0u;
    if (fused)
      params = &with_sums;
  }
  if (!fused && partial_bytes(params) > kStaticPartialBytes)
  {
    char msg[400];
    std::snprintf(msg, sizeof(msg),
                  "rt_hip_trace_device_static: width * height * samples_in_row is %llu, the limit is %llu (12 bytes per "
                  "sub-sample column in a 64 MiB static array) for frames whose pixel sums the trace kernel cannot take "
                  "itself (samples_in_row other than 1, 2, 4, 8, 16, 32, 64, or a scene that needs path stacks or the "
                  "general arithmetic); rt_hip_trace_device has no such limit",
                  frame_items(params), (unsigned long long)(kStaticPartialBytes / (3u * sizeof(float))));
    return fail(RT_HIP_ERR_INVALID, msg);
  }
  hipStream_t const hs = static_cast<hipStream_t>(stream);
  // Nothing in the array has to be cleared by the host: the set-up kernel zeroes the counters, the queue word and the
  // sharing record count on the stream, every column sum of the frame is written by exactly one lane, and the
  // camera table is rewritten for this frame.  Sample sharing is on when its room fits (behind the column sums).
  TraceTarget t;
  t.work = reinterpret_cast<float *>(base + kStaticPartialOffset);
  t.work_bytes = kStaticPartialBytes;
  t.planes = d_sum_planes;
  t.camtab = reinterpret_cast<f4 *>(base + kStaticCamtabOffset);
  t.stats = d_stats ? static_cast<unsigned long long *>(d_stats) : reinterpret_cast<unsigned long long *>(base);
  t.want_stats = d_stats != nullptr;
  return launch_trace(scene, camera, params, t, hs);
}

int rt_hip_resolve_device(const float * d_sum_planes, float * d_rgb, uint32_t width, uint32_t height,
                          uint32_t samples_in_row, void * stream)
{
  if (!g_lib.ready)
    return fail(RT_HIP_ERR_INVALID, "rt_hip_init has not been called");
  if (d_sum_planes == nullptr || d_rgb == nullptr || width == 0 || height == 0 || samples_in_row == 0)
    return fail(RT_HIP_ERR_INVALID, "bad argument");
  return launch_resolve(d_sum_planes, d_rgb, width, height, samples_in_row, static_cast<hipStream_t>(stream));
}

int rt_hip_describe_launch(const rt_hip_scene * scene, const rt_hip_camera * camera,
                           const rt_hip_trace_params * params, rt_hip_launch_info * out)
{
  int const rc = validate(scene, camera, params);
  if (rc != RT_HIP_OK)
    return rc;
  if (out == nullptr)
    return fail(RT_HIP_ERR_INVALID, "null argument");
  bool const fused = frame_ring_slots(scene, params, needs_library_math(scene, camera)) != 0u;
  TraceTarget t;
  t.work_bytes = trace_buffer_bytes(params, fused);  // the workspace rt_hip_trace_device would bring
  t.describe = out;
  return launch_trace(scene, camera, params, t, nullptr);
}

int rt_hip_time_trace_kernel(const rt_hip_scene * scene, const rt_hip_camera * camera,
                             const rt_hip_trace_params * params, uint32_t launches, float * out_avg_ms,
                             rt_hip_stats * out_stats)
{
  int rc = validate(scene, camera, params);
  if (rc != RT_HIP_OK)
    return rc;
  if (launches == 0 || out_avg_ms == nullptr)
    return fail(RT_HIP_ERR_INVALID, "bad argument");
  float * d_work = nullptr, * d_planes = nullptr;
  unsigned long long * d_stats = nullptr;
  f4 * d_camtab = nullptr;
  hipEvent_t start = nullptr, stop = nullptr;
  bool const fused = frame_ring_slots(scene, params, needs_library_math(scene, camera)) != 0u;
  size_t const work_bytes = trace_buffer_bytes(params, fused);
  hipError_t e = hipMalloc((void **)&d_work, work_bytes);
  if (e == hipSuccess) e = hipMalloc((void **)&d_planes, 3 * sizeof(float) * (size_t)params->width * params->height);
  if (e == hipSuccess) e = hipMalloc((void **)&d_camtab, sizeof(f4) * std::max<size_t>(scene->n_padded, 1));
  if (e == hipSuccess) e = hipMalloc((void **)&d_stats, sizeof(unsigned long long) * kStatsWordsAll);
  if (e == hipSuccess) e = hipEventCreate(&start);
  if (e == hipSuccess) e = hipEventCreate(&stop);
  double total_ms = 0.0;
  for (uint32_t i = 0; e == hipSuccess && rc == RT_HIP_OK && i < launches; ++i)
  {
    TraceTarget t;
    t.work = d_work;
    t.work_bytes = work_bytes;
    t.planes = d_planes;
    t.camtab = d_camtab;
    t.stats = d_stats;
    t.want_stats = true;
    t.ev_start = start;
    t.ev_stop = stop;
    rc = launch_trace(scene, camera, params, t, g_lib.stream);
    if (rc != RT_HIP_OK)
      break;
    e = hipStreamSynchronize(g_lib.stream);
    if (e != hipSuccess)
      break;
    float ms = 0.0f;
    e = hipEventElapsedTime(&ms, start, stop);
    total_ms += ms;
  }
  if (e == hipSuccess && rc == RT_HIP_OK)
  {
    *out_avg_ms = (float)(total_ms / launches);
    if (out_stats)
    {
      unsigned long long h[kStatsWords] = {0};
      std::memset(out_stats, 0, sizeof(*out_stats));
      e = hipMemcpy(h, d_stats, sizeof(h), hipMemcpyDeviceToHost);
      if (e == hipSuccess)
      {
        read_stats(h, out_stats);
        out_stats->kernel_ms = *out_avg_ms;
      }
    }
  }
  if (d_work) (void)hipFree(d_work);
  if (d_planes) (void)hipFree(d_planes);
  if (d_stats) (void)hipFree(d_stats);
  if (d_camtab) (void)hipFree(d_camtab);
  if (start) (void)hipEventDestroy(start);
  if (stop) (void)hipEventDestroy(stop);
  if (e != hipSuccess)
    return fail(RT_HIP_ERR_DEVICE, "rt_hip_time_trace_kernel", e);
  return rc;
}

}  // extern "C"

// ---- self-check of rt_core.h's short correctly rounded sequences against the compiler's, on this device ----
namespace
{
__device__ __forceinline__ bool same_float(float a, float b)
{
  return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);  // any NaN equals any NaN
}
__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
// A float of random sign and mantissa with its exponent in [-60, 60].
__device__ __forceinline__ float moderate_float(uint32_t h)
{
  uint32_t const exponent = 127u - 60u + (h >> 9) % 121u;
  return __uint_as_float((h & 0x80000000u) | (exponent << 23) | (mix32(h) & 0x007FFFFFu));
}
// out[0]: sqrt_rn_core, out[1]: inv_sqrt_rn, each over every float they are specified for; out[2]:
// div_rn_core on one pair per float pattern (half of the denominators within 2^-8 of 2, where the exact
// intersection's denominators lie); out[3]: near_root, short sequences against library_math, on discriminants and
// operands that include zeros, tiny, infinite and NaN discriminants and rays that start on the sphere.
__global__ __launch_bounds__(256) void exact_math_check_kernel(unsigned long long * out)
{
  unsigned long long bad[4] = {0ull, 0ull, 0ull, 0ull};
  unsigned long long const stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < (1ull << 32); k += stride)
  {
    uint32_t const i = (uint32_t)k;
    float const x = __uint_as_float(i);
    bool const tiny = (i & 0x7FFFFFFFu) >= 1u && (i & 0x7FFFFFFFu) < 0x0F800000u;  // of either sign
    if (!tiny && !same_float(sqrt_rn_core(x), sqrtf(x)))
      bad[0] += 1ull;
    bool const in_range = (i >= 0x0F800000u && i < 0x6F800000u) || x != x;
    if (in_range && !same_float(inv_sqrt_rn(x, false), 1.0f / sqrtf(x)))
      bad[1] += 1ull;
    uint32_t const h = mix32(i), g = mix32(i ^ 0x9E3779B9u);
    float const d = (h & 1u) ? __uint_as_float(0x40000000u + ((h >> 1) & 0xFFFFu) - 0x8000u) : moderate_float(h);
    float const n = ((g & 63u) == 0u) ? 0.0f : moderate_float(g);
    float const q = div_rn_core(n, d), q_ref = n / d;
    if (!(same_float(q, q_ref) || (n == 0.0f && q == 0.0f)))
      bad[2] += 1ull;
    // operands as the library's bounds leave them: |b| < 2^44, 0 <= disc < 2^90 or negative, or NaN for both
    float disc = fabsf(moderate_float(g)) * 0x1p29f;
    float b = moderate_float(mix32(h ^ g)) * 0x1p-17f;
    switch (i & 31u)
    {
    case 0: disc = 0.0f; break;
    case 1: disc = __uint_as_float(g & 0x0FFFFFFFu); break;   // denormals and tiny numbers
    case 2: disc = 0x1.fffffep89f; break;
    case 3: disc = b = __uint_as_float(0x7FC00000u); break;
    case 4: disc = -disc; break;
    case 5: disc = b * b; break;                               // a ray that starts on the sphere: -b - |b|
    case 6: disc = b * b * 0.99999f; break;
    default: break;
    }
    float const d2 = __uint_as_float(0x40000000u + ((h >> 1) & 0xFFu) - 0x80u);
    float const t = near_root(b, disc, d2, false), t_ref = near_root(b, disc, d2, true);
    // the contract: identical wherever the reference's own tests (disc < 0, t < tmin, t > tmax with tmin >= 2^-20)
    // let the distance through
    bool const used = !(disc < 0.0f) && !(t_ref < 0x1p-20f || t_ref > 0x1p59f);
    bool const used_short = !(disc < 0.0f) && !(t < 0x1p-20f || t > 0x1p59f);
    if (used != used_short || (used && !same_float(t, t_ref)))
      bad[3] += 1ull;
  }
  for (int c = 0; c < 4; ++c)
  {
    unsigned long long v = bad[c];
    for (int off = 32; off > 0; off >>= 1)
      v += __shfl_down(v, off);
    if (lane_here() == 0u && v != 0ull)
      atomicAdd(&out[c], v);
  }
}
}  // namespace

extern "C" int rt_hip_check_exact_math(uint64_t * mismatches4)
{
  if (!g_lib.ready)
    return fail(RT_HIP_ERR_INVALID, "rt_hip_init has not been called");
  if (mismatches4 == nullptr)
    return fail(RT_HIP_ERR_INVALID, "null argument");
  unsigned long long * d_out = nullptr;
  RT_CHECK(hipMalloc((void **)&d_out, 4 * sizeof(unsigned long long)));
  hipError_t e = hipMemsetAsync(d_out, 0, 4 * sizeof(unsigned long long), g_lib.stream);
  if (e == hipSuccess)
  {
    hipLaunchKernelGGL(exact_math_check_kernel, dim3(std::max(g_lib.compute_units, 1) * 8), dim3(256), 0, g_lib.stream, d_out);
    e = hipGetLastError();
  }
  unsigned long long host[4] = {0, 0, 0, 0};
  if (e == hipSuccess)
    e = hipMemcpyAsync(host, d_out, sizeof(host), hipMemcpyDeviceToHost, g_lib.stream);
  if (e == hipSuccess)
    e = hipStreamSynchronize(g_lib.stream);
  (void)hipFree(d_out);
  if (e != hipSuccess)
    return fail(RT_HIP_ERR_DEVICE, "rt_hip_check_exact_math", e);
  for (int c = 0; c < 4; ++c)
    mismatches4[c] = host[c];
  return RT_HIP_OK;
}

namespace
{
// The device's Philox4x32-10 (rt_core.h philox4x32_10: both halves of each product from one v_mad_u64_u32) on three
// known-answer inputs and, folded into one word by exclusive or, on a run of counters and keys.
constexpr uint32_t kPhiloxRun = 1u << 20;
__host__ __device__ inline void philox_run_input(uint32_t k, uint32_t (&in)[6])
{
  for (uint32_t j = 0; j < 6u; ++j)
    in[j] = (k + j) * 0x9E3779B1u ^ ((k >> (j + 1u)) * 0x85EBCA77u) ^ (j * 0xC2B2AE3Du);
}
__host__ __device__ inline uint32_t philox_fold(u4 o) { return o.x ^ (o.y * 3u) ^ (o.z * 5u) ^ (o.w * 7u); }
// Philox4x32-10 as published (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3"), on the host.
inline u4 philox_on_host(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
  for (int r = 0; r < 10; ++r)
  {
    unsigned long long const p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    uint32_t const n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
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
__global__ __launch_bounds__(256) void philox_check_kernel(uint32_t * out)
{
  uint32_t const t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < 3u)
  {
    // the known-answer vectors of the Random123 distribution: zeros, ones, digits of pi
    uint32_t const c[3][4] = {{0u, 0u, 0u, 0u}, {~0u, ~0u, ~0u, ~0u}, {0x243f6a88u, 0x85a308d3u, 0x13198a2eu, 0x03707344u}};
    uint32_t const k[3][2] = {{0u, 0u}, {~0u, ~0u}, {0xa4093822u, 0x299f31d0u}};
    u4 const o = philox4x32_10(c[t][0], c[t][1], c[t][2], c[t][3], k[t][0], k[t][1]);
    out[4u * t + 0u] = o.x;
    out[4u * t + 1u] = o.y;
    out[4u * t + 2u] = o.z;
    out[4u * t + 3u] = o.w;
  }
  uint32_t fold = 0u;
  for (uint32_t i = t; i < kPhiloxRun; i += gridDim.x * blockDim.x)
  {
    uint32_t in[6];
    philox_run_input(i, in);
    fold ^= philox_fold(philox4x32_10(in[0], in[1], in[2], in[3], in[4], in[5]));
  }
  atomicXor(&out[12], fold);
}
}  // namespace

extern "C" int rt_hip_check_philox(uint32_t * words12, uint32_t * run_device, uint32_t * run_host)
{
  if (!g_lib.ready)
    return fail(RT_HIP_ERR_INVALID, "rt_hip_init has not been called");
  if (words12 == nullptr || run_device == nullptr || run_host == nullptr)
    return fail(RT_HIP_ERR_INVALID, "null argument");
  uint32_t * d_out = nullptr;
  RT_CHECK(hipMalloc((void **)&d_out, 13 * sizeof(uint32_t)));
  hipError_t e = hipMemsetAsync(d_out, 0, 13 * sizeof(uint32_t), g_lib.stream);
  if (e == hipSuccess)
  {
    hipLaunchKernelGGL(philox_check_kernel, dim3(64), dim3(256), 0, g_lib.stream, d_out);
    e = hipGetLastError();
  }
  uint32_t host[13] = {0};
  if (e == hipSuccess)
    e = hipMemcpyAsync(host, d_out, sizeof(host), hipMemcpyDeviceToHost, g_lib.stream);
  if (e == hipSuccess)
    e = hipStreamSynchronize(g_lib.stream);
  (void)hipFree(d_out);
  if (e != hipSuccess)
    return fail(RT_HIP_ERR_DEVICE, "rt_hip_check_philox", e);
  for (int k = 0; k < 12; ++k)
    words12[k] = host[k];
  *run_device = host[12];
  // the same run through the routine as published, on the host
  uint32_t fold = 0u;
  for (uint32_t i = 0; i < kPhiloxRun; ++i)
  {
    uint32_t in[6];
    philox_run_input(i, in);
    fold ^= philox_fold(philox_on_host(in[0], in[1], in[2], in[3], in[4], in[5]));
  }
  *run_host = fold;
  return RT_HIP_OK;
}

#if defined(RT_PROFILE_REGIONS)
// Profiling builds only: copies out and clears the per-wave timeline of the last resident-kernel launch
// (four 100 MHz clock readings per wave: entry, staged, done, queue found used up).
extern "C" int rt_hip_debug_wave_times(unsigned long long * out, int waves)
{
  if (waves < 0 || (uint32_t)waves > kTimelineWaves)
    return RT_HIP_ERR_INVALID;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_times), sizeof(unsigned long long) * 4u * (size_t)waves) != hipSuccess)
    return RT_HIP_ERR_DEVICE;
  static unsigned long long zero[4 * kTimelineWaves] = {};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_wave_times), zero, sizeof(zero)) != hipSuccess)
    return RT_HIP_ERR_DEVICE;
  return RT_HIP_OK;
}

// Profiling builds only: reads (and clears) the per-region wave clock ticks summed over all waves.
extern "C" int rt_hip_debug_region_ticks(unsigned long long * out, int count)
{
  unsigned long long host[RT_REGION_COUNT] = {0};
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_region_ticks), sizeof(host)) != hipSuccess)
    return RT_HIP_ERR_DEVICE;
  for (int k = 0; k < count && k < RT_REGION_COUNT; ++k)
    out[k] = host[k];
  unsigned long long const zero[RT_REGION_COUNT] = {0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_region_ticks), zero, sizeof(zero)) != hipSuccess)
    return RT_HIP_ERR_DEVICE;
  return RT_HIP_OK;
}
#endif
