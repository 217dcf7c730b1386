// Issue cost of the vector instructions the trace kernels are made of, with four waves on every SIMD (one block of
// 1024 lanes per CU, held apart by its LDS allocation), as the trace kernels run.  Per instruction kind: a loop of
// eight independent instructions; reported: shader cycles (s_memtime) per instruction and SIMD, the median over the
// blocks, the clock the blocks saw (s_memtime against the 100 MHz s_memrealtime) and the nanoseconds per instruction
// and SIMD that follow.  argv[1]: blocks to launch (256 = the whole chip; fewer leave CUs idle, and the chip then
// holds its clock and issues FMAs faster).  The kernel's own duration is not the figure to read: blocks that the
// dispatcher puts on one CU run one after the other.
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

#define REP8(fmt) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)
#define FLOATS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define PAIRS "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
#define UINTS "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)
#define WIDES "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7)

template <int kKind>
__global__ void __launch_bounds__(1024) rate_kernel(unsigned long long * out, int iters, float seed)
{
  extern __shared__ float lds[];
  lds[threadIdx.x] = seed;
  __syncthreads();
  unsigned const lds_addr = (threadIdx.x / 64u) * 64u;   // wave-uniform: every lane reads the same words
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  float const m = 0.999f + seed * 1e-9f, c = 0.001f + seed * 1e-9f, e = 0.5f + seed * 1e-9f;
  f2 const pm = {m, m}, pc = {c, c};
  unsigned u0 = threadIdx.x * 2654435761u + 1u, u1 = u0 ^ 0x9e3779b9u, u2 = u0 + 77u, u3 = u1 + 99u, u4 = u0 * 3u, u5 = u1 * 5u, u6 = u2 * 7u, u7 = u3 * 9u;
  unsigned const k0 = 0xD2511F53u + (unsigned)iters, k1 = 0xCD9E8D57u + (unsigned)iters;
  unsigned long long w0 = u0, w1 = u1, w2 = u2, w3 = u3, w4 = u4, w5 = u5, w6 = u6, w7 = u7;
  f4v l0 = {0, 0, 0, 0}, l1 = {0, 0, 0, 0};
  f2 l2 = {0, 0};
  asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a0), "v"(a3) : "vcc");
  unsigned long long const t0 = __builtin_amdgcn_s_memtime();
  unsigned long long const r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (kKind == 0) {
#define F(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
      asm volatile(REP8(F) : FLOATS : "v"(m), "v"(c));
#undef F
    } else if constexpr (kKind == 1) {
#define F(i) "v_fmac_f32_e32 %" #i ", %8, %9\n"
      asm volatile(REP8(F) : FLOATS : "v"(m), "v"(c));
#undef F
    } else if constexpr (kKind == 2) {
#define F(i) "v_add_f32_e32 %" #i ", %8, %" #i "\n"
      asm volatile(REP8(F) : FLOATS : "v"(c));
#undef F
    } else if constexpr (kKind == 3) {
#define F(i) "v_mul_f32_e32 %" #i ", %8, %" #i "\n"
      asm volatile(REP8(F) : FLOATS : "v"(m));
#undef F
    } else if constexpr (kKind == 4) {
#define F(i) "v_fma_f32 %" #i ", %" #i ", |%8|, %9\n"
      asm volatile(REP8(F) : FLOATS : "v"(m), "v"(c));
#undef F
    } else if constexpr (kKind == 5) {
#define F(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 31\n"
      asm volatile(REP8(F) : UINTS : "v"(a0));
#undef F
    } else if constexpr (kKind == 6) {
#define F(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
      asm volatile(REP8(F) : FLOATS : "s"(seed), "v"(c));
#undef F
    } else if constexpr (kKind == 7) {
#define F(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
      asm volatile(REP8(F) : UINTS : "v"(k0));
#undef F
    } else if constexpr (kKind == 8) {
#define F(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
      asm volatile(REP8(F) : UINTS : "v"(k0));
#undef F
    } else if constexpr (kKind == 9) {
#define F(i) "v_mad_u64_u32 %" #i ", vcc, %8, %9, %" #i "\n"
      asm volatile(REP8(F) : WIDES : "v"(k0), "v"(k1) : "vcc");
#undef F
    } else if constexpr (kKind == 10) {
#define F(i) "v_rcp_f32_e32 %" #i ", %" #i "\n"
      asm volatile(REP8(F) : FLOATS);
#undef F
    } else if constexpr (kKind == 11) {
#define F(i) "v_sqrt_f32_e32 %" #i ", %" #i "\n"
      asm volatile(REP8(F) : FLOATS);
#undef F
    } else if constexpr (kKind == 12) {
#define F(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
      asm volatile(REP8(F) : PAIRS : "v"(pm), "v"(pc));
#undef F
    } else if constexpr (kKind == 13) {
#define F(i) "v_xor_b32_e32 %" #i ", %8, %" #i "\n"
      asm volatile(REP8(F) : UINTS : "v"(k0));
#undef F
    } else if constexpr (kKind == 14) {
#define F(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x96\n"
      asm volatile(REP8(F) : UINTS : "v"(k0), "v"(k1));
#undef F
    } else if constexpr (kKind == 15) {
#define F(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n"
      asm volatile(REP8(F) : FLOATS : "v"(m) : "vcc");
#undef F
    } else if constexpr (kKind == 16) {
#define F(i) "v_cmp_lt_f32_e32 vcc, %" #i ", %8\n"
      asm volatile(REP8(F) : FLOATS : "v"(m) : "vcc");
#undef F
    } else if constexpr (kKind == 17) {
#define F(i) "v_cmp_lt_f32_e64 s[20:21], %" #i ", %8\n"
      asm volatile(REP8(F) : FLOATS : "v"(m) : "s20", "s21");
#undef F
    } else if constexpr (kKind == 18) {
#define F(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
      asm volatile(REP8(F) : FLOATS : "v"(c), "v"(m));
#undef F
    } else if constexpr (kKind == 19) {
#define F(i) "v_ffbh_u32_e32 %" #i ", %" #i "\n"
      asm volatile(REP8(F) : UINTS);
#undef F
    } else if constexpr (kKind == 20) {
#define F(i) "v_lshlrev_b32_e32 %" #i ", 1, %" #i "\n"
      asm volatile(REP8(F) : UINTS);
#undef F
    } else if constexpr (kKind == 21) {
#define F(i) "v_add_u32_e32 %" #i ", %8, %" #i "\n"
      asm volatile(REP8(F) : UINTS : "v"(k0));
#undef F
    } else if constexpr (kKind == 22) {
#define F(i) "v_addc_co_u32_e32 %" #i ", vcc, %8, %" #i ", vcc\n"
      asm volatile(REP8(F) : UINTS : "v"(k0) : "vcc");
#undef F
    } else if constexpr (kKind == 23) {
#define F(i) "v_mov_b32_e32 %" #i ", %8\n"
      asm volatile(REP8(F) : FLOATS : "v"(m));
#undef F
    } else if constexpr (kKind == 24) {   // sources and destination all different registers
      asm volatile("v_fma_f32 %0, %1, %8, %9\n v_fma_f32 %1, %2, %8, %9\n v_fma_f32 %2, %3, %8, %9\n v_fma_f32 %3, %4, %8, %9\n"
                   "v_fma_f32 %4, %5, %8, %9\n v_fma_f32 %5, %6, %8, %9\n v_fma_f32 %6, %7, %8, %9\n v_fma_f32 %7, %0, %8, %9\n"
                   : FLOATS : "v"(m), "v"(c));
    } else if constexpr (kKind == 25) {
#define F(i) "v_fma_f32 %" #i ", %" #i ", 0.5, %8\n"
      asm volatile(REP8(F) : FLOATS : "v"(c));
#undef F
    } else if constexpr (kKind == 26) {
#define F(i) "v_lshl_or_b32 %" #i ", %" #i ", 1, %8\n"
      asm volatile(REP8(F) : UINTS : "v"(k0));
#undef F
    } else if constexpr (kKind == 27) {
#define F(i) "v_cvt_f32_u32_e32 %" #i ", %" #i "\n"
      asm volatile(REP8(F) : FLOATS);
#undef F
    } else if constexpr (kKind == 28) {
#define F(i) "v_max_f32_e32 %" #i ", %8, %" #i "\n"
      asm volatile(REP8(F) : FLOATS : "v"(c));
#undef F
    } else if constexpr (kKind == 29) {
#define F(i) "v_fmac_f32_e32 %" #i ", %8, %9\n"
      asm volatile(REP8(F) : FLOATS : "s"(seed), "v"(c));
#undef F
    } else if constexpr (kKind == 30) {
#define F(i) "v_rsq_f32_e32 %" #i ", %" #i "\n"
      asm volatile(REP8(F) : FLOATS);
#undef F
    } else if constexpr (kKind == 31) {
#define F(i) "v_sub_f32_e64 %" #i ", %" #i ", |%8|\n"
      asm volatile(REP8(F) : FLOATS : "v"(c));
#undef F
    } else if constexpr (kKind == 34) {
#define F(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
      asm volatile(REP8(F) : FLOATS : "v"(m) : "s20", "s21");
#undef F
    } else if constexpr (kKind == 35) {   // every other instruction a select
      asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_add_f32_e32 %1, %8, %1\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_add_f32_e32 %3, %8, %3\n"
                   "v_cndmask_b32_e32 %4, %4, %8, vcc\n v_add_f32_e32 %5, %8, %5\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_add_f32_e32 %7, %8, %7\n"
                   : FLOATS : "v"(m) : "vcc");
    } else if constexpr (kKind == 36) {   // one select in eight
      asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_add_f32_e32 %1, %8, %1\n v_add_f32_e32 %2, %8, %2\n v_add_f32_e32 %3, %8, %3\n"
                   "v_add_f32_e32 %4, %8, %4\n v_add_f32_e32 %5, %8, %5\n v_add_f32_e32 %6, %8, %6\n v_add_f32_e32 %7, %8, %7\n"
                   : FLOATS : "v"(m) : "vcc");
    } else if constexpr (kKind == 37) {   // a compare and a select that reads it, then six additions
      asm volatile("v_cmp_lt_f32_e32 vcc, %0, %8\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_add_f32_e32 %2, %8, %2\n v_add_f32_e32 %3, %8, %3\n"
                   "v_add_f32_e32 %4, %8, %4\n v_add_f32_e32 %5, %8, %5\n v_add_f32_e32 %6, %8, %6\n v_add_f32_e32 %7, %8, %7\n"
                   : FLOATS : "v"(m) : "vcc");
    } else if constexpr (kKind == 38) {   // select by arithmetic: max
#define F(i) "v_max_f32_e32 %" #i ", %8, %" #i "\n"
      asm volatile(REP8(F) : FLOATS : "v"(m));
#undef F
    } else if constexpr (kKind == 32 || kKind == 33) {
      // the shade filter's mix per 4 spheres: 4 ds_read_b128 + 2 ds_read2_b32 per 48 vector instructions (kind 33: without the reads)
      if constexpr (kKind == 32) {
        if (i % 3 == 0) asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n" : "=v"(l0), "=v"(l1) : "v"(lds_addr));
        if (i % 3 == 1) asm volatile("ds_read2_b32 %0, %1 offset1:1\n" : "=v"(l2) : "v"(lds_addr));
      }
#define F(i) "v_fmac_f32_e32 %" #i ", %8, %9\n"
      asm volatile(REP8(F) : FLOATS : "v"(m), "v"(c));
#undef F
      if constexpr (kKind == 32) {
        if (i % 3 == 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); asm volatile("" :: "v"(l0), "v"(l1), "v"(l2)); }
      }
    }
  }
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y +
            p6.x + p6.y + p7.x + p7.y + (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7) + (float)(w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7) + e;
  if (r == 12345.678f) out[0] = (unsigned long long)r;
  if (threadIdx.x == 0) {
    out[2 + 2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
    out[3 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

static int g_blocks = 256;   // one per CU; fewer: part of the chip idle (argv[1])

template <int kKind>
static void run(char const * name, unsigned long long * d_out)
{
  int const iters = 100000;
  int const blocks = g_blocks;
  size_t const lds_bytes = 100 * 1024;   // one block per CU
  static bool once[64];
  if (!once[kKind]) {
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&rate_kernel<kKind>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    once[kKind] = true;
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  rate_kernel<kKind><<<blocks, 1024, lds_bytes>>>(d_out, 100, 1.0f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  rate_kernel<kKind><<<blocks, 1024, lds_bytes>>>(d_out, iters, 1.0f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double const instr_per_simd = 8.0 * iters * 4;
  static unsigned long long ticks[2 + 2 * 256];
  CHECK(hipMemcpy(ticks, d_out, sizeof(ticks), hipMemcpyDeviceToHost));
  std::vector<double> cyc, mhz, ns;
  for (int b = 0; b < blocks; ++b) {
    cyc.push_back((double)ticks[2 + 2 * b]);
    ns.push_back((double)ticks[3 + 2 * b] * 10.0);
    mhz.push_back((double)ticks[2 + 2 * b] / ((double)ticks[3 + 2 * b] / 100.0));
  }
  std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end()); std::sort(ns.begin(), ns.end());
  printf("%-22s kernel %7.3f ms; blocks: %5.2f shader cycles per instruction per SIMD (median; min %5.2f, max %5.2f), clock %4.0f MHz, "
         "%.3f ns per instruction per SIMD (median; min %.3f, max %.3f)\n", name, ms,
         cyc[blocks / 2] / instr_per_simd, cyc[0] / instr_per_simd, cyc[blocks - 1] / instr_per_simd, mhz[blocks / 2],
         ns[blocks / 2] / instr_per_simd, ns[0] / instr_per_simd, ns[blocks - 1] / instr_per_simd);
}

int main(int argc, char ** argv)
{
  if (argc > 1) g_blocks = std::max(1, std::min(256, atoi(argv[1])));
  printf("%d blocks of 1024 lanes (4 waves per SIMD), one per CU\n", g_blocks);
  unsigned long long * d_out; CHECK(hipMalloc(&d_out, 8 * (2 + 2 * 256)));
  run<0>("v_fma_f32 (3 vgpr)", d_out);
  run<24>("v_fma_f32 d!=a", d_out);
  run<25>("v_fma_f32 inline 0.5", d_out);
  run<4>("v_fma_f32 |abs|", d_out);
  run<6>("v_fma_f32 sgpr", d_out);
  run<1>("v_fmac_f32_e32", d_out);
  run<29>("v_fmac_f32_e32 sgpr", d_out);
  run<2>("v_add_f32_e32", d_out);
  run<3>("v_mul_f32_e32", d_out);
  run<28>("v_max_f32_e32", d_out);
  run<31>("v_sub_f32_e64 |abs|", d_out);
  run<18>("v_med3_f32", d_out);
  run<12>("v_pk_fma_f32", d_out);
  run<5>("v_alignbit_b32", d_out);
  run<26>("v_lshl_or_b32", d_out);
  run<20>("v_lshlrev_b32_e32", d_out);
  run<13>("v_xor_b32_e32", d_out);
  run<14>("v_bitop3_b32", d_out);
  run<21>("v_add_u32_e32", d_out);
  run<22>("v_addc_co_u32", d_out);
  run<7>("v_mul_hi_u32", d_out);
  run<8>("v_mul_lo_u32", d_out);
  run<9>("v_mad_u64_u32", d_out);
  run<10>("v_rcp_f32", d_out);
  run<11>("v_sqrt_f32", d_out);
  run<30>("v_rsq_f32", d_out);
  run<27>("v_cvt_f32_u32", d_out);
  run<19>("v_ffbh_u32", d_out);
  run<23>("v_mov_b32", d_out);
  run<15>("v_cndmask_b32 vcc", d_out);
  run<34>("v_cndmask_b32 sgpr pair", d_out);
  run<35>("cndmask,add alternating", d_out);
  run<36>("1 cndmask + 7 add", d_out);
  run<37>("cmp, cndmask, 6 add", d_out);
  run<16>("v_cmp_lt_f32 vcc", d_out);
  run<17>("v_cmp_lt_f32 sgpr pair", d_out);
  run<33>("fmac, no LDS reads", d_out);
  run<32>("fmac + LDS reads", d_out);
  return 0;
}
