// microbench.hip -- issue rates of the VALU instructions the trace kernel leans on, measured on the
// device it runs on.  Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kIters = 4096;
constexpr int kChains = 8;

enum Op { FMA, PKFMA, MULHI, MULLO, MAD64, SQRT, RCP, RSQ, CMPSEL, PHILOX, ADD, EXP2, SINREV, CMPADDC, ALIGNBIT, FILT_OLD, FILT_NEW, SFMA, PERM, PKRTZ, ANDOR, LSHLOR, BFI, CNDMASK, MED3, MAX3, LSHR };

template <int OP>
__global__ __launch_bounds__(256) void bench(float * out, float seed)
{
  float x[kChains];
  f2 p[kChains];
  uint32_t u[kChains];
  for (int i = 0; i < kChains; ++i)
  {
    x[i] = seed + threadIdx.x * 1e-3f + i;
    p[i] = f2{x[i], x[i] + 1.0f};
    u[i] = (uint32_t)(threadIdx.x * 2654435761u + i);
  }
  float const a = seed * 0.5f + 1.0f, b = seed * 0.25f;
  for (int it = 0; it < kIters; ++it)
  {
#pragma unroll
    for (int i = 0; i < kChains; ++i)
    {
      if (OP == FMA) x[i] = __builtin_fmaf(x[i], a, b);
      if (OP == ADD) x[i] = x[i] + a;
      if (OP == PKFMA) p[i] = __builtin_elementwise_fma(p[i], f2{a, a}, f2{b, b});
      if (OP == MULHI) u[i] = __umulhi(u[i], 0xD2511F53u);
      if (OP == MULLO) u[i] = u[i] * 0xCD9E8D57u + 1u;
      if (OP == MAD64)
      {
        unsigned long long const m = (unsigned long long)u[i] * 0xD2511F53u;
        u[i] = (uint32_t)(m >> 32) ^ (uint32_t)m;
      }
      if (OP == SQRT) x[i] = __builtin_amdgcn_sqrtf(x[i]) + a;
      if (OP == RCP) x[i] = __builtin_amdgcn_rcpf(x[i]) + a;
      if (OP == RSQ) x[i] = __builtin_amdgcn_rsqf(x[i]) + a;
      if (OP == EXP2) x[i] = __builtin_amdgcn_exp2f(x[i]) * b;
      if (OP == SINREV) x[i] = __builtin_amdgcn_sinf(x[i]) + a;
      if (OP == CMPSEL) x[i] = (x[i] >= a && x[i] < b) ? x[i] * a : b;
      if (OP == ALIGNBIT) u[i] = __builtin_amdgcn_alignbit(u[i], __float_as_uint(x[i]), 31);
      if (OP == FILT_NEW)
      {
        float const t = __builtin_fmaf(x[i], a, __builtin_fmaf(x[i], b, __builtin_fmaf(x[i], seed, -b)));
        u[i] = __builtin_amdgcn_alignbit(u[i], __float_as_uint(t), 31);
      }
      if (OP == PERM) u[i] = __builtin_amdgcn_perm(u[i], __float_as_uint(a), 0x07030501u + (uint32_t)it);
      if (OP == PKRTZ)
      {
        auto h = __builtin_amdgcn_cvt_pkrtz(__uint_as_float(u[i]), a);
        u[i] = __builtin_bit_cast(uint32_t, h) + 0x3F800000u;
      }
      if (OP == ANDOR) u[i] = (u[i] & 0x80808080u) | (uint32_t)it;
      if (OP == LSHLOR) u[i] = (u[i] << 1) | (uint32_t)it;
      if (OP == BFI) u[i] = (u[i] & 0x0F0F0F0Fu) | (u[(i + 1) % kChains] & ~0x0F0F0F0Fu);
      if (OP == CNDMASK) x[i] = (u[i] & 1u) ? x[i] : x[(i + 1) % kChains];
      if (OP == MED3) x[i] = __builtin_amdgcn_fmed3f(x[i], a, b + (float)it);
      if (OP == MAX3) x[i] = fmaxf(fmaxf(x[i], a), b - (float)it);
      if (OP == LSHR) u[i] = (u[i] >> 1) + 0x9E3779B9u;
      if (OP == SFMA) x[i] = __builtin_fmaf(x[i], a, x[(i + 1) % kChains]);
      if (OP == PHILOX)
      {
        uint32_t c0 = u[i], c1 = it, c2 = i, c3 = 0, k0 = threadIdx.x, k1 = 7;
#pragma unroll
        for (int r = 0; r < 10; ++r)
        {
          uint32_t const hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
          uint32_t const hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
          c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
          k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        u[i] = c0 ^ c1 ^ c2 ^ c3;
      }
    }
    if (OP == CMPADDC || OP == FILT_OLD)
    {
      float t[kChains];
#pragma unroll
      for (int i = 0; i < kChains; ++i)
        t[i] = (OP == FILT_OLD) ? __builtin_fmaf(x[i], a, __builtin_fmaf(x[i], b, x[i] * seed)) : x[i];
      unsigned long long c0, c1, c2, c3, c4, c5, c6, c7;
      asm("v_cmp_ge_f32_e64 %[c0], %[x0], %[b]\n\t"
          "v_cmp_ge_f32_e64 %[c1], %[x1], %[b]\n\t"
          "v_cmp_ge_f32_e64 %[c2], %[x2], %[b]\n\t"
          "v_cmp_ge_f32_e64 %[c3], %[x3], %[b]\n\t"
          "v_cmp_ge_f32_e64 %[c4], %[x4], %[b]\n\t"
          "v_cmp_ge_f32_e64 %[c5], %[x5], %[b]\n\t"
          "v_cmp_ge_f32_e64 %[c6], %[x6], %[b]\n\t"
          "v_cmp_ge_f32_e64 %[c7], %[x7], %[b]\n\t"
          "v_addc_co_u32_e64 %[m0], %[c0], %[m0], %[m0], %[c0]\n\t"
          "v_addc_co_u32_e64 %[m1], %[c1], %[m1], %[m1], %[c1]\n\t"
          "v_addc_co_u32_e64 %[m2], %[c2], %[m2], %[m2], %[c2]\n\t"
          "v_addc_co_u32_e64 %[m3], %[c3], %[m3], %[m3], %[c3]\n\t"
          "v_addc_co_u32_e64 %[m4], %[c4], %[m4], %[m4], %[c4]\n\t"
          "v_addc_co_u32_e64 %[m5], %[c5], %[m5], %[m5], %[c5]\n\t"
          "v_addc_co_u32_e64 %[m6], %[c6], %[m6], %[m6], %[c6]\n\t"
          "v_addc_co_u32_e64 %[m7], %[c7], %[m7], %[m7], %[c7]"
          : [m0] "+v"(u[0]), [m1] "+v"(u[1]), [m2] "+v"(u[2]), [m3] "+v"(u[3]), [m4] "+v"(u[4]), [m5] "+v"(u[5]),
            [m6] "+v"(u[6]), [m7] "+v"(u[7]), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3),
            [c4] "=&s"(c4), [c5] "=&s"(c5), [c6] "=&s"(c6), [c7] "=&s"(c7)
          : [x0] "v"(t[0]), [x1] "v"(t[1]), [x2] "v"(t[2]), [x3] "v"(t[3]), [x4] "v"(t[4]), [x5] "v"(t[5]),
            [x6] "v"(t[6]), [x7] "v"(t[7]), [b] "v"(b));
    }
  }
  float acc = 0.0f;
  for (int i = 0; i < kChains; ++i)
    acc += x[i] + p[i].x + p[i].y + (float)u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
void run(const char * name, double ops_per_call, int blocks, float * d_out)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  double const lane_ops = (double)blocks * 256 * kIters * kChains * ops_per_call;
  printf("%-8s %8.3f ms  %8.2f Tlane-op/s\n", name, ms, lane_ops / (ms * 1e-3) / 1e12);
}

int main(int argc, char ** argv)
{
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  printf("device %s %s CUs %d clock %d kHz\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
  int const per_cu = argc > 1 ? atoi(argv[1]) : 8;  // blocks of 4 waves per CU = waves per SIMD
  int const blocks = prop.multiProcessorCount * per_cu;
  printf("waves per SIMD %d\n", per_cu);
  float * d_out;
  hipMalloc(&d_out, (size_t)blocks * 256 * sizeof(float));
  run<FMA>("fma", 1, blocks, d_out);
  run<ADD>("add", 1, blocks, d_out);
  run<PKFMA>("pk_fma", 2, blocks, d_out);
  run<MULHI>("mul_hi", 1, blocks, d_out);
  run<MULLO>("mul_lo", 1, blocks, d_out);
  run<MAD64>("mad64", 1, blocks, d_out);
  run<SQRT>("sqrt+add", 1, blocks, d_out);
  run<RCP>("rcp+add", 1, blocks, d_out);
  run<RSQ>("rsq+add", 1, blocks, d_out);
  run<EXP2>("exp2*mul", 1, blocks, d_out);
  run<SINREV>("sin+add", 1, blocks, d_out);
  run<CMPSEL>("cmpsel", 1, blocks, d_out);
  run<PHILOX>("philox", 1, blocks, d_out);
  run<CMPADDC>("cmp+addc", 2, blocks, d_out);
  run<ALIGNBIT>("alignbit", 1, blocks, d_out);
  run<FILT_OLD>("filt_old", 5, blocks, d_out);
  run<FILT_NEW>("filt_new", 4, blocks, d_out);
  run<SFMA>("fma_dep", 1, blocks, d_out);
  run<PERM>("perm", 1, blocks, d_out);
  run<PKRTZ>("pkrtz+xor", 2, blocks, d_out);
  run<ANDOR>("and_or", 1, blocks, d_out);
  run<LSHLOR>("lshl_or", 1, blocks, d_out);
  run<BFI>("bfi", 1, blocks, d_out);
  run<CNDMASK>("and+cnd", 2, blocks, d_out);
  run<MED3>("med3", 1, blocks, d_out);
  run<MAX3>("max3", 1, blocks, d_out);
  run<LSHR>("lshr+add", 2, blocks, d_out);
  hipFree(d_out);
  return 0;
}
