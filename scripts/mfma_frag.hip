// Dev helper (round 5): the MFMA + fragment-read stream of the plane GEMM's inner loop ALONE — operands already in LDS, no global
// traffic, no barriers, 512-thread blocks (two waves per SIMD as in wino4_gemm_kernel), one block per CU (forced by a 96 KB LDS
// allocation).  What is the ceiling of the loop itself, and does the MFMA shape / the wave tile move it?
//   P0  v_mfma_f32_16x16x4_f32, wave tile 80 x 32 (5 x 2 accumulator tiles): 7 ds_read_b128 per 40 MFMAs   (the shipped kernel)
//   P1  v_mfma_f32_16x16x4_f32, wave tile 80 x 64 (5 x 4): 9 reads per 80 MFMAs                               (0.1125 reads / MFMA)
//   P2  v_mfma_f32_32x32x2_f32, wave tile 64 x 32 (2 x 1 tiles of 32 x 32): 3 reads per 8 MFMAs (= 16 MFMA-equivalents of 32 cycles)
//   P3  v_mfma_f32_32x32x2_f32, wave tile 96 x 32 (3 x 1): 4 reads per 12 MFMAs (= 24 equivalents)
//   P4  v_mfma_f32_16x16x4_f32, 10 accumulator tiles, NO reads (register operands)                              (the pipe alone)
// Output: shader cycles per 32-cycle MFMA equivalent and per SIMD (32.0 = the matrix pipe never idles).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ float el(const float4 &v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

template <int P>
__global__ __launch_bounds__(512) void k(const float *__restrict__ data, float *out, int iters, unsigned long long *clk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // 24576 floats = 96 KB: [stage 0..3][192 rows][32]
  for (int i = threadIdx.x; i < 24576; i += blockDim.x) smem[i] = data[i & 4095];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr bool BIG = P == 2 || P == 3;
  constexpr int TM = P == 0 ? 5 : P == 1 ? 5 : P == 2 ? 2 : P == 3 ? 3 : 5;
  constexpr int TN = P == 0 ? 2 : P == 1 ? 4 : P == 2 ? 1 : P == 3 ? 1 : 2;
  f32x4 acc[BIG ? 1 : TN][BIG ? 1 : TM];
  f32x16 big[BIG ? TN : 1][BIG ? TM : 1];
  for (int j = 0; j < (BIG ? 1 : TN); ++j)
    for (int i = 0; i < (BIG ? 1 : TM); ++i) acc[j][i] = f32x4{0, 0, 0, 0};
  for (int j = 0; j < (BIG ? TN : 1); ++j)
    for (int i = 0; i < (BIG ? TM : 1); ++i)
      for (int r = 0; r < 16; ++r) big[j][i][r] = 0.f;
  // fragment addresses: row (tile * 16|32 + lane's row), 16-byte slot by lane group, XOR swizzle as in the kernel
  const int t16 = lane & 15, q = lane >> 4, t32 = lane & 31, h2 = lane >> 5;
  float4 wa0[4], xb0[5], wa1[4], xb1[5];
  const unsigned long long c0 = __builtin_readcyclecounter();
  if (P == 4) {
    float va[8], vb[8];
    for (int i = 0; i < 8; ++i) { va[i] = smem[(lane + 64 * i) & 4095]; vb[i] = smem[(lane + 64 * i + 2048) & 4095]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 5; ++i) acc[j][i] = MFMA16(va[(e + j) & 7], vb[(e + i) & 7], acc[j][i]);
      FENCE();
    }
  } else if (!BIG) {
#define FRAG16(STG, HH, WA, XB)                                                                                     \
  {                                                                                                                 \
    const float *base_ = smem + (STG) * 6144;                                                                       \
    const int slot_ = ((4 * (HH) + q) ^ ((t16 >> 1) & 7)) * 4;                                                      \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) WA[j] =                                                          \
        *reinterpret_cast<const float4 *>(base_ + (128 + 16 * j + t16) * 32 + slot_);   /* distinct rows per tile */   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) XB[i] =                                                          \
        *reinterpret_cast<const float4 *>(base_ + (16 * i + t16) * 32 + slot_);                                     \
  }
#define MF16(WA, XB, E)                                                                                             \
  { _Pragma("unroll") for (int j = 0; j < TN; ++j) _Pragma("unroll") for (int i = 0; i < TM; ++i) acc[j][i] =       \
        MFMA16(el(WA[j], E), el(XB[i], E), acc[j][i]); }
    FRAG16(0, 0, wa0, xb0)
    for (int it = 0; it < iters; ++it) {   // one iteration = one 32-channel chunk = two half-chunks of 4 MFMA steps
      const int st = it & 3;
      MF16(wa0, xb0, 0)
      FRAG16(st, 1, wa1, xb1)
      FENCE();
      MF16(wa0, xb0, 1) MF16(wa0, xb0, 2) MF16(wa0, xb0, 3)
      FRAG16((st + 1) & 3, 0, wa0, xb0)
      FENCE();
      MF16(wa1, xb1, 0) MF16(wa1, xb1, 1) MF16(wa1, xb1, 2) MF16(wa1, xb1, 3)
      FENCE();
      __builtin_amdgcn_s_waitcnt(0xC07F);
      FENCE();
    }
  } else {
#define FRAG32(STG, HH, WA, XB)                                                                                     \
  {                                                                                                                 \
    const float *base_ = smem + (STG) * 6144;                                                                       \
    /* row swizzle f(r) = ((r >> 1) & 7) ^ (((r >> 4) & 1) << 2): the four 16-lane groups of a wave hit four slots */ \
    const int slot_ = ((2 * (HH) + h2) ^ ((t32 >> 1) & 7) ^ (((t32 >> 4) & 1) << 2)) * 4;                           \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) WA[j] =                                                          \
        *reinterpret_cast<const float4 *>(base_ + (128 + 32 * j + t32) * 32 + slot_);                               \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) XB[i] =                                                          \
        *reinterpret_cast<const float4 *>(base_ + (32 * i + t32) * 32 + slot_);                                     \
  }
#define MF32(WA, XB, E)                                                                                             \
  { _Pragma("unroll") for (int j = 0; j < TN; ++j) _Pragma("unroll") for (int i = 0; i < TM; ++i) big[j][i] =       \
        MFMA32(el(WA[j], E), el(XB[i], E), big[j][i]); }
    FRAG32(0, 0, wa0, xb0)
    for (int it = 0; it < iters; ++it) {   // one iteration = one 32-channel chunk = four steps of 4 K = 2 MFMAs per tile
      const int st = it & 3;
      MF32(wa0, xb0, 0)
      FRAG32(st, 1, wa1, xb1)
      FENCE();
      MF32(wa0, xb0, 1) MF32(wa0, xb0, 2) MF32(wa0, xb0, 3)
      FRAG32(st, 2, wa0, xb0)
      FENCE();
      MF32(wa1, xb1, 0) MF32(wa1, xb1, 1) MF32(wa1, xb1, 2) MF32(wa1, xb1, 3)
      FRAG32(st, 3, wa1, xb1)
      FENCE();
      MF32(wa0, xb0, 0) MF32(wa0, xb0, 1) MF32(wa0, xb0, 2) MF32(wa0, xb0, 3)
      FRAG32((st + 1) & 3, 0, wa0, xb0)
      FENCE();
      MF32(wa1, xb1, 0) MF32(wa1, xb1, 1) MF32(wa1, xb1, 2) MF32(wa1, xb1, 3)
      FENCE();
      __builtin_amdgcn_s_waitcnt(0xC07F);
      FENCE();
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int j = 0; j < (BIG ? 1 : TN); ++j)
    for (int i = 0; i < (BIG ? 1 : TM); ++i) s += acc[j][i][0] + acc[j][i][1] + acc[j][i][2] + acc[j][i][3];
  for (int j = 0; j < (BIG ? TN : 1); ++j)
    for (int i = 0; i < (BIG ? TM : 1); ++i)
      for (int r = 0; r < 16; ++r) s += big[j][i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  // the two waves of a SIMD do not interleave evenly (the older one gets the issue slots first): the block's span counts
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 7) {
    atomicMin(&clk[0], c0);
    atomicMax(&clk[1], c1);
  }
}

template <int P>
void run(const char *what, double equiv_per_iter, const float *d, float *out, unsigned long long *clk, int threads = 512) {
  const int iters = 4000;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k<P>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  k<P><<<256, threads, 98304>>>(d, out, 50, clk);
  (void)hipDeviceSynchronize();
  const unsigned long long init[2] = {~0ull, 0ull};
  (void)hipMemcpy(clk, init, 16, hipMemcpyHostToDevice);
  k<P><<<256, threads, 98304>>>(d, out, iters, clk);
  hipError_t e = hipDeviceSynchronize();
  unsigned long long span[2];
  (void)hipMemcpy(span, clk, 16, hipMemcpyDeviceToHost);
  const unsigned long long hc = span[1] - span[0];
  // the waves of a SIMD share one matrix pipe: per SIMD (threads / 256) x equiv_per_iter MFMA equivalents per iteration
  printf("P%d %d waves/SIMD %-92s %6.2f cycles per 32-cycle MFMA equivalent per SIMD  (%s)\n", P, threads / 256, what,
         (double)hc / (threads / 256.0 * equiv_per_iter * iters), hipGetErrorString(e));
}

int main() {
  float *d, *out, h[4096];
  unsigned long long *clk;
  (void)hipMalloc(&d, sizeof(h));
  (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMalloc(&clk, 16);
  srand(7);
  for (int i = 0; i < 4096; ++i) {
    const float u = (rand() + 1.0f) / ((float)RAND_MAX + 2.0f), v = rand() / (float)RAND_MAX;
    h[i] = 0.125f * sqrtf(-2.f * logf(u)) * cosf(6.2831853f * v);
  }
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  run<4>("16x16x4, 10 accumulator tiles, operands in registers, no LDS reads", 80, d, out, clk);
  run<0>("16x16x4, wave tile 80x32: 7 ds_read_b128 per 40 MFMAs (shipped)", 80, d, out, clk);
  run<1>("16x16x4, wave tile 80x64: 9 ds_read_b128 per 80 MFMAs", 160, d, out, clk);
  run<2>("32x32x2, wave tile 64x32: 3 ds_read_b128 per 8 MFMAs (16 equivalents)", 64, d, out, clk);
  run<3>("32x32x2, wave tile 96x32: 4 ds_read_b128 per 12 MFMAs (24 equivalents)", 96, d, out, clk);
  run<4>("16x16x4, registers only, ONE wave per SIMD", 80, d, out, clk, 256);
  run<0>("16x16x4, wave tile 80x32, ONE wave per SIMD", 80, d, out, clk, 256);
  run<1>("16x16x4, wave tile 80x64, ONE wave per SIMD (the round-5 variant's stream)", 160, d, out, clk, 256);
  return 0;
}
