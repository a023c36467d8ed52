// Dev helper: issue rate of v_mfma_f32_16x16x4_f32 from ONE wave per SIMD (256 blocks x 256 threads on MI355X) as a
// function of (a) how many accumulators rotate (dependency distance 1/2/4/8 MFMAs) and (b) whether the accumulators
// live in arch VGPRs or in AGPRs.  Prints cycles per MFMA (s_memtime) — 32 = the matrix pipe's full rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

template <int NACC, bool AGPR>
__global__ __launch_bounds__(256) void k(const float *__restrict__ data, float *out, int iters, unsigned long long *clk) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float va[8], vb[8];
  for (int i = 0; i < 8; ++i) {
    va[i] = data[(threadIdx.x + 64 * i) & 4095];
    vb[i] = data[(threadIdx.x + 64 * i + 2048) & 4095];
  }
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int i = j % NACC;
      if (AGPR)
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(va[j & 7]), "v"(vb[(j + (j >> 3)) & 7]));
      else
        acc[i] = MFMA(va[j & 7], vb[(j + (j >> 3)) & 7], acc[i]);
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 7) clk[0] = c1 - c0;
}

template <int NACC, bool AGPR>
void run(const float *d, float *out, unsigned long long *clk) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k<NACC, AGPR><<<256, 256>>>(d, out, 50, clk);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<NACC, AGPR><<<256, 256>>>(d, out, iters, clk);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long hc;
  (void)hipMemcpy(&hc, clk, 8, hipMemcpyDeviceToHost);
  printf("%d rotating accumulators in %s: %6.1f cycles per MFMA, %6.1f TFLOP/s\n", NACC, AGPR ? "AGPRs" : "VGPRs",
         (double)hc / (32.0 * iters), 2048.0 * 32 * iters * 4 * 256 / ms / 1e9);
}

int main() {
  float *d, *out, h[4096];
  unsigned long long *clk;
  (void)hipMalloc(&d, sizeof(h));
  (void)hipMalloc(&out, 256 * 256 * 4);
  (void)hipMalloc(&clk, 16);
  srand(7);
  for (int i = 0; i < 4096; ++i) {
    const float u = (rand() + 1.0f) / (RAND_MAX + 2.0f), v = rand() / (float)RAND_MAX;
    h[i] = 0.125f * sqrtf(-2.f * logf(u)) * cosf(6.2831853f * v);
  }
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  run<8, false>(d, out, clk);
  run<4, false>(d, out, clk);
  run<2, false>(d, out, clk);
  run<1, false>(d, out, clk);
  run<8, true>(d, out, clk);
  run<4, true>(d, out, clk);
  run<2, true>(d, out, clk);
  run<1, true>(d, out, clk);
  return 0;
}
