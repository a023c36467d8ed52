// Dev helper: how many VALU instructions fit in the shadow of one v_mfma_f32_16x16x4_f32 (32 cycles of matrix pipe)?
// One wave per SIMD: iteration = 1 MFMA + K VALU (independent v_fma on 8 rotating registers, or v_exp), MFMAs either
// independent (8 accumulators in rotation) or one dependent chain.  Then two waves per SIMD (512-thread blocks, one per
// CU forced by a 96 KB LDS allocation): waves 0-3 issue MFMAs only, waves 4-7 VALU only — do the two pipes overlap?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int K, bool CHAIN, bool TRANS>
__global__ __launch_bounds__(256) void k1(const float *__restrict__ data, float *out, int iters, unsigned long long *clk) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float va[8], vb[8], e[8];
  for (int i = 0; i < 8; ++i) {
    va[i] = data[(threadIdx.x + 64 * i) & 4095];
    vb[i] = data[(threadIdx.x + 64 * i + 2048) & 4095];
    e[i] = va[i];
  }
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int a = CHAIN ? 0 : (j & 7);
      acc[a] = MFMA(va[j & 7], vb[(j + 3) & 7], acc[a]);
      FENCE();
#pragma unroll
      for (int v = 0; v < K; ++v) {
        const int r = (j * K + v) & 7;
        e[r] = TRANS ? __builtin_amdgcn_exp2f(e[r]) : fmaf(e[r], 1.0001f, 0.5f);
      }
      FENCE();
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + e[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 7) clk[0] = c1 - c0;
}

// two waves per SIMD: role 0 = MFMA only, role 1 = VALU only (K per "iteration"), role 2 = both in every wave
template <int K, int MODE>
__global__ __launch_bounds__(512) void k2(const float *__restrict__ data, float *out, int iters, unsigned long long *clk) {
  extern __shared__ float pad[];
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float va[8], vb[8], e[8];
  for (int i = 0; i < 8; ++i) {
    va[i] = data[(threadIdx.x + 64 * i) & 4095];
    vb[i] = data[(threadIdx.x + 64 * i + 2048) & 4095];
    e[i] = va[i];
  }
  const int wave = threadIdx.x >> 6;
  const bool do_m = MODE == 2 || wave < 4, do_v = MODE == 2 || wave >= 4;
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (do_m) acc[j & 7] = MFMA(va[j & 7], vb[(j + 3) & 7], acc[j & 7]);
      FENCE();
      if (do_v) {
#pragma unroll
        for (int v = 0; v < K; ++v) {
          const int r = (j * K + v) & 7;
          e[r] = fmaf(e[r], 1.0001f, 0.5f);
        }
      }
      FENCE();
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  float s = pad[0];
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + e[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x == 0 || threadIdx.x == 256) && blockIdx.x == 7) clk[threadIdx.x >> 8] = c1 - c0;
}

static float *d, *out;
static unsigned long long *clk;

template <int K, bool CHAIN, bool TRANS>
void run1() {
  const int iters = 3000;
  k1<K, CHAIN, TRANS><<<256, 256>>>(d, out, 50, clk);
  (void)hipDeviceSynchronize();
  k1<K, CHAIN, TRANS><<<256, 256>>>(d, out, iters, clk);
  (void)hipDeviceSynchronize();
  unsigned long long hc;
  (void)hipMemcpy(&hc, clk, 8, hipMemcpyDeviceToHost);
  printf("1 wave/SIMD  %-11s MFMA + %d %-5s : %6.1f cycles per MFMA\n", CHAIN ? "chained" : "independent", K, TRANS ? "v_exp" : "v_fma",
         (double)hc / (16.0 * iters));
}
template <int K, int MODE>
void run2() {
  const int iters = 3000;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k2<K, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  k2<K, MODE><<<256, 512, 96 * 1024>>>(d, out, 50, clk);
  (void)hipDeviceSynchronize();
  k2<K, MODE><<<256, 512, 96 * 1024>>>(d, out, iters, clk);
  (void)hipDeviceSynchronize();
  unsigned long long hc[2];
  (void)hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
  printf("2 waves/SIMD %s, K=%d: wave0 %6.1f, wave4 %6.1f cycles per iteration (1 MFMA and/or K VALU)\n",
         MODE == 2 ? "both waves MFMA + K v_fma      " : "waves 0-3 MFMA, waves 4-7 v_fma", K, (double)hc[0] / (16.0 * iters), (double)hc[1] / (16.0 * iters));
}

int main() {
  float h[4096];
  (void)hipMalloc(&d, sizeof(h));
  (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMalloc(&clk, 16);
  srand(7);
  for (int i = 0; i < 4096; ++i) {
    const float u = (rand() + 1.0f) / (RAND_MAX + 2.0f), v = rand() / (float)RAND_MAX;
    h[i] = 0.125f * sqrtf(-2.f * logf(u)) * cosf(6.2831853f * v);
  }
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  run1<0, false, false>(); run1<1, false, false>(); run1<2, false, false>(); run1<3, false, false>(); run1<4, false, false>();
  run1<6, false, false>(); run1<8, false, false>(); run1<12, false, false>();
  run1<0, true, false>(); run1<1, true, false>(); run1<2, true, false>(); run1<4, true, false>();
  run1<1, false, true>(); run1<2, false, true>(); run1<4, false, true>();
  run2<4, 0>(); run2<8, 0>(); run2<14, 0>();
  run2<0, 2>(); run2<2, 2>(); run2<4, 2>(); run2<8, 2>();
  return 0;
}
