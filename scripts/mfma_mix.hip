// Dev helper: what does the ORDER of dependent-chain MFMAs, independent MFMAs and VALU cost on one wave per SIMD?
// All patterns: 32 v_mfma_f32_16x16x4_f32 per iteration (16 on ONE accumulator = the hill climb's S chain, 16 rotating
// over four accumulators = its accumulate MFMAs), order pinned with sched_barrier(0).  cycles per MFMA, 32 = full rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int P>
__global__ __launch_bounds__(256) void k(const float *__restrict__ data, float *out, int iters, unsigned long long *clk) {
  f32x4 S = {0, 0, 0, 0}, S2 = {0, 0, 0, 0}, A[4];
  for (int i = 0; i < 4; ++i) A[i] = f32x4{0, 0, 0, 0};
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
      if (P == 0) {  // S A S A ... no VALU
        S = MFMA(va[j & 7], vb[j & 7], S); FENCE();
        A[j & 3] = MFMA(vb[j & 7], va[j & 7], A[j & 3]); FENCE();
      } else if (P == 1) {  // S A v S A v : one independent VALU behind every MFMA
        S = MFMA(va[j & 7], vb[j & 7], S); e[j & 7] = fmaf(e[j & 7], 1.0001f, 0.5f); FENCE();
        A[j & 3] = MFMA(vb[j & 7], va[j & 7], A[j & 3]); e[(j + 4) & 7] = fmaf(e[(j + 4) & 7], 0.9999f, 0.25f); FENCE();
      } else if (P == 2) {  // two VALU behind the A MFMA only, S -> A back to back
        S = MFMA(va[j & 7], vb[j & 7], S); FENCE();
        A[j & 3] = MFMA(vb[j & 7], va[j & 7], A[j & 3]); e[j & 7] = fmaf(e[j & 7], 1.0001f, 0.5f); e[(j + 4) & 7] = fmaf(e[(j + 4) & 7], 0.9999f, 0.25f); FENCE();
      } else if (P == 5) {  // two S chains: S S2 A A, one VALU behind each
        if (j < 8) {
          S = MFMA(va[j & 7], vb[j & 7], S); e[j & 7] = fmaf(e[j & 7], 1.0001f, 0.5f); FENCE();
          S2 = MFMA(va[(j + 1) & 7], vb[j & 7], S2); e[(j + 2) & 7] = fmaf(e[(j + 2) & 7], 1.0001f, 0.5f); FENCE();
          A[(2 * j) & 3] = MFMA(vb[j & 7], va[j & 7], A[(2 * j) & 3]); e[(j + 4) & 7] = fmaf(e[(j + 4) & 7], 0.9999f, 0.25f); FENCE();
          A[(2 * j + 1) & 3] = MFMA(vb[j & 7], va[(j + 3) & 7], A[(2 * j + 1) & 3]); e[(j + 6) & 7] = fmaf(e[(j + 6) & 7], 0.9999f, 0.25f); FENCE();
        }
      }
    }
    if (P == 3 || P == 4) {  // 16 S back to back (nothing between), then 16 A with 0 (P3) / 2 (P4) VALU behind each
#pragma unroll
      for (int j = 0; j < 16; ++j) { S = MFMA(va[j & 7], vb[j & 7], S); FENCE(); }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        A[j & 3] = MFMA(vb[j & 7], va[j & 7], A[j & 3]);
        if (P == 4) { e[j & 7] = fmaf(e[j & 7], 1.0001f, 0.5f); e[(j + 4) & 7] = fmaf(e[(j + 4) & 7], 0.9999f, 0.25f); }
        FENCE();
      }
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += A[i][0] + A[i][1] + A[i][2] + A[i][3];
  for (int i = 0; i < 8; ++i) s += e[i];
  s += S[0] + S[1] + S[2] + S[3] + S2[0] + S2[1] + S2[2] + S2[3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 7) clk[0] = c1 - c0;
}

template <int P>
void run(const char *what, const float *d, float *out, unsigned long long *clk) {
  const int iters = 4000;
  k<P><<<256, 256>>>(d, out, 50, clk);
  (void)hipDeviceSynchronize();
  k<P><<<256, 256>>>(d, out, iters, clk);
  (void)hipDeviceSynchronize();
  unsigned long long hc;
  (void)hipMemcpy(&hc, clk, 8, hipMemcpyDeviceToHost);
  printf("P%d %-86s %6.1f cycles per MFMA\n", P, what, (double)hc / (32.0 * iters));
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
  run<0>("chain / independent alternating (S A S A), no VALU", d, out, clk);
  run<1>("S v A v: one independent VALU behind every MFMA", d, out, clk);
  run<2>("S A v v: the chain MFMA directly followed by the independent one, two VALU behind that", d, out, clk);
  run<3>("16 chain MFMAs back to back, then the 16 independent ones, no VALU", d, out, clk);
  run<4>("16 chain MFMAs back to back, then 16 x (independent MFMA, v, v)", d, out, clk);
  run<5>("two chains: S S2 A A with one VALU behind each", d, out, clk);
  return 0;
}
