// Dev helper: practical fp32 MFMA ceiling on this box (pure v_mfma_f32_16x16x4_f32 issue).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(512) void k(float *out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  a += threadIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float *d;
  hipMalloc(&d, 256 * 512 * 4 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int blocks : {256, 240}) {
    for (int threads : {256, 512}) {
      const int iters = 2000;
      k<10><<<blocks, threads>>>(d, 10, 1.f, 1.f);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      k<10><<<blocks, threads>>>(d, iters, 1.0001f, 0.9999f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      double fl = 2.0 * 16 * 16 * 4 * 10 * 8 * (double)iters * (threads / 64) * blocks;
      printf("blocks %d threads %d: %.3f ms  %.1f TFLOP/s\n", blocks, threads, ms, fl / ms / 1e9);
    }
  }
  return 0;
}
