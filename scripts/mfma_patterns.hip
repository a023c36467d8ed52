// Dev helper: what limits fp32 MFMA issue in patterns the kernels use (MI355X).  Each variant does the same
// number of v_mfma_f32_16x16x4_f32 per wave; reports TFLOP/s at 1, 2 and 4 waves per SIMD.
//   P0 independent accumulators, constant operands (the ceiling)
//   P1 the hill-climbing pattern: ONE dependent chain interleaved 1:1 with four chains of four
//   P2 like P0 but every MFMA takes its A/B from a different register (16 live operand registers)
//   P3 P0 plus an independent v_exp/v_mul pair per 4 MFMAs
//   P4 P0 plus a ds_read_b128 per 4 MFMAs feeding the operands (LDS-resident fragments)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

template <int P>
__global__ __launch_bounds__(1024) void k(float *out, int iters, float a, float b) {
  __shared__ float4 lds[1024];
  lds[threadIdx.x] = make_float4(a, b, a, b);
  __syncthreads();
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float va[8], vb[8];
  for (int i = 0; i < 8; ++i) {
    va[i] = a + i * 1e-6f + threadIdx.x * 1e-7f;
    vb[i] = b - i * 1e-6f;
  }
  float e = a;
  for (int it = 0; it < iters; ++it) {
    if (P == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = MFMA(a, b, acc[i]);
    } else if (P == 1) {
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        acc[4] = MFMA(va[kk & 7], vb[kk & 7], acc[4]);            // dependent chain
        acc[kk & 3] = MFMA(vb[kk & 7], va[kk & 7], acc[kk & 3]);  // 4 chains of 4
      }
    } else if (P == 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = MFMA(va[i], vb[(i + r) & 7], acc[i]);
    } else if (P == 3) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = MFMA(a, b, acc[i]);
        e = __expf(e * 0.999f);
        e = __expf(e * 0.998f);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float4 f = lds[(threadIdx.x + 17 * (r + it)) & 1023];
        acc[(4 * r) & 7] = MFMA(f.x, b, acc[(4 * r) & 7]);
        acc[(4 * r + 1) & 7] = MFMA(f.y, b, acc[(4 * r + 1) & 7]);
        acc[(4 * r + 2) & 7] = MFMA(f.z, b, acc[(4 * r + 2) & 7]);
        acc[(4 * r + 3) & 7] = MFMA(f.w, b, acc[(4 * r + 3) & 7]);
      }
    }
  }
  float s = e;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int P>
void run(float *d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int threads : {256, 512, 1024}) {
    const int iters = 3000, blocks = 256;
    k<P><<<blocks, threads>>>(d, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<P><<<blocks, threads>>>(d, iters, 1.0001f, 0.9999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * 16 * 16 * 4 * 32 * (double)iters * (threads / 64) * blocks;
    printf("P%d  %d waves/SIMD: %7.3f ms  %6.1f TFLOP/s\n", P, threads / 256, ms, fl / ms / 1e9);
  }
}
int main() {
  float *d;
  hipMalloc(&d, 256 * 1024 * 4);
  run<0>(d);
  run<1>(d);
  run<2>(d);
  run<3>(d);
  run<4>(d);
  return 0;
}
