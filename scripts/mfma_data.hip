// Dev helper: does the fp32 MFMA rate on MI355X depend on the operand DATA?  Same instruction stream (8 independent
// accumulators, v_mfma_f32_16x16x4_f32 back to back, 1 / 2 waves per SIMD on all 256 CUs), operands = zeros, a
// constant, uniform [0,1), N(0,1), or unit-norm-like small values.  Reports TFLOP/s, cycles per MFMA per SIMD and the
// effective shader clock (s_memtime ticks per s_memrealtime 100 MHz tick) measured inside the kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

__global__ __launch_bounds__(512) void k(const float *__restrict__ data, float *out, int iters, unsigned long long *clk) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float va[8], vb[8];
  for (int i = 0; i < 8; ++i) {
    va[i] = data[(threadIdx.x + 64 * i) & 4095];
    vb[i] = data[(threadIdx.x + 64 * i + 2048) & 4095];
  }
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = MFMA(va[i], vb[(i + r) & 7], acc[i]);
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 7) {
    clk[0] = c1 - c0;
    clk[1] = r1 - r0;
  }
  if ((threadIdx.x & 63) == 0) {  // per-wave realtime stamps: [block][wave][start,end]; HW_ID in the top bits of start
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    clk[2 + 2 * 256 * 8 + blockIdx.x * 8 + (threadIdx.x >> 6)] = hwid;
    clk[2 + 2 * (blockIdx.x * 8 + (threadIdx.x >> 6))] = r0;
    clk[3 + 2 * (blockIdx.x * 8 + (threadIdx.x >> 6))] = r1;
  }
}

static float gauss() {
  const float u = (rand() + 1.0f) / (RAND_MAX + 2.0f), v = rand() / (float)RAND_MAX;
  return sqrtf(-2.f * logf(u)) * cosf(6.2831853f * v);
}

int main() {
  float *d, *out, h[4096];
  unsigned long long *clk, hc[2];
  hipMalloc(&d, sizeof(h));
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&clk, 16 + 256 * 8 * 16 + 256 * 8 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char *names[] = {"zeros", "constant 1.0001", "uniform [0,1)", "normal N(0,1)", "normal * 0.125 (unit-norm 64-d rows)", "relu(normal): half zeros"};
  for (int mode = 3; mode < 4; ++mode) {
    srand(7);
    for (int i = 0; i < 4096; ++i) {
      float g = gauss();
      h[i] = mode == 0 ? 0.f : mode == 1 ? 1.0001f : mode == 2 ? rand() / (float)RAND_MAX : mode == 3 ? g : mode == 4 ? 0.125f * g : (g > 0 ? g : 0.f);
    }
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int threads : {256, 512}) {
      const int iters = 20000, blocks = 256;
      k<<<blocks, threads>>>(d, out, 100, clk);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      k<<<blocks, threads>>>(d, out, iters, clk);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
      const double nm = 32.0 * iters * (threads / 256);  // MFMAs per SIMD
      const double fl = 2048.0 * 32 * (double)iters * (threads / 64) * blocks;
      if (mode == 3) {
        static unsigned long long st[2 + 256 * 16 + 256 * 8];
        hipMemcpy(st, clk, sizeof(st), hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull;
        for (int i = 0; i < 256 * (threads / 64); ++i) { int b = i / (threads / 64), w = i % (threads / 64); if (st[2 + 2 * (b * 8 + w)] < t0) t0 = st[2 + 2 * (b * 8 + w)]; }
        for (int b : {0, 1, 7, 100, 255})
          for (int w = 0; w < threads / 64; ++w)
          {
            const unsigned id = (unsigned)st[2 + 2 * 256 * 8 + b * 8 + w];
            printf("   block %3d wave %d: start %8.1f us  end %8.1f us  hw_id %08x wave_slot %u simd %u pipe %u cu %u sh %u se %u\n", b, w, (st[2 + 2 * (b * 8 + w)] - t0) / 100.0,
                   (st[3 + 2 * (b * 8 + w)] - t0) / 100.0, id, id & 15, (id >> 4) & 3, (id >> 6) & 3, (id >> 8) & 15, (id >> 12) & 1, (id >> 13) & 7);
          }
      }
      printf("%-40s %d wave(s)/SIMD: %7.3f ms %6.1f TFLOP/s  %5.1f cyc/MFMA/SIMD  clock %.0f MHz (%.0f us in-kernel)\n", names[mode], threads / 256,
             ms, fl / ms / 1e9, (double)hc[0] / nm, 100.0 * hc[0] / hc[1], hc[1] / 100.0);
    }
  }
  return 0;
}
