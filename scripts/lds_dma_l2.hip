// Round 6: how fast can the LDS-DMA path (global_load_lds_dwordx4: HBM / L2 -> LDS without VGPRs) fill LDS
// chip-wide when the SOURCE IS L2-RESIDENT, with 8 waves per CU issuing — the plane GEMM's situation (its operands are re-read
// from L2: HBM-side traffic is 1.10x algorithmic while the LDS fill is many times that)?  DESIGN.md (round 3) rejected a
// split-precision bf16 plane GEMM with "the LDS-DMA ceiling is 6.4 TB/s", a figure measured for HBM streams.  Gate: >= 12 TB/s
// chip-wide from L2 would make the six-product bf16 kernel worth building.
//
// 256 blocks (one per CU, forced by a 96 KB LDS allocation) x 512 threads; every wave keeps DEPTH 1-KB DMAs in flight (counted
// vmcnt) into its own LDS ring; the source footprint decides where the data comes from: 1 MB (one L2 slice set, shared by all
// blocks of an XCD), 64 MB (Infinity Cache), 2 GB (HBM).  `--shared 0` gives every block its own slice of the footprint.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/lds_dma_l2 scripts/lds_dma_l2.hip && scripts/lds_dma_l2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__device__ __forceinline__ void glds16(const float *g, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int DEPTH>
__global__ __launch_bounds__(512) void k(const float *__restrict__ src, size_t footprint_floats, int iters, int shared, float *out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // 96 KB: [8 waves][DEPTH <= 12][256 floats]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = (unsigned)(size_t)(smem) + (unsigned)(wave * 12 * 1024);
  // a block walks its footprint in 8 KB steps (8 waves x 1 KB), wrapping around; all blocks the same addresses when shared
  const size_t block_off = shared ? 0 : ((size_t)blockIdx.x * (footprint_floats / gridDim.x)) & ~(size_t)2047;
  const size_t span = shared ? footprint_floats : footprint_floats / gridDim.x;
  size_t pos = (size_t)wave * 256 + lane * 4;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      glds16(src + block_off + pos, lds0 + d * 1024);
      pos += 2048;
      if (pos >= span) pos -= span;
    }
    wait_vmcnt<DEPTH / 2>();     // keep at least half the ring in flight across iterations
  }
  wait_vmcnt<0>();
  __syncthreads();
  if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = smem[0];
}

template <int DEPTH>
static double run(const float *src, size_t fp_bytes, int shared, int iters) {
  const size_t lds = 96 * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(k<DEPTH>, dim3(256), dim3(512), lds, 0, src, fp_bytes / 4, iters / 4, shared, (float *)nullptr);
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  hipLaunchKernelGGL(k<DEPTH>, dim3(256), dim3(512), lds, 0, src, fp_bytes / 4, iters, shared, (float *)nullptr);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = 256.0 * 8 * (double)iters * DEPTH * 1024.0;
  return bytes / (ms * 1e-3) / 1e12;
}

int main(int argc, char **argv) {
  const size_t cap = (size_t)2 << 30;
  float *src = nullptr;
  if (hipMalloc(&src, cap) != hipSuccess) return 1;
  hipMemset(src, 0, cap);
  printf("| source footprint | shared by all blocks | DMAs in flight per wave | chip-wide LDS fill TB/s |\n|---|---|---:|---:|\n");
  const size_t fps[] = {(size_t)256 << 10, (size_t)1 << 20, (size_t)2 << 20, (size_t)16 << 20, (size_t)64 << 20, (size_t)2 << 30};
  const char *names[] = {"256 KB (L2)", "1 MB (L2)", "2 MB (L2)", "16 MB (L2 x 8 / MALL)", "64 MB (MALL)", "2 GB (HBM)"};
  for (int f = 0; f < 6; ++f)
    for (int shared = 1; shared >= 0; --shared) {
      if (!shared && fps[f] < ((size_t)16 << 20)) continue;   // private slices need a footprint that splits 256 ways
      const int iters = 4000;
      printf("| %s | %s | 4 | %.2f |\n", names[f], shared ? "yes" : "no", run<4>(src, fps[f], shared, iters));
      printf("| %s | %s | 8 | %.2f |\n", names[f], shared ? "yes" : "no", run<8>(src, fps[f], shared, iters / 2));
      printf("| %s | %s | 12 | %.2f |\n", names[f], shared ? "yes" : "no", run<12>(src, fps[f], shared, iters / 3));
      fflush(stdout);
    }
  hipFree(src);
  return 0;
}
