// Seeded von-Mises-Fisher mean-shift clustering on gfx950 — hand-written HIP.
//
// Replaces /root/reference/lib/utils/mean_shift.py (cosine metric):
//   select_smart_seeds       :128-189  -> fps_step_kernel        (HBM/L2-bound streaming + grid argmax)
//   seed_hill_climbing_ball  :79-109   -> hc_iter_kernel         (fp32 MFMA, W never materialised)
//                                         + hc_finalize_kernel   (partial reduce + L2 normalise)
//   connected_components     :41-76    -> seed_cc_kernel         (one wavefront, ballots)
//   mean_shift_smart_init    :211-227  -> assign_kernel          (fp32 MFMA + row argmin + histogram)
//                                         + relabel_swap_kernel  (largest cluster <-> label 0)
//
// Data layout: X is pixel-major [batch][n][64] fp32 (256 B per pixel), seeds Z [batch][m][64].
// All reductions have a fixed order, so results are run-to-run deterministic.
#include "common.h"
#include "prof.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>

#include <mutex>
#include <utility>

namespace uoc {

constexpr int C = UOC_EMBED_DIM;  // 64 channels
constexpr int FPS_THREADS = 256;
constexpr int FPS_MAX_BLOCKS = 1024;
constexpr int HC_THREADS = 256;
constexpr int HC_MAX_BLOCKS = 1024;
constexpr int ZP = 72;  // LDS row pitch (floats) of the seed tile: 18 x 16 B => conflict-free b128 fragment reads
constexpr int NLAB = UOC_MAX_SEEDS;

struct ArgMax {
  float val;
  int idx;
};

__device__ __forceinline__ bool better(const ArgMax &a, const ArgMax &b) {
  // torch.argmax semantics: larger value wins, ties -> lower index.
  return a.val > b.val || (a.val == b.val && a.idx < b.idx);
}

__device__ __forceinline__ ArgMax wave_argmax(ArgMax v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    ArgMax o;
    o.val = __shfl_xor(v.val, off);
    o.idx = __shfl_xor(v.idx, off);
    if (better(o, v)) v = o;
  }
  return v;
}

// -------------------------------------------------------------------------------------------
// Farthest-point seed selection, one launch per step.
//   step s: (a) every block reduces the previous step's per-block argmax partials -> index of
//   seed s; (b) block 0 records seed s; (c) all blocks stream X once: d = 0.5(1 - x.seed),
//   dmin = min(dmin, d), per-block argmax(dmin) -> partials for step s+1.
// 16 lanes share one pixel row (float4 each => a wave reads 1 KiB contiguous per load).
// -------------------------------------------------------------------------------------------
// NH = number of 64-channel halves of an embedding (1: the 64-d fields of every shipped mode but 'cat';
// 2: 128-d fields stored as two planes X[b][h][n][64], seeds / Z likewise [b][h][m][64]); a dot product
// is the sum over the halves.
template <int NH>
__global__ __launch_bounds__(FPS_THREADS) void fps_step_kernel(
    const float *__restrict__ X, int n, int m, int step, int num_init, const int *__restrict__ first_index,
    float *__restrict__ dmin, float *__restrict__ seeds, int *__restrict__ indices,
    const ArgMax *__restrict__ part_in, ArgMax *__restrict__ part_out) {
  const int b = blockIdx.y;
  const int nblk = gridDim.x;
  X += (size_t)b * NH * n * C;
  dmin += (size_t)b * n;
  seeds += (size_t)b * NH * m * C;
  indices += (size_t)b * m;
  part_in += (size_t)b * FPS_MAX_BLOCKS;
  part_out += (size_t)b * FPS_MAX_BLOCKS;

  __shared__ ArgMax red[FPS_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // continuation (select_smart_seeds(init_seeds=..., num_init_seeds=k), :142-170): rows 0..k-1 of `seeds` are the
  // caller's; their steps only fold their distances into dmin and record index -1
  const bool given = step < num_init;
  int cur;
  if (given) {
    cur = -1;
  } else if (step == 0) {
    cur = first_index[b];
    if (cur < 0 || cur >= n) cur = 0;  // the host mirrors validate; never read outside X
  } else {
    ArgMax best = {-INFINITY, INT_MAX};
    for (int i = tid; i < nblk; i += FPS_THREADS) {
      ArgMax a = part_in[i];
      if (better(a, best)) best = a;
    }
    best = wave_argmax(best);
    if (lane == 0) red[wave] = best;
    __syncthreads();
    best = red[0];
#pragma unroll
    for (int w = 1; w < FPS_THREADS / 64; ++w)
      if (better(red[w], best)) best = red[w];
    cur = best.idx;
    if (cur < 0 || cur >= n) cur = 0;  // all-NaN distances leave idx = INT_MAX (the on-chip kernel clamps too)
    __syncthreads();
  }
  if (blockIdx.x == 0) {
    if (tid == 0) indices[step] = cur;
    if (!given && tid < NH * C)
      seeds[((size_t)(tid / C) * m + step) * C + tid % C] = X[((size_t)(tid / C) * n + cur) * C + tid % C];
  }
  if (step == m - 1) return;  // the distances to the last seed are never consumed (:174 uses [:, :i])

  const int t = lane & 15, g = lane >> 4;
  float4 sv[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h)
    sv[h] = given ? *reinterpret_cast<const float4 *>(seeds + ((size_t)h * m + step) * C + 4 * t)
                  : *reinterpret_cast<const float4 *>(X + ((size_t)h * n + cur) * C + 4 * t);
  ArgMax best = {-INFINITY, INT_MAX};
  const int nchunk = (n + 63) >> 6;
  for (int chunk = blockIdx.x * (FPS_THREADS / 64) + wave; chunk < nchunk; chunk += nblk * (FPS_THREADS / 64)) {
    const int base = chunk << 6;
    float4 x[NH][16];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int p = base + 4 * i + g;
        x[h][i] = (p < n) ? *reinterpret_cast<const float4 *>(X + ((size_t)h * n + p) * C + 4 * t)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    float mine = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float s = x[0][i].x * sv[0].x;
      s = fmaf(x[0][i].y, sv[0].y, s);
      s = fmaf(x[0][i].z, sv[0].z, s);
      s = fmaf(x[0][i].w, sv[0].w, s);
#pragma unroll
      for (int h = 1; h < NH; ++h) {
        s = fmaf(x[h][i].x, sv[h].x, s);
        s = fmaf(x[h][i].y, sv[h].y, s);
        s = fmaf(x[h][i].z, sv[h].z, s);
        s = fmaf(x[h][i].w, sv[h].w, s);
      }
      s = row16_sum(s);
      if (t == i) mine = s;
    }
    const int p = base + 4 * t + g;
    if (p < n) {
      float d = 0.5f * (1.0f - mine);
      if (step > 0) d = fminf(d, dmin[p]);
      dmin[p] = d;
      if (d > best.val) {  // p ascends per lane: strict '>' keeps the lowest index
        best.val = d;
        best.idx = p;
      }
    }
  }
  best = wave_argmax(best);
  if (lane == 0) red[wave] = best;
  __syncthreads();
  if (tid == 0) {
    best = red[0];
#pragma unroll
    for (int w = 1; w < FPS_THREADS / 64; ++w)
      if (better(red[w], best)) best = red[w];
    part_out[blockIdx.x] = best;
  }
}

// -------------------------------------------------------------------------------------------
// Persistent farthest-point sampling: ONE launch for all m steps.  Every block keeps its share of
// X resident on chip for the whole call: each LANE owns up to 3 whole pixels in VGPRs (3 x 64 floats)
// plus one in LDS, i.e. up to 2048 pixels per CU, so a step costs no HBM/L2 traffic — the new seed
// row is 64 wave-uniform scalars and a pixel's distance is a private 64-FMA chain (no cross-lane
// reduction at all; ~1 VALU instruction per pixel-channel).  The grid-wide argmax is a tagged-granule
// all-gather (MI355X_MICROARCH "R2": the 8-byte {value, index|tag} word written with ONE agent-scope
// atomic store is its own flag; no fences, no counters): each block publishes its (max, lowest
// index) for step s in slot [s&1][block], one wave per block sweeps the row until every tag equals
// s+1.  Two slots suffice: a block can only be one step ahead of the slowest reader.
// Placement independent; needs all blocks co-resident => launched cooperatively with grid <= #CUs
// (512-thread blocks, one per CU); every spin is bounded and sets *status on expiry.
// Measured per step (480x640, 256 blocks): exchange ~3.2 us, compute + block reduce ~1.3 us.
// -------------------------------------------------------------------------------------------
__device__ int g_fps_timeout = 0;  // sticky: some block's bounded spin expired (read + cleared by uoc_ms_check)

constexpr int FPP_THREADS = 512;
constexpr int FPP_WAVES = FPP_THREADS / 64;
constexpr int FPP_RS = 3;  // pixels per lane held in registers
constexpr int FPP_LS = 1;  // pixels per lane held in LDS
constexpr int FPP_SLOTS = FPP_RS + FPP_LS;
constexpr int FPP_PIX_PER_BLOCK = FPP_THREADS * FPP_SLOTS;  // 2048

// Wave-wide argmax with torch semantics (max value, ties -> lowest index), result wave-uniform.
// DPP row reductions (every lane of a 16-lane row ends with the row result) + 4 v_readlane.
__device__ __forceinline__ ArgMax wave_argmax_fast(ArgMax a) {
  float v = a.val;
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  const int vi = __float_as_int(v);
  const float m = fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(vi, 0)), __int_as_float(__builtin_amdgcn_readlane(vi, 16))),
                        fmaxf(__int_as_float(__builtin_amdgcn_readlane(vi, 32)), __int_as_float(__builtin_amdgcn_readlane(vi, 48))));
  int i = (a.val == m) ? a.idx : INT_MAX;
  i = min(i, dpp_i<0xB1>(i));
  i = min(i, dpp_i<0x4E>(i));
  i = min(i, dpp_i<0x141>(i));
  i = min(i, dpp_i<0x140>(i));
  const int mi = min(min(__builtin_amdgcn_readlane(i, 0), __builtin_amdgcn_readlane(i, 16)),
                     min(__builtin_amdgcn_readlane(i, 32), __builtin_amdgcn_readlane(i, 48)));
  return ArgMax{m, mi};
}

__device__ __forceinline__ unsigned long long fpp_pack(float val, int idx, int tag) {
  return ((unsigned long long)__float_as_uint(val) << 32) | (unsigned)(idx & 0xFFFFFF) | ((unsigned)(tag & 0xFF) << 24);
}

__global__ __launch_bounds__(FPP_THREADS) void fps_persistent_kernel(
    const float *__restrict__ X, int n, int m, int bpi, int nslots, const int *__restrict__ first_index,
    float *__restrict__ seeds, int *__restrict__ indices, unsigned long long *gran, int *status) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [FPP_LS][16 float4-chunks][FPP_THREADS] float4
  __shared__ ArgMax red[FPP_WAVES];
  __shared__ int s_idx;
  const int item = blockIdx.x / bpi, blk = blockIdx.x - item * bpi;
  X += (size_t)item * n * C;
  seeds += (size_t)item * m * C;
  indices += (size_t)item * m;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // pixel of (slot j, thread tid): consecutive threads own consecutive pixels; slots ascend per lane
  const int pbase = blk * (FPP_THREADS * nslots) + tid;  // + j * FPP_THREADS
  float4 *lds = reinterpret_cast<float4 *>(smem) + tid;  // chunk c4 at lds[c4 * FPP_THREADS]

  // ---- load this thread's pixels once -------------------------------------------------------
  // Register pixels 0 and 1 are kept as PAIRS (round 5): their two dot products run as v_pk_fma_f32 — two fused multiply-adds per
  // lane and instruction, each pixel's sum still one sequential fma chain over the channels (bit-identical to the scalar form),
  // the seed row's scalars broadcast through op_sel.  The step's 256 dependent-chain FMAs per lane were 0.9 of its ~5.5 us.
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  static_assert(FPP_RS == 3, "pixels 0/1 packed, pixel 2 scalar");
  f32x2 x01[C];
  float x2[C];
#pragma unroll
  for (int j = 0; j < FPP_RS; ++j) {
    const int p = pbase + j * FPP_THREADS;
    const bool ok = j < nslots && p < n;
#pragma unroll
    for (int c4 = 0; c4 < C / 4; ++c4) {
      const float4 v = ok ? *reinterpret_cast<const float4 *>(X + (size_t)p * C + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (j == 0) x01[4 * c4 + k].x = e[k];
        if (j == 1) x01[4 * c4 + k].y = e[k];
        if (j == 2) x2[4 * c4 + k] = e[k];
      }
    }
  }
  if (nslots > FPP_RS) {
    const int p = pbase + FPP_RS * FPP_THREADS;
#pragma unroll
    for (int c4 = 0; c4 < C / 4; ++c4)
      lds[c4 * FPP_THREADS] = (p < n) ? *reinterpret_cast<const float4 *>(X + (size_t)p * C + 4 * c4)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float dm[FPP_SLOTS];
#pragma unroll
  for (int j = 0; j < FPP_SLOTS; ++j) dm[j] = 0.f;

  int cur = __builtin_amdgcn_readfirstlane(first_index[item]);
  if (cur < 0 || cur >= n) cur = 0;  // the host mirrors validate; never read outside X
  if (blk == 0) {
    if (tid == 0) indices[0] = cur;
    if (tid < 16) *reinterpret_cast<float4 *>(seeds + 4 * tid) = *reinterpret_cast<const float4 *>(X + (size_t)cur * C + 4 * tid);
  }
#ifdef UOC_FPS_TIMING
  unsigned long long tc_comp = 0, tc_red = 0, tc_sweep = 0, tc_tail = 0;
#define UOC_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define UOC_T(v)
#endif
  for (int step = 0; step + 1 < m; ++step) {
    UOC_T(t1_);
    // the seed row: `cur` is wave-uniform, so these are scalar loads and the FMAs take SGPR operands
    const float *__restrict__ srow = X + (size_t)cur * C;
    ArgMax best = {-INFINITY, INT_MAX};
    f32x2 s01 = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < C; ++c) s01 = __builtin_elementwise_fma(x01[c], f32x2{srow[c], srow[c]}, s01);
    float s2 = 0.f;
    if (nslots > 2) {
#pragma unroll
      for (int c = 0; c < C; ++c) s2 = fmaf(x2[c], srow[c], s2);
    }
#pragma unroll
    for (int j = 0; j < FPP_RS; ++j) {
      if (j < nslots) {
        const float s_ = j == 0 ? s01.x : j == 1 ? s01.y : s2;
        float d_ = 0.5f * (1.0f - s_);
        if (step > 0) d_ = fminf(d_, dm[j]);
        dm[j] = d_;
        const int p_ = pbase + j * FPP_THREADS;
        if (p_ < n && d_ > best.val) {  // slots ascend per lane: '>' keeps the lowest index
          best.val = d_;
          best.idx = p_;
        }
      }
    }
    if (nslots > FPP_RS) {
      float s_ = 0.f;
#pragma unroll
      for (int c4 = 0; c4 < C / 4; ++c4) {
        const float4 v = lds[c4 * FPP_THREADS];
        s_ = fmaf(v.x, srow[4 * c4 + 0], s_);
        s_ = fmaf(v.y, srow[4 * c4 + 1], s_);
        s_ = fmaf(v.z, srow[4 * c4 + 2], s_);
        s_ = fmaf(v.w, srow[4 * c4 + 3], s_);
      }
      float d_ = 0.5f * (1.0f - s_);
      if (step > 0) d_ = fminf(d_, dm[FPP_RS]);
      dm[FPP_RS] = d_;
      const int p_ = pbase + FPP_RS * FPP_THREADS;
      if (p_ < n && d_ > best.val) {
        best.val = d_;
        best.idx = p_;
      }
    }
    UOC_T(t2_);
    best = wave_argmax_fast(best);
    if (lane == 0) red[wave] = best;
    __syncthreads();
    UOC_T(t3_);
    if (wave == 0) {
      ArgMax b2 = lane < FPP_WAVES ? red[lane] : ArgMax{-INFINITY, INT_MAX};
      b2 = wave_argmax_fast(b2);
      unsigned long long *row = gran + ((size_t)(step & 1) * gridDim.x + (size_t)item * bpi);
      const int tag = (step + 1) & 0xFF;
      if (lane == 0) __hip_atomic_store(row + blk, fpp_pack(b2.val, b2.idx, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // sweep the item's row until every granule carries this step's tag
      ArgMax acc = {-INFINITY, INT_MAX};
      unsigned spins = 0;
      bool done = false;
      while (!done) {
        bool ok = true;
        ArgMax a = {-INFINITY, INT_MAX};
        unsigned long long gv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)  // bpi <= 256: all of this lane's polls are in flight together
          gv[k] = (lane + 64 * k < bpi) ? __hip_atomic_load(row + lane + 64 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                        : ((unsigned long long)tag << 24);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned long long v = gv[k];
          ok &= (int)((v >> 24) & 0xFF) == tag;
          if (lane + 64 * k < bpi) {
            ArgMax c = {__uint_as_float((unsigned)(v >> 32)), (int)(v & 0xFFFFFF)};
            if (better(c, a)) a = c;
          }
        }
        if (__all(ok)) {
          acc = a;
          done = true;
        } else {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1u << 22) || (((spins & 1023) == 0) && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            if (lane == 0) {
              __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(&g_fps_timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            acc = ArgMax{0.f, 0};
            done = true;
          }
        }
      }
      acc = wave_argmax_fast(acc);
      if (lane == 0) s_idx = acc.idx;
    }
    __syncthreads();
    UOC_T(t4_);
    cur = __builtin_amdgcn_readfirstlane(s_idx);
    if (cur < 0 || cur >= n) cur = 0;  // only reachable after a timeout
    if (blk == 0) {
      if (tid == 0) indices[step + 1] = cur;
      if (tid < 16)
        *reinterpret_cast<float4 *>(seeds + (size_t)(step + 1) * C + 4 * tid) =
            *reinterpret_cast<const float4 *>(X + (size_t)cur * C + 4 * tid);
    }
#ifdef UOC_FPS_TIMING
    const unsigned long long t5_ = __builtin_amdgcn_s_memtime();
    tc_comp += t2_ - t1_;
    tc_red += t3_ - t2_;
    tc_sweep += t4_ - t3_;
    tc_tail += t5_ - t4_;
#endif
  }
#ifdef UOC_FPS_TIMING
  if ((blockIdx.x == 0 || blockIdx.x == gridDim.x / 2) && tid == 0)
    printf("[fps timing blk %d] per step (cycles): compute %.1f reduce %.1f sweep %.1f tail %.1f\n", (int)blockIdx.x,
           tc_comp / (double)(m - 1), tc_red / (double)(m - 1), tc_sweep / (double)(m - 1), tc_tail / (double)(m - 1));
#endif
#undef UOC_T
}

// -------------------------------------------------------------------------------------------
// One hill-climbing iteration.  Each wave owns 16-pixel tiles and all ST seed tiles:
//   S^T[pixel][seed] = X Z^T       16 x v_mfma_f32_16x16x4_f32 per (16 px x 16 seeds)
//   W = exp(kappa S)               in registers: the D fragment of step 1 IS the A fragment of step 3
//   acc[seed][chan] += W^T X       16 x v_mfma_f32_16x16x4_f32
// K-index (channel) and N-index (channel) permutations are chosen so every global load is a
// float4: xa[v] = X[p=t][16v+4q..], xb[r] = X[p=4q+r][4t..].  W (122.9 MB in the reference) never
// exists in memory.  Wave partials are reduced through LDS; block partials go to HBM and are
// reduced (fixed order) + L2-normalised by hc_finalize_kernel.
// Two other formulations were built, verified and measured on MI355X against this one (113 us per iteration
// incl. finalize at 480x640): "seed tile per wave", Z fragments and a 16x64 accumulator in registers, no LDS, 4
// waves/SIMD, every wave loading the pixels itself: 137 us (7x the L1/L2 traffic); the same with the pixels
// streamed once per block through an LDS-DMA ring (94 VGPRs, 4 waves/SIMD): 113-117 us.  Without any exp()
// that kernel takes 100 us, with __expf 104 us.  A fourth, register-blocked one (two seed tiles per wave resident
// in registers, two pixel tiles per 16 LDS reads = 0.125 reads per MFMA, 3 waves/SIMD, two-phase software
// pipeline): 109 us.  All four sit at ~56 % of the fp32 MFMA peak (12 % of the MFMAs are the padding of 100 seeds
// to 7 tiles), so the simplest one stays; no single bound was found (clock 2.4 GHz, 850 W under this kernel).
// -------------------------------------------------------------------------------------------
#ifndef UOC_EXP
#define UOC_EXP expf
#endif
__device__ __forceinline__ float f4c(const float4 &v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// NH = 2 (128-d fields as two 64-channel planes): S sums over both halves; the accumulators of one block
// cover ONE half (blockIdx.z), so S is computed twice — the price of keeping the 64-d register tiling.
template <int ST, int NH>
__global__ __launch_bounds__(HC_THREADS) __attribute__((amdgpu_waves_per_eu(2))) void hc_iter_kernel(const float *__restrict__ X, int n,
                                                             const float *__restrict__ Z, int m, float kappa,
                                                             float *__restrict__ partial_, int nvb) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Zs = smem;  // [NH][ST*16][ZP]; later reused as the cross-wave reduction buffer
  const int b = blockIdx.y;
  const int nblk = nvb;                    // VIRTUAL blocks (hc_virtual_blocks: n only), walked by the physical ones
  const int hz = NH > 1 ? blockIdx.z : 0;  // the half this block accumulates
  X += (size_t)b * NH * n * C;
  Z += (size_t)b * NH * m * C;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = lane & 15, q = lane >> 4;
  for (int vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
  float *partial = partial_ + (((size_t)b * nblk + vb) * NH + hz) * (ST * 16) * C;

  for (int i = tid; i < NH * ST * 16 * (C / 4); i += HC_THREADS) {
    const int h = i / (ST * 16 * (C / 4)), j = i % (ST * 16 * (C / 4));
    const int row = j / (C / 4), c4 = j % (C / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < m) v = *reinterpret_cast<const float4 *>(Z + ((size_t)h * m + row) * C + 4 * c4);
    *reinterpret_cast<float4 *>(Zs + (h * ST * 16 + row) * ZP + 4 * c4) = v;
  }
  __syncthreads();

  f32x4 acc[ST][4];
#pragma unroll
  for (int s = 0; s < ST; ++s)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[s][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ntile = (n + 15) >> 4;
  const int stride = nblk * (HC_THREADS / 64);
  int tile = vb * (HC_THREADS / 64) + wave;

  float4 xa[NH * 4], xb[4];
  auto load_tile = [&](int tl, float4(&a)[NH * 4], float4(&bb)[4]) {
    const int pa = tl * 16 + t;
#pragma unroll
    for (int v = 0; v < NH * 4; ++v)
      a[v] = (tl < ntile && pa < n)
                 ? *reinterpret_cast<const float4 *>(X + ((size_t)(v >> 2) * n + pa) * C + 16 * (v & 3) + 4 * q)
                 : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int pb = tl * 16 + 4 * q + r;
      bb[r] = (tl < ntile && pb < n) ? *reinterpret_cast<const float4 *>(X + ((size_t)hz * n + pb) * C + 4 * t)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  load_tile(tile, xa, xb);

  for (; tile < ntile; tile += stride) {
    float4 na[NH * 4], nb[4];
    if (NH == 1) load_tile(tile + stride, na, nb);  // software prefetch of the wave's next tile
    // Software pipeline over the seed tiles (3 stages, fully unrolled): in step i the 16-deep
    // DEPENDENT MFMA chain S_i = X Z_i^T is interleaved 1:1 with the 16 INDEPENDENT accumulate MFMAs of
    // tile i-2 (hides the 40-cycle dependent-accumulator latency), while exp() of tile i-1 runs on the
    // VALU underneath.
    f32x4 Sv[ST];
    float wv[ST][4];
    // The seed fragments are loop-invariant, and hipcc would hoist all 28 b128 reads (112 VGPRs) out
    // of the tile loop, leaving one wave per SIMD.  An opaque zero offset keeps them as per-tile LDS
    // reads (cheap: 28 KB per 224 MFMAs) so that two waves per SIMD fit and cover each other's stalls.
    int zo = 0;
    asm volatile("" : "+v"(zo));
    const float *Zt = Zs + zo;
#pragma unroll
    for (int i = 0; i < ST + 2; ++i) {
      if (i >= 1 && i - 1 < ST) {
#pragma unroll
        for (int r = 0; r < 4; ++r) wv[i - 1][r] = UOC_EXP(kappa * Sv[i - 1][r]);
      }
      if (i < ST) Sv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (NH > 1 && i < ST) {  // second half of the dot products (not interleaved: 'cat' is the rare mode)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const int v = k >> 2, e = k & 3;
          const float4 zb = *reinterpret_cast<const float4 *>(Zt + (ST * 16 + 16 * i + t) * ZP + 16 * v + 4 * q);
          Sv[i] = mfma4(f4c(xa[4 + v], e), f4c(zb, e), Sv[i]);
        }
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (i < ST) {
          const int v = k >> 2, e = k & 3;
          const float4 zb = *reinterpret_cast<const float4 *>(Zt + (16 * i + t) * ZP + 16 * v + 4 * q);
          Sv[i] = mfma4(f4c(xa[v], e), f4c(zb, e), Sv[i]);
        }
        if (i >= 2) {
          const int r = k >> 2, ct = k & 3;
          acc[i - 2][ct] = mfma4(wv[i - 2][r], f4c(xb[r], ct), acc[i - 2][ct]);
        }
      }
    }
    if (NH == 1) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        xa[v] = na[v];
        xb[v] = nb[v];
      }
    } else {
      load_tile(tile + stride, xa, xb);  // no double buffering: the registers go to the second half of the row
    }
  }

  // ---- cross-wave reduction (fixed order: (w0 + w2) + (w1 + w3)) through LDS -------------
  f32x4 *red = reinterpret_cast<f32x4 *>(smem);  // [2][ST*4][64] f32x4
  __syncthreads();
  if (wave >= 2) {
#pragma unroll
    for (int s = 0; s < ST; ++s)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) red[((wave - 2) * ST * 4 + s * 4 + ct) * 64 + lane] = acc[s][ct];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int s = 0; s < ST; ++s)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[s][ct] += red[(wave * ST * 4 + s * 4 + ct) * 64 + lane];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int s = 0; s < ST; ++s)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) red[(s * 4 + ct) * 64 + lane] = acc[s][ct];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int s = 0; s < ST; ++s) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[s][ct] += red[(s * 4 + ct) * 64 + lane];
      // lane (t,q) reg r holds newZ[seed 16s+4q+r][channel 4t+ct]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float4 o = make_float4(acc[s][0][r], acc[s][1][r], acc[s][2][r], acc[s][3][r]);
        *reinterpret_cast<float4 *>(partial + (size_t)(16 * s + 4 * q + r) * C + 4 * t) = o;
      }
    }
  }
  __syncthreads();  // the next virtual block reloads the seeds into the buffer the reduction just used
  }  // virtual blocks
}

// -------------------------------------------------------------------------------------------
// The same iteration, register-resident formulation (64-d fields; default).
//
// What the disassembly of hc_iter_kernel shows (round 2): hipcc re-orders the source's 1:1 interleave into runs of
// 4-16 DEPENDENT MFMAs on one accumulator, issues every seed-fragment ds_read right before its first use
// (s_waitcnt lgkmcnt straight after the read: the LDS latency is exposed 14 times per pixel tile) and expands expf()
// into 14 VALU instructions, 5 of them range guards that cannot trigger for |kappa S| <= 20.  So here
//   * ONE wave per SIMD owns up to 512 registers: the seed fragments of all ST tiles (16 VGPRs each) stay in registers
//     for the whole kernel — no LDS access in the main loop at all, nothing to wait for but the pixel prefetch;
//   * the instruction order is pinned with sched_barrier(0) fences between k-steps: S-chain MFMA, accumulate MFMA
//     (another accumulator), then one slice of the exp() arithmetic of the previous seed tile in their shadow;
//   * exp() is the same arithmetic as the library's expf (x*log2e split hi/lo, v_exp, ldexp), bit-identical for the
//     in-range arguments of this kernel, without the guards: 9 VALU.
// -------------------------------------------------------------------------------------------
// exp(kappa * s) as seven single-instruction steps, so that the kernel can place them one by one (elements round-robin)
// and never issues two dependent VALU in a row.  fp32 MFMA and VALU share the SIMD's fp32 lanes on gfx950
// (scripts/mfma_shadow.hip: MFMA + K v_fma = 35.5 + 2K cycles, a v_exp 8 more, from one wave or from two), so every
// VALU instruction here is paid for in matrix-pipe time: x = kappa*s rounded like the reference's elementwise
// multiply; t = fl(x log2e); v_exp(t) (1 ulp over the whole range, |t| <= 29 here); the rounding error of t,
// e = (x*log2e_hi - t) + x*log2e_lo, re-enters as 2^e = 1 + e ln2.  Within ~1.5 ulp of exp(x); the library expf spends
// four more instructions (rndne / sub / cvt / ldexp) on a range reduction v_exp does not need and five on guards that
// cannot trigger for |x| <= 20.
// The four elements of a lane (pixels 4q .. 4q+3 of the seed tile) go through the arithmetic steps as TWO PAIRS on
// v_pk_mul_f32 / v_pk_fma_f32 (the same IEEE operation per element as the scalar instruction: bit-identical), v_exp per
// element: 16 VALU instructions per seed tile instead of 28.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct ExpState {
  f32x2 x, t, e, p;
};
constexpr int EXP_OPS = 16;   // per seed tile: ops 0-1 x = kappa s | 2-3 t | 4-5, 6-7 e | 8-11 v_exp | 12-13 e ln2 | 14-15 w
#pragma clang fp contract(off)
template <int OP>
__device__ __forceinline__ void exp_op(ExpState (&es)[2], const f32x4 &s, float kappa, float (&w)[4]) {
  constexpr int P = OP & 1;   // the pair (elements 2P, 2P + 1) of the packed ops
  ExpState &e = es[P];
  if (OP < 2) e.x = f32x2{kappa, kappa} * f32x2{s[2 * P], s[2 * P + 1]};
  else if (OP < 4) e.t = e.x * f32x2{0x1.715476p+0f, 0x1.715476p+0f};                                   // log2(e) hi (0x3fb8aa3b)
  else if (OP < 6) e.e = __builtin_elementwise_fma(e.x, f32x2{0x1.715476p+0f, 0x1.715476p+0f}, -e.t);
  else if (OP < 8) e.e = __builtin_elementwise_fma(e.x, f32x2{0x1.4ae0bep-26f, 0x1.4ae0bep-26f}, e.e);   // log2(e) lo (0x32a5705f)
  else if (OP < 12) {
    constexpr int r = OP - 8;
    es[r >> 1].p[r & 1] = __builtin_amdgcn_exp2f(es[r >> 1].t[r & 1]);
  } else if (OP < 14) e.e = e.e * f32x2{0x1.62e43p-1f, 0x1.62e43p-1f};                                   // ln 2
  else {
    const f32x2 v = __builtin_elementwise_fma(e.p, e.e, e.p);
    w[2 * P] = v.x;
    w[2 * P + 1] = v.y;
  }
}
#pragma clang fp contract(fast)


// One pixel tile (16 pixels) against all ST seed tiles.  3-stage pipeline over the seed tiles, fully unrolled; step i:
//   1. the S-chain of tile i: 16 DEPENDENT MFMAs on one accumulator, strictly back to back — consecutive accumulation
//      into the same registers runs at the matrix pipe's full 32-cycle rate, but any instruction between two of them
//      (another MFMA, a VALU) costs 4-9 cycles per MFMA (scripts/mfma_mix.hip: S A S A 36.0, S v A v 40.5, against
//      32.0 for 16 S then 16 A);
//   2. the 16 independent accumulate MFMAs of tile i-2 (four accumulators in rotation), each followed by two or
//      three single VALU steps of the exp() of tile i-1, elements round-robin so no VALU waits on the one before it
//      (a VALU behind an independent MFMA costs ~1.7 cycles of issue, same microbenchmark: 35.3).
// sched_barrier(0) after every slot keeps hipcc from regrouping.
// QUAD: the last seed tile holds at most 4 seeds (m = 100 = 6 x 16 + 4) and runs on v_mfma_f32_4x4x1_16B_f32 — sixteen
// independent 4x4 outer products per instruction, an eighth of the 16x16x4 instruction's time — instead of a padded
// 16-seed tile (12 % of the kernel's MFMAs were that padding).  With block = (channel phase t/4, pixel group q):
//   S:    A = xc (lane (t,q): pixel 4q + t%4, channel 16(t/4) + k), B = zb[last] loaded as Z[seed 16(ST-1) + t%4] in the
//         same channels; D reg r = the partial dot product of pixel 4q+r with seed t%4 over the block's 16 channels; the
//         four channel phases of a row meet through two DPP row rotations ((p0 + p2) + (p1 + p3) on every lane).
//   The result has the layout of a regular tile's S (reg r <-> pixel 4q+r) with seed t%4 in place of seed t, so exp()
//   and the accumulate step keep their code: acc[last][ct] += W[pixel 4q+r][seed i] X[pixel 4q+r][chan 4t+ct] with
//   block = (channel group t/4, pixel group q), summed over the four pixel groups once per virtual block.
template <int ST, bool QUAD, int I>
__device__ __forceinline__ void hcr_step(const float4 (&xa)[4], const float4 (&xb)[4], const float4 (&xc)[4],
                                         const float4 (&zb)[ST][4], f32x4 (&acc)[ST][4], f32x4 (&Sv)[ST + 2],
                                         float (&wv)[ST + 2][4], ExpState (&es)[2], float kappa) {
  // Sv / wv carry two spare rows so that the (never executed) I-1 / I-2 references of the first steps stay in range
  constexpr bool do_s = I < ST, do_a = I >= 2, do_e = I >= 1 && I - 1 < ST;
  constexpr int IS = I < ST ? I : 0, IE = I >= 1 ? I - 1 : 0, IA = I >= 2 ? I - 2 : 0;
  constexpr bool quad_s = QUAD && IS == ST - 1, quad_a = QUAD && IA == ST - 1;
  if (do_s) {
    Sv[IS] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int v = k >> 2, e = k & 3;
      if (quad_s)
        Sv[IS] = __builtin_amdgcn_mfma_f32_4x4x1f32(f4c(xc[v], e), f4c(zb[IS][v], e), Sv[IS], 0, 0, 0);
      else
        Sv[IS] = mfma4(f4c(xa[v], e), f4c(zb[IS][v], e), Sv[IS]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (quad_s) {   // the four channel phases (lanes t, t+4, t+8, t+12 of a row): identical sum order on every lane
#pragma unroll
      for (int r = 0; r < 4; ++r) Sv[IS][r] += dpp_f<0x128>(Sv[IS][r]);   // row_ror:8
#pragma unroll
      for (int r = 0; r < 4; ++r) Sv[IS][r] += dpp_f<0x124>(Sv[IS][r]);   // row_ror:4
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  constexpr int nops = do_e ? EXP_OPS : 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (do_a) {
      const int r = k >> 2, ct = k & 3;
      if (quad_a)
        acc[IA][ct] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[IA][r], f4c(xb[r], ct), acc[IA][ct], 0, 0, 0);
      else
        acc[IA][ct] = mfma4(wv[IA][r], f4c(xb[r], ct), acc[IA][ct]);
    }
    // the VALU ops that belong behind this slot, in 4 clusters per seed tile (behind MFMAs 3, 7, 11, 15): every
    // MFMA -> VALU switch costs ~2.7 cycles on top of the VALU's own 2 (scripts/mfma_shadow.hip)
    const int c = k >> 2, o0 = (k & 3) == 3 ? c * nops / 4 : 0, o1 = (k & 3) == 3 ? (c + 1) * nops / 4 : 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {           // 16 / 4 = 4 ops per cluster; fixed trip count so it unrolls
      const int o = o0 + j;
      if (o >= o1) continue;
      switch (o) {                          // pair-minor within a step: consecutive ops are independent
        case 0: exp_op<0>(es, Sv[IE], kappa, wv[IE]); break;
        case 1: exp_op<1>(es, Sv[IE], kappa, wv[IE]); break;
        case 2: exp_op<2>(es, Sv[IE], kappa, wv[IE]); break;
        case 3: exp_op<3>(es, Sv[IE], kappa, wv[IE]); break;
        case 4: exp_op<4>(es, Sv[IE], kappa, wv[IE]); break;
        case 5: exp_op<5>(es, Sv[IE], kappa, wv[IE]); break;
        case 6: exp_op<6>(es, Sv[IE], kappa, wv[IE]); break;
        case 7: exp_op<7>(es, Sv[IE], kappa, wv[IE]); break;
        case 8: exp_op<8>(es, Sv[IE], kappa, wv[IE]); break;
        case 9: exp_op<9>(es, Sv[IE], kappa, wv[IE]); break;
        case 10: exp_op<10>(es, Sv[IE], kappa, wv[IE]); break;
        case 11: exp_op<11>(es, Sv[IE], kappa, wv[IE]); break;
        case 12: exp_op<12>(es, Sv[IE], kappa, wv[IE]); break;
        case 13: exp_op<13>(es, Sv[IE], kappa, wv[IE]); break;
        case 14: exp_op<14>(es, Sv[IE], kappa, wv[IE]); break;
        default: exp_op<15>(es, Sv[IE], kappa, wv[IE]); break;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int ST, bool QUAD, int... Is>
__device__ __forceinline__ void hcr_tile_steps(const float4 (&xa)[4], const float4 (&xb)[4], const float4 (&xc)[4],
                                               const float4 (&zb)[ST][4], f32x4 (&acc)[ST][4], float kappa,
                                               std::integer_sequence<int, Is...>) {
  f32x4 Sv[ST + 2];
  float wv[ST + 2][4];
  ExpState es[2];
  (hcr_step<ST, QUAD, Is>(xa, xb, xc, zb, acc, Sv, wv, es, kappa), ...);
}

template <int ST, bool QUAD>
__device__ __forceinline__ void hcr_tile(const float4 (&xa)[4], const float4 (&xb)[4], const float4 (&xc)[4],
                                         const float4 (&zb)[ST][4], f32x4 (&acc)[ST][4], float kappa) {
  hcr_tile_steps<ST, QUAD>(xa, xb, xc, zb, acc, kappa, std::make_integer_sequence<int, ST + 2>{});
}

// THE SHIPPED KERNEL: one wave per SIMD (4 waves per block, one block per CU), every wave all ST seed tiles (~330 of
// its 512 registers), X read exactly once.  Cost model that fits the measurements: a pixel tile costs 32 cycles per MFMA
// (224) + 2 per VALU (~230) + 8 per v_exp (28) + ~2.7 per MFMA->VALU switch, i.e. the floor of this formulation is
// ~8 000 cycles per tile against 7 168 of pure MFMA.  The 112 KB of LDS serve the cross-wave reductions only.
//
// VIRTUAL blocks: a field is always cut into nvb blocks of 4 waves (nvb depends on n only, hc_virtual_blocks).  Which
// pixel tiles meet in which partial sum, and the order of every fp32 addition, therefore do not depend on how many
// fields share the launch (batch), on the CU count or on the grid.
//
// FLAT ITEM SCHEDULE (round 6).  The launch's batch x nvb virtual blocks form ONE list that a 1-D grid of at most one
// block per CU works through (HcPlan):
//   * items [0, n1) run whole, a block staying inside one field where the counts allow it, else taking items g, g + grid,
//     ... (when the field changes the seed fragments are reloaded under the previous item's reduction);
//   * the TAIL items [n1, items) — what is left after the full rounds — run as `parts` SEED-TILE parts each (7 seed tiles
//     = 3 + 4, 2 + 2 + 3 or 5 x 1 + 2, the 4-seed tile riding with the last part).  A part walks the virtual block's pixel tiles exactly as the whole
//     item would and accumulates ITS seed tiles only: every accumulator sees the same MFMAs in the same order, so the
//     partial sums are bit-identical whatever the split (tests/test_meanshift_gpu.py), while 87 left-over virtual blocks
//     (7 crops = 343 on 256 CUs) occupy 174 CUs for 4/7 of a round instead of 87 CUs for a whole one.  Parts are listed
//     largest first and dealt out forwards, then backwards (two passes at most): with sorted items that IS the
//     longest-processing-time rule.  The price is reading the tail's pixels `parts` times (from L2).
// The plan is a pure function of (batch, nvb, CU count) evaluated on the host by makespan (hc_make_plan).
// The walk is ONE software pipeline: the first pixel tile of the next item is already in flight while the current one is
// reduced through LDS and stored (per item that leaves the reduction itself, ~2 us against ~56 us of tiles).
// (Round 2's two-waves-per-SIMD variant with the seeds split between the waves measured the same 87.5 vs 88.0 us and
// fetched X twice; it was removed in round 3 — fp32 MFMA and VALU share the SIMD's lanes, DESIGN.md.)
struct HcPlan {
  int nvb;     // virtual blocks per field
  int items;   // batch * nvb
  int n1;      // items [0, n1) run whole
  int parts;   // every item in [n1, items) runs as this many seed-tile parts (1: n1 == items)
  int grid;    // physical blocks
};
struct HcGeom {
  const float *X;   // [batch][n][64]
  int n, ntile, stride;
  int lane, wave, t, q;
};
// Pixel tiles: branch-free loads (clamped row index).  A clamped xa row only produces a finite S / W for a pixel
// whose xb row is zeroed, so out-of-range pixels contribute exactly 0; xb is zeroed only in the (rare) partial tile.
// Rows are addressed from the launch's base X by a 32-bit GLOBAL pixel index (field b starts at row b * n; batch * n
// rows of 256 bytes always fit: 2^31 rows would be 550 GB), so the field of a tile is a scalar offset, not a pointer.
__device__ __forceinline__ void hc_load_tile(const HcGeom &ge, int b, int tl, float4 (&a)[4], float4 (&bb)[4]) {
  const int row0 = b * ge.n;
  const int base = row0 + min(tl, ge.ntile - 1) * 16, last = row0 + ge.n - 1;
  const float *__restrict__ X = ge.X;
  const int pa = min(base + ge.t, last);
#pragma unroll
  for (int v = 0; v < 4; ++v) a[v] = *reinterpret_cast<const float4 *>(X + (size_t)(unsigned)pa * C + 16 * v + 4 * ge.q);
  __builtin_amdgcn_sched_barrier(0);   // these four leave before the other view's address arithmetic (all loads behind
                                       // all of it: 551 vs 546 us at 150 pixel tiles per wave)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int pb = min(base + 4 * ge.q + r, last);
    bb[r] = *reinterpret_cast<const float4 *>(X + (size_t)(unsigned)pb * C + 4 * ge.t);
  }
}
// Third view of a tile for the 4x4x1 S step: pixel 4q + t%4, channels 16(t/4) + 4v .. +3.  It feeds the LAST seed tile,
// 60 % into the pixel tile, so it is loaded for the CURRENT tile at its top (its lines arrived with the other two views a
// tile earlier) instead of being prefetched and handed over: 16 VALU moves and 16 registers less per pixel tile.
__device__ __forceinline__ void hc_load_quad_view(const HcGeom &ge, int b, int tl, float4 (&cc)[4]) {
  const int row0 = b * ge.n;
  const int base = row0 + min(tl, ge.ntile - 1) * 16, last = row0 + ge.n - 1;
  const int pc = min(base + 4 * ge.q + (ge.t & 3), last);
#pragma unroll
  for (int v = 0; v < 4; ++v) cc[v] = *reinterpret_cast<const float4 *>(ge.X + (size_t)(unsigned)pc * C + 16 * (ge.t >> 2) + 4 * v);
}
// applied when the tile becomes the current one (a select on freshly loaded data would force a wait at the load)
__device__ __forceinline__ void hc_mask_tile(const HcGeom &ge, int tl, float4 (&bb)[4]) {
  const int base = min(tl, ge.ntile - 1) * 16;
  if (base + 16 > ge.n) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (base + 4 * ge.q + r >= ge.n) bb[r] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
// seed fragments of ST tiles starting at row 0 of Zs (m rows are valid): zb[i][v] = Zs[seed 16i+t][16v+4q .. +3]
// (B operand of S^T = X Z^T), zero rows beyond m.  Unconditional loads from a clamped row + a select: a predicated
// load would get its own branch and its own s_waitcnt vmcnt(0), i.e. ST serialised memory latencies.
template <int ST, bool QUAD>
__device__ __forceinline__ void hc_seed_loads(const HcGeom &ge, const float *__restrict__ Zs, int m, float4 (&zb)[ST][4]) {
  const int t = ge.t, q = ge.q;
#pragma unroll
  for (int i = 0; i < ST; ++i)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (QUAD && i == ST - 1)   // the 4x4x1 tile: seed 16(ST-1) + t%4, channels 16(t/4) + 4v .. +3
        zb[i][v] = *reinterpret_cast<const float4 *>(Zs + (size_t)min(16 * i + (t & 3), m - 1) * C + 16 * (t >> 2) + 4 * v);
      else
        zb[i][v] = *reinterpret_cast<const float4 *>(Zs + (size_t)min(16 * i + t, m - 1) * C + 16 * v + 4 * q);
    }
}
// Rows beyond m can only sit in the LAST of the ST tiles (ST = ceil(m / 16) for a whole item; a part either ends with the
// launch's last tile or holds full tiles only).
template <int ST, bool QUAD>
__device__ __forceinline__ void hc_seed_mask(const HcGeom &ge, int m, float4 (&zb)[ST][4]) {
  const bool beyond = 16 * (ST - 1) + (QUAD ? (ge.t & 3) : ge.t) >= m;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    zb[ST - 1][v].x = beyond ? 0.f : zb[ST - 1][v].x;
    zb[ST - 1][v].y = beyond ? 0.f : zb[ST - 1][v].y;
    zb[ST - 1][v].z = beyond ? 0.f : zb[ST - 1][v].z;
    zb[ST - 1][v].w = beyond ? 0.f : zb[ST - 1][v].w;
  }
}

// All-zero accumulators from the matrix pipe (0 x 0 + 0): 28 MFMAs that run while the wave waits at a barrier or for its
// first pixels.  112 v_accvgpr_write cost VALU issue — hipcc put them (twice) at the head of every item, 2 us per item.
template <int ST>
__device__ __forceinline__ void hc_zero_acc(f32x4 (&acc)[ST][4]) {
  float z = 0.f;
  asm volatile("" : "+v"(z));   // opaque: mfma(0, 0, 0) must not fold into register writes
#pragma unroll
  for (int s = 0; s < ST; ++s)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[s][ct] = mfma4(z, z, f32x4{0.f, 0.f, 0.f, 0.f});
}

// ONE item: the pixel tiles of virtual block vb of field b against the ST seed tiles whose fragments are in zb; the
// partial sums go to dst (row 0 = the first of these seeds, 64 floats per row).  xa / xb / xc hold the item's first
// tile on entry and the first tile of the NEXT item (field nb, tile nt) on exit.  Znext != nullptr: the next item belongs
// to another field — its seed fragments are requested as soon as the last pixel tile is done and arrive under the
// cross-wave reduction.
template <int ST, bool QUAD, bool KQ>
__device__ __forceinline__ void hc_item(const HcGeom &ge, float4 (&zb)[ST][4], float kappa, int b, int vb, int nb,
                                        int nt, float4 (&xa)[4], float4 (&xb)[4], float4 (&xc)[4], int &held_b,
                                        int &held_t, f32x4 (&acc)[ST][4], f32x4 *red, float *__restrict__ dst,
                                        const float *__restrict__ Znext, int m) {
  // acc: all zero on entry, all zero again on exit
  const int lane = ge.lane, wave = ge.wave, t = ge.t, q = ge.q;
  f32x4 *red_wave = red + (size_t)wave * ST * 4 * 64;
  // The loads of the wave's NEXT tile (the first tile of the next item behind the last one of this item) are issued
  // before the current tile's MFMAs and first read after them.  (hipcc undoes a plain `xa = na` double buffer: it
  // coalesces the copy, rotates the loop and ends up with load-then-wait at the top of every tile, ~0.8 us exposed.)
  // Ping-pong: a tile reads its pixels from one register set and prefetches the next tile into the other; the loop body
  // holds two tiles with the roles swapped, so no registers are handed over per tile (32 VALU moves only when an item ends
  // on the first half of the body).  The sched_barrier keeps each half's loads above its first MFMA.
  float4 ya[4], yb[4];
  auto one_tile = [&](int tile, float4(&ca)[4], float4(&cb)[4], float4(&na)[4], float4(&nbv)[4]) {
    const bool more = tile + ge.stride < ge.ntile;   // invariant: (held_b, held_t) == (b, tile), pixels in ca / cb
    const int nxt = more ? tile + ge.stride : nt;
    if (KQ) hc_load_quad_view(ge, b, tile, xc);
    hc_load_tile(ge, more ? b : nb, nxt, na, nbv);
    __builtin_amdgcn_sched_barrier(0);
    hcr_tile<ST, QUAD>(ca, cb, xc, zb, acc, kappa);
    __builtin_amdgcn_sched_barrier(0);
    hc_mask_tile(ge, nxt, nbv);
    held_b = more ? b : nb;
    held_t = nxt;
  };
  for (int tile = vb * 4 + wave; tile < ge.ntile; tile += ge.stride) {
    one_tile(tile, xa, xb, ya, yb);
    tile += ge.stride;
    if (tile >= ge.ntile) {        // the next item's first tile sits in the second set: hand it over
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                     : "=&v"(xa[v].x), "=&v"(xa[v].y), "=&v"(xa[v].z), "=&v"(xa[v].w)
                     : "v"(ya[v].x), "v"(ya[v].y), "v"(ya[v].z), "v"(ya[v].w));
        asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                     : "=&v"(xb[v].x), "=&v"(xb[v].y), "=&v"(xb[v].z), "=&v"(xb[v].w)
                     : "v"(yb[v].x), "v"(yb[v].y), "v"(yb[v].z), "v"(yb[v].w));
      }
      break;
    }
    one_tile(tile, ya, yb, xa, xb);
  }
  if (Znext) hc_seed_loads<ST, QUAD>(ge, Znext, m, zb);
  // ---- the item is complete: the four waves' accumulators meet in LDS, (w0 + w1) + (w2 + w3) ----
  if (QUAD) {   // the 4x4x1 tile's accumulators are partial over the pixel groups q: lanes l, l^16, l^32, l^48 -> (q0+q1)+(q2+q3)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[ST - 1][ct][r];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        acc[ST - 1][ct][r] = v;      // every lane now holds the total; lanes q = 0 are seeds 16(ST-1) .. +3, the rest is
      }                              // written to rows >= m of the partial, which nobody reads
  }
#pragma unroll
  for (int s = 0; s < ST; ++s)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) red_wave[(s * 4 + ct) * 64 + lane] = acc[s][ct];
  hc_zero_acc<ST>(acc);   // on the matrix pipe, under the barrier
  __syncthreads();
  for (int s = wave; s < ST; s += 4) {
    f32x4 a[4][4], o[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int w = 0; w < 4; ++w) a[ct][w] = red[((w * ST + s) * 4 + ct) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);   // all 16 reads in flight before the first add (hipcc paired each read with its wait)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) o[ct] = (a[ct][0] + a[ct][1]) + (a[ct][2] + a[ct][3]);
    // lane (t,q) reg r holds newZ[seed 16s+4q+r][channel 4t+ct]
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<float4 *>(dst + (size_t)(16 * s + 4 * q + r) * C + 4 * t) =
          make_float4(o[0][r], o[1][r], o[2][r], o[3][r]);
  }
  __syncthreads();  // the reduction buffer is written again by the next item
  if (Znext) hc_seed_mask<ST, QUAD>(ge, m, zb);
}

// A tail part: seed tiles [s0, s0 + ST) of the launch's tiles, with its own fragments and its own first pixel tile (no
// hand-over of registers between the bodies: hipcc then keeps the whole-item loop's live ranges to itself — with the
// next item's pixels carried across the bodies it parked them in AGPRs and paid 66 extra VALU moves per pixel tile).
template <int ST, bool QUAD>
__device__ __forceinline__ void hc_part(const HcGeom &ge, const float *__restrict__ Zb, int m, int s0, float kappa, int b,
                                        int vb, f32x4 *red, float *__restrict__ dst_item) {
  float4 zb[ST][4];
  hc_seed_loads<ST, QUAD>(ge, Zb + (size_t)16 * s0 * C, m - 16 * s0, zb);
  float4 xa[4], xb[4], xc[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) xc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  int held_b = b, held_t = vb * 4 + ge.wave;
  hc_load_tile(ge, b, held_t, xa, xb);
  hc_seed_mask<ST, QUAD>(ge, m - 16 * s0, zb);
  hc_mask_tile(ge, held_t, xb);
  f32x4 acc[ST][4];
  hc_zero_acc<ST>(acc);
  hc_item<ST, QUAD, QUAD>(ge, zb, kappa, b, vb, b, held_t, xa, xb, xc, held_b, held_t, acc, red,
                          dst_item + (size_t)16 * s0 * C, nullptr, 0);
}

template <int ST, bool QUAD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void hc_iter_flat_kernel(
    const float *__restrict__ X, int n, const float *__restrict__ Z, int m, float kappa, float *__restrict__ partial,
    HcPlan plan) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4 *red = reinterpret_cast<f32x4 *>(smem);  // [4 waves][ST*4][64] f32x4
  const int tid = threadIdx.x;
  HcGeom ge;
  ge.X = X;
  ge.n = n;
  ge.ntile = (n + 15) >> 4;
  ge.stride = plan.nvb * 4;
  ge.lane = tid & 63;
  ge.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  ge.t = ge.lane & 15;
  ge.q = ge.lane >> 4;
  const int g = blockIdx.x;
  constexpr bool SPLIT = ST == 7 && QUAD;   // the seed-tile parts exist for the 97..100-seed launches (the reference's 100)

  // ---- whole items.  If every block can stay inside ONE field (q items per block, q | nvb, whole fields only) block
  //      (field f, x) takes the virtual blocks x, x + nvb/q, ...: the pixel tiles of its items then share their pages
  //      (a wave's tiles are 4 MB apart at 480x640, the next item's 1 MB further on).  Otherwise block g takes items g,
  //      g + grid, ... (4 x 480x640: 294 us against 286 — every item in another field; still dense across the grid,
  //      where a contiguous chunk per block had 256 blocks read 4 KB pieces 16 KB apart) ----
  if (g < plan.n1) {
    const int q = plan.n1 / plan.grid;
    const bool affine = q >= 1 && q * plan.grid == plan.n1 && plan.n1 % plan.nvb == 0 && plan.nvb % q == 0;
    const int per = affine ? plan.nvb / q : 0;                       // blocks per field
    const int step = affine ? per : plan.grid;                       // item stride of a block
    const int first = affine ? (g / per) * plan.nvb + g % per : g;   // its first item
    const int count = affine ? q : (plan.n1 - g + plan.grid - 1) / plan.grid;
    int b = first / plan.nvb, vb = first % plan.nvb;
    float4 xa[4], xb[4], xc[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) xc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    int held_b = b, held_t = vb * 4 + ge.wave;   // the tile whose pixels sit in xa / xb / xc
    hc_load_tile(ge, b, held_t, xa, xb);
    float4 zb[ST][4];
    hc_seed_loads<ST, QUAD>(ge, Z + (size_t)b * m * C, m, zb);
    hc_mask_tile(ge, held_t, xb);
    hc_seed_mask<ST, QUAD>(ge, m, zb);
    f32x4 acc[ST][4];
    hc_zero_acc<ST>(acc);
    for (int j = 0; j < count; ++j) {
      const int item = first + j * step;
      const int nitem = j + 1 < count ? item + step : item;   // behind the last one: a harmless reload
      const int nb = nitem / plan.nvb, nvb_ = nitem % plan.nvb;
      // (only a wave that had no tile in the previous item — tiny fields — does not hold this item's first tile yet)
      if (held_b != b || held_t != vb * 4 + ge.wave) {
        held_b = b;
        held_t = vb * 4 + ge.wave;
        hc_load_tile(ge, b, held_t, xa, xb);
        hc_mask_tile(ge, held_t, xb);
      }
      hc_item<ST, QUAD, QUAD>(ge, zb, kappa, b, vb, nb, nvb_ * 4 + ge.wave, xa, xb, xc, held_b, held_t, acc, red,
                              partial + ((size_t)b * plan.nvb + vb) * (ST * 16) * C,
                              nb != b ? Z + (size_t)nb * m * C : nullptr, m);
      b = nb;
      vb = nvb_;
    }
  }
  // ---- tail items as seed-tile parts: largest parts first, dealt forwards, then backwards ----
  if constexpr (SPLIT) {
    const int R = plan.items - plan.n1, T = R * plan.parts;
    for (int k = 0; k < 2; ++k) {
      const int tpos = k * plan.grid + ((k & 1) ? plan.grid - 1 - g : g);
      if (tpos >= T) break;
      const int p = tpos / R, item = plan.n1 + tpos % R;
      const int b = item / plan.nvb, vb = item % plan.nvb;
      const float *Zb = Z + (size_t)b * m * C;
      float *dst = partial + ((size_t)b * plan.nvb + vb) * (ST * 16) * C;
#define UOC_HC_PART(SUB, Q4, S0) hc_part<SUB, Q4>(ge, Zb, m, S0, kappa, b, vb, red, dst)
      // seed tiles 0..5 are full, tile 6 holds the last <= 4 seeds (a tenth of a full tile's time): balanced splits
      const int P = plan.parts;
      if (P == 2) {                                   // {0 1 2} {3 4 5 6}
        if (p == 0) UOC_HC_PART(4, true, 3); else UOC_HC_PART(3, false, 0);
      } else if (P == 3) {                            // {0 1} {2 3} {4 5 6}
        if (p == 0) UOC_HC_PART(3, true, 4); else UOC_HC_PART(2, false, 2 * (p - 1));
      } else {                                        // 6: {0} {1} {2} {3} {4} {5 6}
        if (p == 0) UOC_HC_PART(2, true, 5); else UOC_HC_PART(1, false, p - 1);
      }
#undef UOC_HC_PART
    }
  }
}

// Z[seed] = normalize(sum_blk partial[blk][seed])  (F.normalize, eps 1e-12; mean_shift.py:107)
// partial [b][blk][NH][rows][64] -> Z [b][NH][m][64], the norm runs over all NH * 64 channels.
// One block per seed: 16 lanes x float4 cover a 256-byte row, so a wave reads the rows of 4 blocks per load and
// the 4 waves keep 64 rows in flight per step (4 independent accumulators); fixed summation order.
template <int NH>
__global__ __launch_bounds__(256) void hc_finalize_kernel(const float *__restrict__ partial, int nblk, int rows,
                                                          int m, float *__restrict__ Z) {
  const int b = blockIdx.y, seed = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c4 = lane & 15, sub = lane >> 4;
  const int slot = w * 4 + sub;  // 0..15: which of 16 concurrently read block rows
  __shared__ float4 red[NH][4][16];
  auto add4 = [](float4 a, float4 c) { return make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w); };
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const float *src = partial + (((size_t)b * nblk * NH + h) * rows + seed) * C + 4 * c4;
    const size_t bs = (size_t)NH * rows * C;  // block stride
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    auto ld = [&](int blk) { return *reinterpret_cast<const float4 *>(src + (size_t)blk * bs); };
    int blk = slot;
    for (; blk + 48 < nblk; blk += 64) {
      s0 = add4(s0, ld(blk));
      s1 = add4(s1, ld(blk + 16));
      s2 = add4(s2, ld(blk + 32));
      s3 = add4(s3, ld(blk + 48));
    }
    for (; blk < nblk; blk += 16) s0 = add4(s0, ld(blk));
    float4 s = add4(add4(s0, s1), add4(s2, s3));
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      s.x += __shfl_xor(s.x, off);
      s.y += __shfl_xor(s.y, off);
      s.z += __shfl_xor(s.z, off);
      s.w += __shfl_xor(s.w, off);
    }
    if (sub == 0) red[h][w][c4] = s;
  }
  __syncthreads();
  if (w == 0) {  // all 64 lanes take part in the shuffles; lanes with sub != 0 mirror sub 0
    float4 v[NH];
    float ss = 0.f;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      v[h] = add4(add4(red[h][0][c4], red[h][1][c4]), add4(red[h][2][c4], red[h][3][c4]));
      ss = fmaf(v[h].x, v[h].x, ss);
      ss = fmaf(v[h].y, v[h].y, ss);
      ss = fmaf(v[h].z, v[h].z, ss);
      ss = fmaf(v[h].w, v[h].w, ss);
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);  // over the 16 channel groups
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);
    if (sub == 0) {
#pragma unroll
      for (int h = 0; h < NH; ++h)
        *reinterpret_cast<float4 *>(Z + (((size_t)b * NH + h) * m + seed) * C + 4 * c4) =
            make_float4(v[h].x / nrm, v[h].y / nrm, v[h].z / nrm, v[h].w / nrm);
    }
  }
}

// -------------------------------------------------------------------------------------------
// Seed connected components: inherently sequential over seeds (mean_shift.py:53-74), m <= 128,
// so ONE wavefront per batch item; lane l owns seeds l and l+64.  Quirks kept: the component
// takes the MODE of already-present labels (ties -> smallest) and overwrites every member.
// -------------------------------------------------------------------------------------------
template <int NH>
__global__ __launch_bounds__(64) void seed_cc_kernel(const float *__restrict__ Z, int m, float eps,
                                                     int *__restrict__ seed_labels, int *__restrict__ num_unique) {
  constexpr int CW = NH * C;  // a seed row in LDS: the NH halves back to back, pitch CW + 1
  extern __shared__ __attribute__((aligned(16))) float Zs[];  // [NLAB][CW + 1]
  const int b = blockIdx.x, lane = threadIdx.x;
  Z += (size_t)b * NH * m * C;
  for (int i = lane; i < NH * m * C; i += 64) {
    const int h = i / (m * C), j = i % (m * C);
    Zs[(j / C) * (CW + 1) + h * C + (j % C)] = Z[i];
  }
  __syncthreads();
  const bool have0 = lane < m, have1 = lane + 64 < m;
  int lab0 = -1, lab1 = -1, next = 0;
  for (int i = 0; i < m; ++i) {
    const int li = (i < 64) ? __shfl(lab0, i) : __shfl(lab1, i - 64);
    if (li != -1) continue;
    float dot0 = 0.f, dot1 = 0.f;
    const float *zi = Zs + i * (CW + 1);
    const float *z0 = Zs + lane * (CW + 1);
    const float *z1 = Zs + (lane + 64) * (CW + 1);
    if (have0)
      for (int c = 0; c < CW; ++c) dot0 = fmaf(z0[c], zi[c], dot0);
    if (have1)
      for (int c = 0; c < CW; ++c) dot1 = fmaf(z1[c], zi[c], dot1);
    const bool in0 = have0 && (0.5f * (1.0f - dot0) <= eps);
    const bool in1 = have1 && (0.5f * (1.0f - dot1) <= eps);
    unsigned long long lm0 = __ballot(in0 && lab0 != -1), lm1 = __ballot(in1 && lab1 != -1);
    const bool any_unl = (__ballot(in0 && lab0 == -1) | __ballot(in1 && lab1 == -1)) != 0ull;
    // distinct values among members' labels (the value -1 counts as one, :66)
    int best_cnt = 0, best_lab = INT_MAX, distinct = any_unl ? 1 : 0;
    while (lm0 | lm1) {
      const int L = lm0 ? __shfl(lab0, __ffsll((long long)lm0) - 1) : __shfl(lab1, __ffsll((long long)lm1) - 1);
      const unsigned long long c0 = __ballot(in0 && lab0 == L), c1 = __ballot(in1 && lab1 == L);
      const int cnt = __popcll(c0) + __popcll(c1);
      if (cnt > best_cnt || (cnt == best_cnt && L < best_lab)) {
        best_cnt = cnt;
        best_lab = L;
      }
      ++distinct;
      lm0 &= ~c0;
      lm1 &= ~c1;
    }
    int lab;
    if (distinct > 1) {
      lab = best_lab;
    } else {
      lab = next++;
    }
    if (in0) lab0 = lab;
    if (in1) lab1 = lab;
  }
  if (have0) seed_labels[(size_t)b * m + lane] = lab0;
  if (have1) seed_labels[(size_t)b * m + lane + 64] = lab1;
  // len(torch.unique(seed_labels)) (mean_shift.py:218)
  unsigned long long r0 = __ballot(have0), r1 = __ballot(have1);
  int uniq = 0;
  while (r0 | r1) {
    const int L = r0 ? __shfl(lab0, __ffsll((long long)r0) - 1) : __shfl(lab1, __ffsll((long long)r1) - 1);
    r0 &= ~__ballot(have0 && lab0 == L);
    r1 &= ~__ballot(have1 && lab1 == L);
    ++uniq;
  }
  if (lane == 0) num_unique[b] = uniq;
}

// -------------------------------------------------------------------------------------------
// Nearest-seed assignment: S = X Z^T on fp32 MFMA, d = 0.5(1 - S), argmin over seeds
// (ties -> lowest seed index, torch.argmin), label = seed_labels[argmin], per-label histogram.
// -------------------------------------------------------------------------------------------
template <int ST, int NH>
__global__ __launch_bounds__(HC_THREADS) void assign_kernel(const float *__restrict__ X, int n,
                                                            const float *__restrict__ Z,
                                                            const int *__restrict__ seed_labels, int m,
                                                            int *__restrict__ labels, int *__restrict__ closest,
                                                            int *__restrict__ counts) {
  extern __shared__ __attribute__((aligned(16))) float Zs[];  // [NH][ST * 16][ZP]
  __shared__ int slab[NLAB];
  __shared__ int hist[NLAB];
  const int b = blockIdx.y;
  X += (size_t)b * NH * n * C;
  Z += (size_t)b * NH * m * C;
  seed_labels += (size_t)b * m;
  labels += (size_t)b * n;
  if (closest) closest += (size_t)b * n;
  counts += (size_t)b * NLAB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = lane & 15, q = lane >> 4;
  for (int i = tid; i < NH * ST * 16 * (C / 4); i += HC_THREADS) {
    const int h = i / (ST * 16 * (C / 4)), j = i % (ST * 16 * (C / 4));
    const int row = j / (C / 4), c4 = j % (C / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < m) v = *reinterpret_cast<const float4 *>(Z + ((size_t)h * m + row) * C + 4 * c4);
    *reinterpret_cast<float4 *>(Zs + (h * ST * 16 + row) * ZP + 4 * c4) = v;
  }
  if (tid < NLAB) {
    slab[tid] = tid < m ? seed_labels[tid] : 0;
    hist[tid] = 0;
  }
  __syncthreads();

  const int ntile = (n + 15) >> 4;
  for (int tile = blockIdx.x * (HC_THREADS / 64) + wave; tile < ntile; tile += gridDim.x * (HC_THREADS / 64)) {
    const int pa = tile * 16 + t;
    float4 xa[NH * 4];
#pragma unroll
    for (int v = 0; v < NH * 4; ++v)
      xa[v] = (pa < n) ? *reinterpret_cast<const float4 *>(X + ((size_t)(v >> 2) * n + pa) * C + 16 * (v & 3) + 4 * q)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
    float bd[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    int bi[4] = {INT_MAX, INT_MAX, INT_MAX, INT_MAX};
#pragma unroll
    for (int s = 0; s < ST; ++s) {
      f32x4 S = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int v = 0; v < NH * 4; ++v) {
        const float4 zb =
            *reinterpret_cast<const float4 *>(Zs + ((v >> 2) * ST * 16 + 16 * s + t) * ZP + 16 * (v & 3) + 4 * q);
        S = mfma4(xa[v].x, zb.x, S);
        S = mfma4(xa[v].y, zb.y, S);
        S = mfma4(xa[v].z, zb.z, S);
        S = mfma4(xa[v].w, zb.w, S);
      }
      const int seed = 16 * s + t;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = 0.5f * (1.0f - S[r]);
        if (seed < m && d < bd[r]) {  // s ascends: strict '<' keeps the lowest seed index
          bd[r] = d;
          bi[r] = seed;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float d = bd[r];
      int i = bi[r];
#define UOC_MIN_STEP(CTRL)                                   \
  {                                                          \
    const float od = dpp_f<CTRL>(d);                         \
    const int oi = dpp_i<CTRL>(i);                           \
    if (od < d || (od == d && oi < i)) {                     \
      d = od;                                                \
      i = oi;                                                \
    }                                                        \
  }
      UOC_MIN_STEP(0xB1) UOC_MIN_STEP(0x4E) UOC_MIN_STEP(0x141) UOC_MIN_STEP(0x140)
#undef UOC_MIN_STEP
      bi[r] = i;
    }
    const int sel = (t == 0) ? bi[0] : (t == 1) ? bi[1] : (t == 2) ? bi[2] : bi[3];
    const int p = tile * 16 + 4 * q + t;
    if (t < 4 && p < n) {
      const int lab = slab[sel];
      labels[p] = lab;
      if (closest) closest[p] = sel;
      atomicAdd(&hist[lab], 1);
    }
  }
  __syncthreads();
  if (tid < NLAB && hist[tid]) atomicAdd(&counts[tid], hist[tid]);
}

// "assign zero to the largest cluster" (mean_shift.py:217-227): only labels in
// range(num_unique) are counted; first maximum wins; swap 0 <-> label_max.
__global__ __launch_bounds__(256) void relabel_swap_kernel(int *__restrict__ labels, int n,
                                                           const int *__restrict__ counts,
                                                           const int *__restrict__ num_unique) {
  const int b = blockIdx.y;
  labels += (size_t)b * n;
  counts += (size_t)b * NLAB;
  __shared__ int s_big;
  if (threadIdx.x == 0) {
    int num = num_unique[b];
    if (num > NLAB) num = NLAB;
    int big = 0, bc = INT_MIN;
    for (int i = 0; i < num; ++i)
      if (counts[i] > bc) {
        bc = counts[i];
        big = i;
      }
    s_big = big;
  }
  __syncthreads();
  const int big = s_big;
  if (big == 0) return;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    const int l = labels[p];
    if (l == 0)
      labels[p] = big;
    else if (l == big)
      labels[p] = 0;
  }
}

// ------------------------------- host side -------------------------------------------------
struct MsWorkspace {
  float *dmin;        // [batch][n]
  ArgMax *part[2];    // [batch][FPS_MAX_BLOCKS] ping-pong
  float *hc_partial;  // [batch][nblk][128][64]
  int *counts;        // [batch][128]
  int *num_unique;    // [batch]
  int *seed_labels;   // [batch][128]
  float *Z;           // [batch][128][64]
  int hc_nblk;
  int nh;  // 64-channel halves per embedding (1, or 2 for 128-d fields)
  size_t total;
};

// 64-d fields run the register-resident kernel (one wave per SIMD), 128-d fields the LDS-fragment kernel.
static int hc_variant() { return 2; }

// Virtual blocks of a hill-climbing launch: a function of the field size ONLY (about 16 pixel tiles per wave, at most 256
// blocks of 4 waves), so the fp32 summation order of the new seed positions is the same whether a field is clustered
// alone, with three other frames or among 30 crops.  (Round 2 derived the block count from the batch: a frame's label
// map could then depend on its launch-set mates.)
static int hc_virtual_blocks(int n) {
  const int ntile = (n + 15) / 16;
  const int tiles_per_wave = 16;   // part of the summation order (the last bits of the seeds): a constant
  int nvb = (ntile + 4 * tiles_per_wave - 1) / (4 * tiles_per_wave);
  if (nvb > 256) nvb = 256;
  if (nvb < 1) nvb = 1;
  return nvb;
}

// Physical blocks per field: the count p that minimises (rounds of CUs the grid needs) x (virtual blocks per physical
// block); one 4-wave block per CU (register-resident kernel) or two (LDS-fragment kernel).
static int hc_physical_blocks(int batch, int nvb, int per_cu) {
  const int slots = (device_num_cu() > 0 ? device_num_cu() : 256) * per_cu;
  int best = 1;
  long best_cost = -1;
  for (int p = 1; p <= nvb; ++p) {
    const long rounds = ((long)p * batch + slots - 1) / slots;
    const long cost = rounds * ((nvb + p - 1) / p);
    if (best_cost < 0 || cost < best_cost) {   // ties: fewer blocks, each walking more virtual blocks (one prologue, pipelined)
      best_cost = cost;
      best = p;
    }
  }
  return best;
}

// grid of the per-pixel kernels that have no cross-pixel sums (assign): any block count gives the same result
static int hc_blocks(int batch, int n, int nh = 1, bool lds_kernel = false) {
  (void)nh;
  (void)lds_kernel;
  const int ntile = (n + 15) / 16;
  int nblk = 512 / (batch > 0 ? batch : 1);
  if (nblk < 8) nblk = 8;
  const int maxb = (ntile + 3) / 4;
  if (nblk > maxb) nblk = maxb;
  if (nblk > HC_MAX_BLOCKS) nblk = HC_MAX_BLOCKS;
  if (nblk < 1) nblk = 1;
  return nblk;
}

static MsWorkspace carve(void *base, int batch, int n, int nh = 1) {
  MsWorkspace w;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    void *p = base ? (void *)((char *)base + off) : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  w.hc_nblk = hc_virtual_blocks(n);
  w.dmin = (float *)take((size_t)batch * n * sizeof(float));
  w.part[0] = (ArgMax *)take((size_t)batch * FPS_MAX_BLOCKS * sizeof(ArgMax));
  w.part[1] = (ArgMax *)take((size_t)batch * FPS_MAX_BLOCKS * sizeof(ArgMax));
  w.hc_partial = (float *)take((size_t)batch * w.hc_nblk * nh * NLAB * C * sizeof(float));
  w.counts = (int *)take((size_t)batch * NLAB * sizeof(int));
  w.num_unique = (int *)take((size_t)batch * sizeof(int));
  w.seed_labels = (int *)take((size_t)batch * NLAB * sizeof(int));
  w.Z = (float *)take((size_t)batch * nh * NLAB * C * sizeof(float));
  w.nh = nh;
  w.total = off;
  return w;
}

static int check_common(const void *X, int batch, int n, int m, void *ws, size_t ws_bytes, int nh = 1) {
  UOC_REQUIRE(X != nullptr, "X is null");
  UOC_REQUIRE(batch >= 1 && batch <= 65535, "batch=%d out of range [1,65535]", batch);
  UOC_REQUIRE(n >= 1, "n=%d must be >= 1", n);
  UOC_REQUIRE(m >= 1 && m <= UOC_MAX_SEEDS, "num_seeds=%d out of range [1,%d]", m, UOC_MAX_SEEDS);
  UOC_REQUIRE(((uintptr_t)X & 15) == 0, "X must be 16-byte aligned");
  const size_t need = carve(nullptr, batch, n, nh).total;
  UOC_REQUIRE(ws != nullptr && ws_bytes >= need, "workspace too small: %zu < %zu", ws_bytes, need);
  UOC_REQUIRE(((uintptr_t)ws & 255) == 0, "workspace must be 256-byte aligned");
  return UOC_OK;
}

static int fps_blocks(int n) {
  const int nchunk = (n + 63) / 64;
  int nblk = (nchunk + 3) / 4;
  if (nblk > FPS_MAX_BLOCKS) nblk = FPS_MAX_BLOCKS;
  if (nblk < 1) nblk = 1;
  return nblk;
}

static int g_fps_persistent = -1;  // -1: read UOC_FPS_PERSISTENT on first use
static std::atomic<int> g_fps_fallbacks{0};
static std::atomic<int> g_fps_stream_ordering{0};  // uoc_ms_set_stream_ordering: callers that launch from several streams
static int fps_persistent_plan(int batch, int n, int *bpi, int *nslots) {
  const int g_num_cu = device_num_cu();  // of the CURRENT device
  if (g_num_cu <= 0) return 0;
  if (g_fps_persistent < 0) {
    const char *e = getenv("UOC_FPS_PERSISTENT");
    g_fps_persistent = e ? atoi(e) : 1;
  }
  if (!g_fps_persistent || n >= (1 << 24)) return 0;
  int b = g_num_cu / batch;  // blocks per item: as many as stay co-resident
  if (b < 1) return 0;
  if (b > 256) b = 256;
  const int per_block = (n + b - 1) / b;
  int ns = (per_block + FPP_THREADS - 1) / FPP_THREADS;  // pixels per lane
  if (ns > FPP_SLOTS) return 0;  // does not fit on chip: split the batch / use the streaming kernel
  if (ns < 1) ns = 1;
  // Pack the field onto as few CUs as its pixels need (all FPP_SLOTS pixel slots of every lane: 480x640 on 150 instead of
  // 200 CUs).  The kernel is bound by the per-step grid exchange, not by its 64 FMAs per pixel, and the CUs it does not
  // occupy run other streams' kernels meanwhile: 150.0 -> 158.3 frames/s sustained (round 3).
  const int pack = FPP_SLOTS;
  if (pack > ns) ns = pack < FPP_SLOTS ? pack : FPP_SLOTS;
  b = (n + FPP_THREADS * ns - 1) / (FPP_THREADS * ns);  // drop blocks that would own no pixel
  *bpi = b;
  *nslots = ns;
  return 1;
}

static int run_select_seeds_streaming(const float *X, int batch, int n, int m, const int32_t *first, float *seeds,
                                      int32_t *indices, const MsWorkspace &w, hipStream_t st, int num_init = 0);

// One event per device that orders the persistent sampling kernels of all streams (see run_select_seeds).
struct FpsChain {
  std::mutex mu;
  hipEvent_t ev[kMaxDevices] = {};
  hipEvent_t event() {
    const int d = current_device();
    if (!ev[d] && hipEventCreateWithFlags(&ev[d], hipEventDisableTiming) != hipSuccess) ev[d] = nullptr;
    return ev[d];
  }
};
static FpsChain &fps_chain() {
  static FpsChain *c = new FpsChain();  // never destructed: static destruction order vs. the HIP runtime is undefined
  return *c;
}
// A stream that is being captured into a hipGraph (the one-frame-at-a-time replay path, fcn/graph_replay.py) gets a PLAIN
// launch of the persistent grid and no event chain: a cooperative launch and a wait on an event recorded outside the
// capture are not capturable.  Co-residency then rests on the plan (grid <= the device's resident capacity) and on the
// caller replaying such graphs on ONE stream per device, with nothing else running beside them.

static int run_select_seeds(const float *X, int batch, int n, int m, const int32_t *first, float *seeds,
                            int32_t *indices, const MsWorkspace &w, hipStream_t st) {
  // Persistent path: as many items per cooperative launch as stay co-resident; a larger batch
  // (stage 2 with > 8 ROIs) is split into several launches rather than dropped to the streaming kernel.
  int done = 0;
  while (m >= 2 && done < batch && w.nh == 1) {  // the on-chip kernel holds 64-d rows; 128-d fields stream
    // the largest number of fields that stays on chip, then the remaining fields spread evenly over the launches
    // that needs (21 crop fields: 7 + 7 + 7 instead of halving down to 6 + 6 + 6 + 3).  Fewer launches matter: never
    // using the LDS pixel slot (28 crop fields as 4 x 7 instead of 10 + 9 + 9) measured 123.5 vs 126.0 frames/s.
    int sub = batch - done, bpi = 0, nslots = 0;
    while (sub > 1 && !fps_persistent_plan(sub, n, &bpi, &nslots)) --sub;
    if (!fps_persistent_plan(sub, n, &bpi, &nslots)) break;
    const int launches = (batch - done + sub - 1) / sub;
    sub = (batch - done + launches - 1) / launches;
    if (!fps_persistent_plan(sub, n, &bpi, &nslots)) break;  // (a smaller batch always fits if a larger one does)
    unsigned long long *gran = reinterpret_cast<unsigned long long *>(w.part[0]);
    const size_t gbytes = (size_t)2 * sub * bpi * sizeof(unsigned long long);
    if (gbytes + sizeof(unsigned long long) > (size_t)batch * FPS_MAX_BLOCKS * sizeof(ArgMax)) break;
    int *status = reinterpret_cast<int *>(gran + (size_t)2 * sub * bpi);   // behind the granules: ONE fill for both
    UOC_HIP_CHECK(hipMemsetAsync(gran, 0, gbytes + sizeof(unsigned long long), st));  // tag 0 = "not published": re-initialised every call
    // the LDS pixel slot is only touched when a lane owns more than FPP_RS pixels; without it the kernel needs no
    // dynamic LDS at all and can share a CU with another stream's convolution blocks (two frames in flight)
    const size_t lds = (nslots > FPP_RS) ? (size_t)FPP_LS * (C / 4) * FPP_THREADS * sizeof(float4) : 0;
    static DeviceOnce attr_set;
    if (!attr_set.done()) {
      UOC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&fps_persistent_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)((size_t)FPP_LS * (C / 4) * FPP_THREADS * sizeof(float4))));
      attr_set.mark();
    }
    const float *Xc = X + (size_t)done * n * C;
    const int32_t *fc = first + done;
    float *sc = seeds + (size_t)done * m * C;
    int32_t *ic = indices + (size_t)done * m;
    void *args[] = {(void *)&Xc, (void *)&n, (void *)&m, (void *)&bpi, (void *)&nslots, (void *)&fc,
                    (void *)&sc, (void *)&ic, (void *)&gran, (void *)&status};
    hipError_t e;
    {
      // Persistent grids from different streams (two frames in flight) must not be partially resident at the same
      // time — each would spin for peers the other one keeps off the chip.  A per-device event chain runs them one
      // after the other; everything else on the two streams still overlaps.
      FpsChain &chain = fps_chain();
      std::lock_guard<std::mutex> lock(chain.mu);
      const bool capturing = stream_is_capturing(st);
      hipEvent_t ev = (g_fps_stream_ordering.load() && !capturing) ? chain.event() : nullptr;
      if (ev) UOC_HIP_CHECK(hipStreamWaitEvent(st, ev, 0));
      {
        ProfScope prof(KC_FPS_STEP, st, 2.0 * sub * (double)n * C * (m - 1), 4.0 * sub * (double)n * C,
                       ProfTag{{n, sub, bpi, nslots}});
        if (!capturing)
          e = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(&fps_persistent_kernel), dim3(sub * bpi),
                                         dim3(FPP_THREADS), args, (unsigned)lds, st);
        else
          e = hipLaunchKernel(reinterpret_cast<const void *>(&fps_persistent_kernel), dim3(sub * bpi), dim3(FPP_THREADS),
                              args, lds, st);
      }
      if (e == hipSuccess && ev) UOC_HIP_CHECK(hipEventRecord(ev, st));
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();  // not co-resident on this device: the rest goes to the streaming path
      break;
    }
    done += sub;
  }
  if (done >= batch) return UOC_OK;
  if (g_fps_persistent != 0 && w.nh == 1 && m >= 2) {
    // Observable fallback: the streaming kernel sums the 64-d dot product in another order, so a near-tied argmax may
    // pick another pixel than the on-chip kernel would have.  Count it and say so once.
    if (g_fps_fallbacks.fetch_add(1) == 0)
      fprintf(stderr, "[uoc] farthest-point sampling: %d of %d field(s) of n=%d ran on the streaming kernel (the on-chip "
              "kernel does not fit / is not co-resident on this device); see uoc_ms_fps_fallbacks()\n", batch - done, batch, n);
  }
  return run_select_seeds_streaming(X + (size_t)done * w.nh * n * C, batch - done, n, m, first + done,
                                    seeds + (size_t)done * w.nh * m * C, indices + (size_t)done * m, w, st);
}

static int run_select_seeds_streaming(const float *X, int batch, int n, int m, const int32_t *first, float *seeds,
                                      int32_t *indices, const MsWorkspace &w, hipStream_t st, int num_init) {
  const int nblk = fps_blocks(n);
  for (int s = 0; s < m; ++s) {
    dim3 grid(nblk, batch);  // gridDim.x doubles as the partial count, so it is the same every step
    const bool last = s == m - 1;
    ProfScope prof(KC_FPS_STEP, st, last ? 0.0 : 2.0 * batch * n * C * w.nh,
                   last ? 0.0 : 4.0 * batch * ((double)n * C * w.nh + 2.0 * n));
    if (w.nh == 2)
      hipLaunchKernelGGL(fps_step_kernel<2>, grid, dim3(FPS_THREADS), 0, st, X, n, m, s, num_init, first, w.dmin, seeds,
                         indices, w.part[(s + 1) & 1], w.part[s & 1]);
    else
      hipLaunchKernelGGL(fps_step_kernel<1>, grid, dim3(FPS_THREADS), 0, st, X, n, m, s, num_init, first, w.dmin, seeds,
                         indices, w.part[(s + 1) & 1], w.part[s & 1]);
  }
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

// ---- the flat item schedule of the register-resident hill-climbing kernel (HcPlan, hc_iter_flat_kernel) ----
// Relative cost of an item on one CU, in units of one 16-seed tile over the virtual block's pixel tiles (9.25 us at 16
// pixel tiles per wave), fitted to scripts/hc_parts.py on MI355X (profiles/r06_hc_parts.md).
constexpr double HC_COST_TILES = 6.1;      // a whole item: six 16-seed tiles + the 4-seed tile (a tenth of a full one)
constexpr double HC_COST_ITEM = 0.76;      // first item of a block, or any part (own fragments + first pixel tile: 7 us)
constexpr double HC_COST_NEXT = 0.3;       // every further whole item of a block (pipelined)
static double hc_part_cost(int P, int p) {   // part p of a P-way split (mirrors the kernel's table: largest first)
  const double tiles = P == 2 ? (p == 0 ? 3.1 : 3.0) : P == 3 ? (p == 0 ? 2.1 : 2.0) : (p == 0 ? 1.1 : 1.0);
  return tiles + HC_COST_ITEM;
}
// Makespan (relative) of a plan under the kernel's static schedule: contiguous chunks of whole items, then the tail
// parts largest first, dealt forwards and backwards.
static double hc_plan_cost(const HcPlan &p) {
  double phase1 = 0.0;
  if (p.n1 > 0) {
    const int c = (p.n1 + p.grid - 1) / p.grid;
    phase1 = c * HC_COST_TILES + HC_COST_ITEM + (c - 1) * HC_COST_NEXT;
  }
  const int R = p.items - p.n1, T = R * p.parts;
  double tail = 0.0;
  for (int g = 0; g < p.grid && T > 0; ++g) {
    double c = 0.0;
    for (int k = 0; k < 2; ++k) {
      const int t = k * p.grid + ((k & 1) ? p.grid - 1 - g : g);
      if (t < T) c += hc_part_cost(p.parts, t / R);
    }
    if (c > tail) tail = c;
  }
  return phase1 + tail;
}
static EnvInt g_hc_parts("UOC_HC_PARTS", 0);   // speed-only (bit-identical): 0 = by makespan, 1 / 2 / 3 / 6 = force
static HcPlan hc_make_plan(int batch, int nvb, bool splittable) {
  const int cus = device_num_cu() > 0 ? device_num_cu() : 256;
  HcPlan whole;
  whole.nvb = nvb;
  whole.items = batch * nvb;
  whole.n1 = whole.items;
  whole.parts = 1;
  whole.grid = whole.items < cus ? whole.items : cus;
  if (!splittable) return whole;
  const int force = g_hc_parts.get();
  if (force == 1) return whole;
  HcPlan best = whole;
  double best_cost = hc_plan_cost(whole);
  static const int cand[3] = {2, 3, 6};
  for (int ci = 0; ci < 3; ++ci) {
    const int P = cand[ci];
    if (force > 1 && force != P) continue;
    const int q = whole.items / cus;
    for (int back = 0; back <= 1; ++back) {   // the tail alone, or the tail + the last full round
      if (q - back < 0) break;
      HcPlan c = whole;
      c.parts = P;
      c.n1 = (q - back) * cus;
      const int T = (c.items - c.n1) * P;
      if (T == 0) continue;
      c.grid = c.n1 > 0 ? cus : (T < cus ? T : cus);
      if (T > 2 * c.grid) continue;
      const double cost = hc_plan_cost(c);
      if (force > 1 ? (best.parts != P || cost < best_cost) : cost < 0.985 * best_cost) {
        best = c;
        best_cost = cost;
      }
    }
  }
  return best;
}

template <int ST, int NH>
static void launch_hc(const float *X, int batch, int n, float *Z, int m, float kappa, int iters,
                      const MsWorkspace &w, hipStream_t st) {
  const size_t zbytes = (size_t)NH * ST * 16 * ZP * sizeof(float);
  const size_t rbytes = (size_t)2 * ST * 4 * 64 * sizeof(f32x4);
  const size_t lds = zbytes > rbytes ? zbytes : rbytes;
  static DeviceOnce attr_set;
  if (!attr_set.done() && lds > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&hc_iter_kernel<ST, NH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set.mark();
  }
  const bool reg = NH == 1 && hc_variant() == 2;   // register-resident kernel, one wave per SIMD
  const size_t lds_reg = (size_t)4 * ST * 4 * 64 * sizeof(f32x4);
  const int nvb = w.hc_nblk;
  const int last = m - 16 * (ST - 1);                  // seeds in the last tile
  const bool quad = ST >= 2 && last >= 1 && last <= 4;   // they run on the 4x4x1 MFMA instead of a padded 16-seed tile
  HcPlan plan = {};
  if constexpr (NH == 1) {
    if (reg) {
      plan = hc_make_plan(batch, nvb, ST == 7 && quad);
      static DeviceOnce attr_reg;
      if (!attr_reg.done() && lds_reg > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&hc_iter_flat_kernel<ST, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_reg);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&hc_iter_flat_kernel<ST, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_reg);
        attr_reg.mark();
      }
    }
  }
  const int phys = reg ? plan.grid : hc_physical_blocks(batch, nvb, 2);
  for (int it = 0; it < iters; ++it) {
    {
      // NH = 2 recomputes S for each half of the accumulators: (2 + 1) / 2 of the algorithmic flops per half
      ProfScope prof(KC_HC_ITER, st, 4.0 * batch * m * (double)n * C * NH,
                     4.0 * batch * ((double)n * C * NH + 2.0 * m * C * NH), ProfTag{{n, batch, nvb, phys}});
      if constexpr (NH == 1) {
        if (reg) {
          if (quad)
            hipLaunchKernelGGL((hc_iter_flat_kernel<ST, true>), dim3(plan.grid), dim3(256), lds_reg, st, X, n, Z, m, kappa,
                               w.hc_partial, plan);
          else
            hipLaunchKernelGGL((hc_iter_flat_kernel<ST, false>), dim3(plan.grid), dim3(256), lds_reg, st, X, n, Z, m, kappa,
                               w.hc_partial, plan);
        }
      }
      if (!reg)
        hipLaunchKernelGGL((hc_iter_kernel<ST, NH>), dim3(phys, batch, NH), dim3(HC_THREADS), lds, st, X, n, Z, m,
                           kappa, w.hc_partial, nvb);
    }
    ProfScope prof(KC_HC_FINALIZE, st, 0.0, 4.0 * batch * w.hc_nblk * NH * ST * 16.0 * C);
    hipLaunchKernelGGL(hc_finalize_kernel<NH>, dim3(m, batch), dim3(256), 0, st, w.hc_partial, w.hc_nblk, ST * 16, m, Z);
  }
}

template <int NH>
static void run_hill_climb_nh(const float *X, int batch, int n, float *Z, int m, float kappa, int iters,
                              const MsWorkspace &w, hipStream_t st) {
  switch ((m + 15) / 16) {
    case 1: launch_hc<1, NH>(X, batch, n, Z, m, kappa, iters, w, st); break;
    case 2: launch_hc<2, NH>(X, batch, n, Z, m, kappa, iters, w, st); break;
    case 3: launch_hc<3, NH>(X, batch, n, Z, m, kappa, iters, w, st); break;
    case 4: launch_hc<4, NH>(X, batch, n, Z, m, kappa, iters, w, st); break;
    case 5: launch_hc<5, NH>(X, batch, n, Z, m, kappa, iters, w, st); break;
    case 6: launch_hc<6, NH>(X, batch, n, Z, m, kappa, iters, w, st); break;
    case 7: launch_hc<7, NH>(X, batch, n, Z, m, kappa, iters, w, st); break;
    default: launch_hc<8, NH>(X, batch, n, Z, m, kappa, iters, w, st); break;
  }
}

static int run_hill_climb(const float *X, int batch, int n, float *Z, int m, float kappa, int iters,
                          const MsWorkspace &w, hipStream_t st) {
  if (w.nh == 2)
    run_hill_climb_nh<2>(X, batch, n, Z, m, kappa, iters, w, st);
  else
    run_hill_climb_nh<1>(X, batch, n, Z, m, kappa, iters, w, st);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

template <int ST, int NH>
static void launch_assign(const float *X, int batch, int n, const float *Z, const int *seed_labels, int m,
                          int *labels, int *closest, const MsWorkspace &w, hipStream_t st) {
  int nblk = hc_blocks(batch, n, 1, true) * 2;
  const int maxb = ((n + 15) / 16 + 3) / 4;
  if (nblk > maxb) nblk = maxb;
  const size_t lds = (size_t)NH * ST * 16 * ZP * sizeof(float);
  static DeviceOnce attr_set;
  if (!attr_set.done() && lds > 48 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&assign_kernel<ST, NH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set.mark();
  }
  ProfScope prof(KC_ASSIGN, st, 2.0 * batch * m * (double)n * C * NH, 4.0 * batch * ((double)n * C * NH + n));
  hipLaunchKernelGGL((assign_kernel<ST, NH>), dim3(nblk, batch), dim3(HC_THREADS), lds, st, X, n, Z, seed_labels, m,
                     labels, closest, w.counts);
}

template <int NH>
static void run_assign_nh(const float *X, int batch, int n, const float *Z, const int *seed_labels, int m, int *labels,
                          int *closest, const MsWorkspace &w, hipStream_t st) {
  switch ((m + 15) / 16) {
    case 1: launch_assign<1, NH>(X, batch, n, Z, seed_labels, m, labels, closest, w, st); break;
    case 2: launch_assign<2, NH>(X, batch, n, Z, seed_labels, m, labels, closest, w, st); break;
    case 3: launch_assign<3, NH>(X, batch, n, Z, seed_labels, m, labels, closest, w, st); break;
    case 4: launch_assign<4, NH>(X, batch, n, Z, seed_labels, m, labels, closest, w, st); break;
    case 5: launch_assign<5, NH>(X, batch, n, Z, seed_labels, m, labels, closest, w, st); break;
    case 6: launch_assign<6, NH>(X, batch, n, Z, seed_labels, m, labels, closest, w, st); break;
    case 7: launch_assign<7, NH>(X, batch, n, Z, seed_labels, m, labels, closest, w, st); break;
    default: launch_assign<8, NH>(X, batch, n, Z, seed_labels, m, labels, closest, w, st); break;
  }
}

static int run_assign(const float *X, int batch, int n, const float *Z, const int *seed_labels,
                      const int *num_unique, int m, int *labels, int *closest, const MsWorkspace &w,
                      hipStream_t st) {
  UOC_HIP_CHECK(hipMemsetAsync(w.counts, 0, (size_t)batch * NLAB * sizeof(int), st));
  if (w.nh == 2)
    run_assign_nh<2>(X, batch, n, Z, seed_labels, m, labels, closest, w, st);
  else
    run_assign_nh<1>(X, batch, n, Z, seed_labels, m, labels, closest, w, st);
  int rb = (n + 255) / 256;
  if (rb > 512) rb = 512;
  ProfScope prof(KC_RELABEL, st, 0.0, 8.0 * batch * n);
  hipLaunchKernelGGL(relabel_swap_kernel, dim3(rb, batch), dim3(256), 0, st, labels, n, w.counts, num_unique);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

static int run_seed_cc(const float *Z, int batch, int m, float eps, int *seed_labels, int *num_unique, int nh,
                       hipStream_t st) {
  ProfScope prof(KC_SEED_CC, st, 0.0, 4.0 * batch * m * C * nh);
  const size_t lds = (size_t)NLAB * (nh * C + 1) * sizeof(float);
  if (nh == 2) {
    static DeviceOnce attr_set;
    if (!attr_set.done()) {
      UOC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&seed_cc_kernel<2>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_set.mark();
    }
    hipLaunchKernelGGL(seed_cc_kernel<2>, dim3(batch), dim3(64), lds, st, Z, m, eps, seed_labels, num_unique);
  } else {
    hipLaunchKernelGGL(seed_cc_kernel<1>, dim3(batch), dim3(64), lds, st, Z, m, eps, seed_labels, num_unique);
  }
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

}  // namespace uoc

using namespace uoc;

extern "C" {

int uoc_ms_set_persistent_fps(int on) {
  g_fps_persistent = on ? 1 : 0;
  return UOC_OK;
}

int uoc_ms_fps_fallbacks(void) { return g_fps_fallbacks.load(); }

int uoc_ms_set_stream_ordering(int on) {
  g_fps_stream_ordering.store(on ? 1 : 0);
  return UOC_OK;
}

int uoc_shutdown(void) {
  // releases the per-device ordering events while the HIP runtime is still alive (a live event at process exit
  // crashes the rocprofv3 tool's finaliser); the host mirrors call it from a Python atexit hook
  FpsChain &chain = fps_chain();
  std::lock_guard<std::mutex> lock(chain.mu);
  for (int d = 0; d < kMaxDevices; ++d)
    if (chain.ev[d]) {
      (void)hipEventDestroy(chain.ev[d]);
      chain.ev[d] = nullptr;
    }
  return UOC_OK;
}

int uoc_ms_check(void *stream) {
  hipStream_t st = (hipStream_t)stream;
  int flag = 0;
  UOC_HIP_CHECK(hipMemcpyFromSymbolAsync(&flag, HIP_SYMBOL(g_fps_timeout), sizeof(int), 0, hipMemcpyDeviceToHost, st));
  UOC_HIP_CHECK(hipStreamSynchronize(st));
  if (!flag) return UOC_OK;
  const int zero = 0;
  UOC_HIP_CHECK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_fps_timeout), &zero, sizeof(int), 0, hipMemcpyHostToDevice, st));
  UOC_HIP_CHECK(hipStreamSynchronize(st));
  set_error("farthest-point sampling: a block's grid-wide exchange timed out (blocks not co-resident?); results of "
            "the affected clustering call are invalid");
  return UOC_ETIMEDOUT;
}

size_t uoc_ms_workspace_bytes(int batch, int n, int m) {
  (void)m;
  if (batch < 1 || n < 1) return 0;
  return carve(nullptr, batch, n).total;
}

int uoc_ms_select_seeds(const float *d_X, int batch, int n, int m, const int32_t *d_first_index, float *d_seeds,
                        int32_t *d_indices, void *d_ws, size_t ws_bytes, void *stream) {
  if (int rc = check_common(d_X, batch, n, m, d_ws, ws_bytes)) return rc;
  UOC_REQUIRE(d_first_index && d_seeds && d_indices, "null output/first_index pointer");
  return run_select_seeds(d_X, batch, n, m, d_first_index, d_seeds, d_indices, carve(d_ws, batch, n),
                          (hipStream_t)stream);
}

int uoc_ms_select_seeds_from(const float *d_X, int batch, int n, int m, int num_init, const int32_t *d_first_index,
                             float *d_seeds, int32_t *d_indices, void *d_ws, size_t ws_bytes, void *stream) {
  if (int rc = check_common(d_X, batch, n, m, d_ws, ws_bytes)) return rc;
  UOC_REQUIRE(d_seeds && d_indices && ((uintptr_t)d_seeds & 15) == 0, "seeds / indices null or seeds not 16-byte aligned");
  UOC_REQUIRE(num_init >= 0 && num_init <= m, "num_init=%d out of range [0, %d]", num_init, m);
  UOC_REQUIRE(num_init > 0 || d_first_index, "null first_index pointer");
  if (num_init == 0)
    return run_select_seeds(d_X, batch, n, m, d_first_index, d_seeds, d_indices, carve(d_ws, batch, n), (hipStream_t)stream);
  return run_select_seeds_streaming(d_X, batch, n, m, d_first_index, d_seeds, d_indices, carve(d_ws, batch, n),
                                    (hipStream_t)stream, num_init);
}

int uoc_ms_hill_climb(const float *d_X, int batch, int n, float *d_Z, int m, float kappa, int iters, void *d_ws,
                      size_t ws_bytes, void *stream) {
  if (int rc = check_common(d_X, batch, n, m, d_ws, ws_bytes)) return rc;
  UOC_REQUIRE(d_Z != nullptr && ((uintptr_t)d_Z & 15) == 0, "Z null or not 16-byte aligned");
  UOC_REQUIRE(iters >= 0, "iters=%d must be >= 0", iters);
  return run_hill_climb(d_X, batch, n, d_Z, m, kappa, iters, carve(d_ws, batch, n), (hipStream_t)stream);
}

int uoc_ms_seed_components(const float *d_Z, int batch, int m, float epsilon, int32_t *d_seed_labels,
                           int32_t *d_num_unique, void *stream) {
  UOC_REQUIRE(d_Z && d_seed_labels && d_num_unique, "null pointer");
  UOC_REQUIRE(batch >= 1 && m >= 1 && m <= UOC_MAX_SEEDS, "batch=%d m=%d out of range", batch, m);
  return run_seed_cc(d_Z, batch, m, epsilon, d_seed_labels, d_num_unique, 1, (hipStream_t)stream);
}

int uoc_ms_assign(const float *d_X, int batch, int n, const float *d_Z, const int32_t *d_seed_labels,
                  const int32_t *d_num_unique, int m, int32_t *d_labels, int32_t *d_closest, void *d_ws,
                  size_t ws_bytes, void *stream) {
  if (int rc = check_common(d_X, batch, n, m, d_ws, ws_bytes)) return rc;
  UOC_REQUIRE(d_Z && d_seed_labels && d_num_unique && d_labels, "null pointer");
  return run_assign(d_X, batch, n, d_Z, d_seed_labels, d_num_unique, m, d_labels, d_closest, carve(d_ws, batch, n),
                    (hipStream_t)stream);
}

int uoc_ms_cluster(const float *d_X, int batch, int n, int m, float kappa, int iters, float epsilon,
                   const int32_t *d_first_index, int32_t *d_labels, int32_t *d_indices, float *d_Z_out,
                   int32_t *d_seed_labels_out, void *d_ws, size_t ws_bytes, void *stream) {
  if (int rc = check_common(d_X, batch, n, m, d_ws, ws_bytes)) return rc;
  UOC_REQUIRE(d_first_index && d_labels && d_indices, "null pointer");
  UOC_REQUIRE(iters >= 0, "iters=%d must be >= 0", iters);
  hipStream_t st = (hipStream_t)stream;
  MsWorkspace w = carve(d_ws, batch, n);
  float *Z = d_Z_out ? d_Z_out : w.Z;
  int *sl = d_seed_labels_out ? d_seed_labels_out : w.seed_labels;
  if (int rc = run_select_seeds(d_X, batch, n, m, d_first_index, Z, d_indices, w, st)) return rc;
  if (int rc = run_hill_climb(d_X, batch, n, Z, m, kappa, iters, w, st)) return rc;
  if (int rc = run_seed_cc(Z, batch, m, epsilon, sl, w.num_unique, 1, st)) return rc;
  return run_assign(d_X, batch, n, Z, sl, w.num_unique, m, d_labels, nullptr, w, st);
}

size_t uoc_ms_workspace_bytes_wide(int batch, int n, int m, int halves) {
  (void)m;
  if (batch < 1 || n < 1 || halves < 1 || halves > 2) return 0;
  return carve(nullptr, batch, n, halves).total;
}

int uoc_ms_cluster_wide(const float *d_X, int halves, int batch, int n, int m, float kappa, int iters, float epsilon,
                        const int32_t *d_first_index, int32_t *d_labels, int32_t *d_indices, float *d_Z_out,
                        int32_t *d_seed_labels_out, void *d_ws, size_t ws_bytes, void *stream) {
  UOC_REQUIRE(halves == 1 || halves == 2, "halves=%d (64-d or 128-d embeddings only)", halves);
  if (int rc = check_common(d_X, batch, n, m, d_ws, ws_bytes, halves)) return rc;
  UOC_REQUIRE(d_first_index && d_labels && d_indices, "null pointer");
  UOC_REQUIRE(iters >= 0, "iters=%d must be >= 0", iters);
  hipStream_t st = (hipStream_t)stream;
  MsWorkspace w = carve(d_ws, batch, n, halves);
  float *Z = d_Z_out ? d_Z_out : w.Z;
  int *sl = d_seed_labels_out ? d_seed_labels_out : w.seed_labels;
  if (int rc = run_select_seeds(d_X, batch, n, m, d_first_index, Z, d_indices, w, st)) return rc;
  if (int rc = run_hill_climb(d_X, batch, n, Z, m, kappa, iters, w, st)) return rc;
  if (int rc = run_seed_cc(Z, batch, m, epsilon, sl, w.num_unique, halves, st)) return rc;
  return run_assign(d_X, batch, n, Z, sl, w.num_unique, m, d_labels, nullptr, w, st);
}

}  // extern "C"
