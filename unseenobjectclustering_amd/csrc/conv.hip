// fp32 implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_16x16x4_f32) with the
// BatchNorm shift, residual add and ReLU fused into the epilogue, plus the small layout /
// pooling kernels of the ResNet34-8s backbone.
//
// Replaces the cuDNN/ATen calls behind /root/reference/lib/networks/resnet.py:
//   conv3x3 (+dilation, stride)  :24-41,57-73   conv1x1 downsample :215-219
//   stem conv7x7 s2 p3           :141-144       MaxPool2d(3,2,1)   :145
//   fc = Conv2d(512, 64, 1)      resnet_dilated.py:303
//
// GEMM view: D[cout][pixel] = sum_k W[cout][k] * A[pixel][k], k = (tap, cin).  NHWC activations
// make every K-chunk (one tap, 32 input channels) a contiguous 128-byte row per pixel, and the
// weights are stored [tap][cout][cin] so the weight rows have the same shape.  Both operands
// are staged through LDS (double-buffered, register-staged so the row pitch can be padded) and
// read back as ds_read_b128 MFMA fragments; the K index inside a chunk is permuted
// (lane q supplies k = 16h + 4q + e) so one b128 read feeds four MFMA steps.
// With weights as the MFMA "A" operand, each lane ends up holding 4 CONSECUTIVE output channels
// of one pixel, so the epilogue is float4 loads/stores in NHWC.
#include "conv.h"

#include <mutex>
#include "dma.h"
#include "prof.h"

#include <stdlib.h>

namespace uoc {

constexpr int BK = 32;   // K-chunk (floats)
constexpr int BKP = 40;  // padded LDS row pitch: 10 x 16 B => the b128 fragment reads are bank-conflict free

__device__ __forceinline__ f32x4 mfma4c(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}


// -------------------------------------------------------------------------------------------
// Production variant: operands go HBM/L2 -> LDS directly (global_load_lds_dwordx4, no VGPR staging,
// no ds_write pass) into a 3-stage ring, two K-chunks ahead of the MFMAs, with counted vmcnt waits
// and one raw s_barrier per chunk.  The LDS image of a DMA is lane-linear (8 lanes x 16 B = one
// 128-byte row, no padding possible), so bank conflicts are removed by an XOR swizzle applied on
// the SOURCE address and again on the fragment read: physical 16-B slot s of row r holds logical
// slot s ^ ((r >> 1) & 7).  Out-of-image taps read a 128-byte zero page instead of being predicated.
// Measured on the layer4 shape (ablation): register staging cost 14 % (global-load
// wait) + 8 % (ds_write + barrier); this variant removes both.
// -------------------------------------------------------------------------------------------
__device__ float4 g_zero_page[8];  // zero-initialised device memory

// LDS-DMA primitives (glds16 / blds16 / make_srd / wait_vmcnt): csrc/dma.h

template <int BM, int BN, int WAVES_M, int WAVES_N, bool STEM, int VARIANT = 0>
__global__ __launch_bounds__(WAVES_M *WAVES_N * 64) void conv_glds_kernel(ConvParams p, int ntiles, int mtiles) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int R = BM + BN;   // rows per stage (activation rows, then weight rows), 32 floats each
  constexpr int RPP = NT / 8;  // rows per DMA pass: every wave moves 8 rows (1 KiB) per instruction
  constexpr int NPASS = (R + RPP - 1) / RPP;
  constexpr int STAGE = R * BK;  // floats per stage
  static_assert(WM % 16 == 0 && WN % 16 == 0 && R % 8 == 0, "tile shape");

  extern __shared__ __attribute__((aligned(16))) float smem[];  // [3][R][32]

  const int total = p.G * ntiles * mtiles;
  const int per_xcd = (total + 7) >> 3;
  const int work = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (work >= total) return;
  const int g = work / (ntiles * mtiles);
  const int rem = work - g * (ntiles * mtiles);
  const int nt = rem / mtiles, mt = rem - nt * mtiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int HoWo = p.Ho * p.Wo;
  const int M = p.B * HoWo;
  const int Kc = STEM ? 32 : p.Cin;
  const int T = STEM ? p.KH : p.KH * p.KW;
  const int cpt = STEM ? 1 : p.Cin / BK;
  const int nk = T * cpt;

  const float *__restrict__ in = p.in + (size_t)g * p.B * p.H * p.W * p.Cin;
  const float *__restrict__ w = p.w + (size_t)g * T * p.Cout * Kc;
  const float *__restrict__ bias = p.bias ? p.bias + (size_t)g * p.Cout : nullptr;
  const float *__restrict__ res = p.res ? p.res + (size_t)g * M * p.Cout : nullptr;
  float *__restrict__ out = p.out + (size_t)g * M * p.Cout;
  const float *zero = reinterpret_cast<const float *>(g_zero_page);
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) char *)smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int t = lane & 15, q = lane >> 4;

  // ---- per-pass DMA descriptors (row r of the stage, physical slot s, logical slot c) -----------
  int d_r0[NPASS];   // first row of this wave's 8-row group (wave-uniform)
  int d_off[NPASS];  // A row: image base (+4c unless STEM) ; W row: (n0+n)*Kc + 4c
  int d_iy0[NPASS], d_ix0[NPASS];
  bool d_w[NPASS];
#pragma unroll
  for (int j = 0; j < NPASS; ++j) {
    int r0 = j * RPP + wave * 8;
    if (r0 >= R) r0 = R - 8;  // surplus wave: re-copy the last group (identical bytes) so every wave issues NPASS DMAs
    d_r0[j] = r0;
    const int r = r0 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    d_w[j] = r >= BM;
    d_iy0[j] = -(1 << 28);
    d_ix0[j] = 0;
    if (r >= BM) {
      d_off[j] = (n0 + (r - BM)) * Kc + 4 * c;
    } else {
      const int m = m0 + r;
      d_off[j] = 0;
      if (m < M) {
        const int b = m / HoWo;
        const int rr = m - b * HoWo;
        const int oy = rr / p.Wo, ox = rr - oy * p.Wo;
        d_iy0[j] = oy * p.stride - p.pad;
        d_ix0[j] = ox * p.stride - p.pad + (STEM ? c : 0);
        d_off[j] = b * p.H * p.W * p.Cin + (STEM ? 0 : 4 * c);
      }
    }
  }

  // ---- non-stem layers: buffer-descriptor DMA (blds16).  Passes are split by KIND at compile time: NPA passes over the
  // BM activation rows, then NPW passes over the BN weight rows (same total as NPASS for every tile family); a pass
  // whose rows would start beyond its kind's range re-copies the kind's last 8 rows (identical bytes). ----
  constexpr int NPA = (BM + RPP - 1) / RPP, NPW = (BN + RPP - 1) / RPP;
  static_assert(STEM || NPA + NPW == NPASS, "pass split");
  const unsigned bias_bytes = (unsigned)((p.pad * p.W + p.pad) * p.Cin) * 4u;   // most negative pixel offset
  const v4i srd_a = make_srd(reinterpret_cast<const char *>(in) - bias_bytes,
                             (unsigned)((size_t)p.B * p.H * p.W * p.Cin * 4) + bias_bytes);
  const v4i srd_w = make_srd(w, (unsigned)((size_t)T * p.Cout * Kc * 4));
  int l_r0[NPASS];          // first row of the wave's 8-row group in the stage
  unsigned l_voff[NPASS];   // per-lane byte offset (activations: bias + image base + (iy0*W + ix0)*Cin + 4c)
  unsigned l_mask[NPASS];   // activations: bit `tap` set if that tap lies inside the image for the lane's pixel
  if (!STEM) {
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
      const bool isw = j >= NPA;
      int r0 = (isw ? BM + (j - NPA) * RPP : j * RPP) + wave * 8;
      if (!isw && r0 >= BM) r0 = BM - 8;
      if (isw && r0 >= R) r0 = R - 8;
      l_r0[j] = r0;
      const int r = r0 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      l_mask[j] = 0;
      l_voff[j] = kOobVoff;
      if (isw) {
        l_voff[j] = (unsigned)(((n0 + (r - BM)) * Kc + 4 * c) * 4);
      } else {
        const int m = m0 + r;
        if (m < M) {
          const int b = m / HoWo;
          const int rr = m - b * HoWo;
          const int oy = rr / p.Wo, ox = rr - oy * p.Wo;
          const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
          l_voff[j] = bias_bytes + (unsigned)((b * p.H * p.W * p.Cin + (iy0 * p.W + ix0) * p.Cin + 4 * c) * 4);
          unsigned mk = 0, bit = 1;
          for (int kh = 0, iy = iy0; kh < p.KH; ++kh, iy += p.dil) {
            const bool oky = (unsigned)iy < (unsigned)p.H;
            for (int kw = 0, ix = ix0; kw < p.KW; ++kw, ix += p.dil, bit <<= 1)
              if (oky && (unsigned)ix < (unsigned)p.W) mk |= bit;
          }
          l_mask[j] = mk;
        }
      }
    }
  }
  // one chunk of the lean path: two scalars (activation / weight offset of the chunk), per activation pass a bit test
  // and a select, per DMA the LDS destination
#define UOC_ISSUE_LEAN(STG)                                                                                    \
  {                                                                                                            \
    const unsigned soff_a_ = (unsigned)__builtin_amdgcn_readfirstlane(                                         \
        ((q_kh * p.dil * p.W + q_kw * p.dil) * p.Cin + q_c0) * 4);                                             \
    const unsigned soff_w_ = (unsigned)__builtin_amdgcn_readfirstlane((q_tap * p.Cout * Kc + q_c0) * 4);       \
    _Pragma("unroll") for (int j = 0; j < NPASS; ++j) {                                                        \
      const unsigned dst_ = lds_base + (unsigned)(((STG)*STAGE + l_r0[j] * BK) * sizeof(float));               \
      if (j < NPA) {                                                                                           \
        const unsigned v_ = ((l_mask[j] >> q_tap) & 1u) ? l_voff[j] : kOobVoff;                                \
        blds16(srd_a, v_, soff_a_, dst_);                                                                      \
      } else {                                                                                                 \
        blds16(srd_w, l_voff[j], soff_w_, dst_);                                                               \
      }                                                                                                        \
    }                                                                                                          \
  }

#define UOC_ISSUE_ONE(KN, STG, J)                                                                              \
  if ((J) < NPASS) {                                                                                           \
    const int j = (J) < NPASS ? (J) : 0;                                                                       \
    /* K order: cin slice outer, tap inner (L2 reuse of the slice); (tap, kh, kw, c0) of chunk KN are the    \
       running scalars q_tap.. advanced by UOC_ADVANCE (no integer divisions in the loop) */                   \
    const int tap = q_tap, c0 = q_c0, kh = STEM ? q_tap : q_kh, kw = STEM ? 0 : q_kw;                          \
    const size_t woff = (size_t)tap * p.Cout * Kc + c0;                                                        \
    const int iy = d_iy0[j] + kh * p.dil;                                                                      \
    const int ix = d_ix0[j] + kw * p.dil;                                                                      \
    const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;                              \
    /* branch-free source select: weight row | in-image activation row | zero page */                          \
    const unsigned long long aw = (unsigned long long)(w + woff + d_off[j]);                                   \
    const unsigned long long aa =                                                                              \
        (unsigned long long)(in + d_off[j] + (iy * p.W + ix) * p.Cin + (STEM ? 0 : c0));                       \
    const unsigned long long src = d_w[j] ? aw : (ok ? aa : (unsigned long long)zero);                         \
    glds16(reinterpret_cast<const float *>(src),                                                               \
           lds_base + (unsigned)(((STG)*STAGE + d_r0[j] * BK) * sizeof(float)));                               \
  }
#define UOC_ISSUE(KN, STG)                                                                                     \
  {                                                                                                            \
    if (!STEM) UOC_ISSUE_LEAN(STG) else {                                                                      \
      UOC_ISSUE_ONE(KN, STG, 0) UOC_ISSUE_ONE(KN, STG, 1) UOC_ISSUE_ONE(KN, STG, 2) UOC_ISSUE_ONE(KN, STG, 3)  \
      UOC_ISSUE_ONE(KN, STG, 4) UOC_ISSUE_ONE(KN, STG, 5) UOC_ISSUE_ONE(KN, STG, 6) UOC_ISSUE_ONE(KN, STG, 7)  \
    }                                                                                                          \
  }
#define UOC_FRAG2(STG, HH, WA, XB)                                                                           \
  {                                                                                                          \
    const float *base_ = smem + (STG)*STAGE;                                                                 \
    const int slot_ = ((4 * (HH) + q) ^ ((t >> 1) & 7)) * 4;                                                 \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) WA[j] =                                                   \
        *reinterpret_cast<const float4 *>(base_ + (BM + wn * WN + 16 * j + t) * BK + slot_);                 \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) XB[i] =                                                   \
        *reinterpret_cast<const float4 *>(base_ + (wm * WM + 16 * i + t) * BK + slot_);                      \
  }
#define UOC_MFMA_E(WA, XB, E)                                                                                  \
  {                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) _Pragma("unroll") for (int i = 0; i < TM; ++i) acc[j][i] =  \
        mfma4c(WA[j].E, XB[i].E, acc[j][i]);                                                                   \
  }

  f32x4 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // running decomposition of the chunk being ISSUED: tap = kh*KW + kw within the cin slice c0
  int q_tap = 0, q_kh = 0, q_kw = 0, q_c0 = 0;
#define UOC_ADVANCE()              \
  {                                \
    ++q_tap;                       \
    if (++q_kw == p.KW) {          \
      q_kw = 0;                    \
      ++q_kh;                      \
    }                              \
    if (q_tap == T) {              \
      q_tap = q_kh = q_kw = 0;     \
      q_c0 += BK;                  \
    }                              \
  }
  float4 wa0[TN], xb0[TM], wa1[TN], xb1[TM];
  UOC_ISSUE(0, 0)
  UOC_ADVANCE()
  if (nk > 1) {
    UOC_ISSUE(1, 1)
    UOC_ADVANCE()
    wait_vmcnt<NPASS>();
  } else {
    wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();
  UOC_FRAG2(0, 0, wa0, xb0)
  if (VARIANT >= 3) UOC_FRAG2(0, 1, wa1, xb1)
  int s_cur = 0, s_nxt = 1, s_nn = 2;  // ring positions of chunks kc, kc+1, kc+2
  const bool early = wave < (WAVES_M * WAVES_N) / 2;
  static_assert(NPASS <= 8, "DMA passes per chunk");
  for (int kc = 0; kc < nk; ++kc) {
    // (Spreading the DMAs between the MFMA groups was measured 5-15 % SLOWER than this burst.)
    // VARIANT != 0: timing ablations only (wrong results): 1 = no DMA in the loop, 2 = also no barrier /
    // waits, 3 = also no fragment reads.
    // The DMA burst (address VALU + 5 LDS-DMAs) is issued BEFORE the h=0 MFMAs by the first half of
    // the waves and AFTER them by the second half (waves w and w+NW/2 share a SIMD), so the two waves
    // of a SIMD never do their address arithmetic at the same time.  Same vmcnt accounting either way.
    // Ablation (layer4 shape): removing the DMAs altogether is worth +12 %, the barrier +3 %, the
    // fragment reads +4 %; the stagger itself measured neutral, i.e. the DMA cost is not issue-slot
    // contention (suspected: L2->LDS traffic lowers the sustained clock under the power cap).
    if (kc + 2 < nk && VARIANT < 1 && early) UOC_ISSUE(kc + 2, s_nn)
    // Fragment-read placement (checked in the ISA): hipcc sinks reads to just before their first use and
    // its waitcnt pass is conservative across the loop back-edge (lgkmcnt(0) in front of the first MFMA
    // of the iteration).  So: the first MFMA group runs on fragments whose reads were drained at the END
    // of the previous iteration, the h=1 reads are issued behind it and pinned there, and every wait
    // that can fire lands 30-40 MFMAs after the reads it covers.
    UOC_MFMA_E(wa0, xb0, x)
    if (VARIANT < 3) UOC_FRAG2(s_cur, 1, wa1, xb1)
    __builtin_amdgcn_sched_barrier(0);
    UOC_MFMA_E(wa0, xb0, y)
    UOC_MFMA_E(wa0, xb0, z)
    UOC_MFMA_E(wa0, xb0, w)
    if (kc + 2 < nk && VARIANT < 1 && !early) UOC_ISSUE(kc + 2, s_nn)
    if (kc + 2 < nk) UOC_ADVANCE()
    if (VARIANT < 2) {
      if (kc + 2 < nk)
        wait_vmcnt<NPASS>();  // chunk kc+1 has landed; chunk kc+2 may still be in flight
      else
        wait_vmcnt<0>();
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) as a BUILTIN so hipcc's scoreboard sees the h=1 reads retire
      __builtin_amdgcn_s_barrier();
    }
    if (kc + 1 < nk && VARIANT < 3) UOC_FRAG2(s_nxt, 0, wa0, xb0)
    __builtin_amdgcn_sched_barrier(0);
    UOC_MFMA_E(wa1, xb1, x)
    UOC_MFMA_E(wa1, xb1, y)
    UOC_MFMA_E(wa1, xb1, z)
    UOC_MFMA_E(wa1, xb1, w)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): next chunk's h=0 fragments, issued 40 MFMAs ago
    __builtin_amdgcn_sched_barrier(0);
    const int tmp = s_cur;
    s_cur = s_nxt;
    s_nxt = s_nn;
    s_nn = tmp;
  }
#undef UOC_ADVANCE
#undef UOC_ISSUE
#undef UOC_ISSUE_LEAN
#undef UOC_ISSUE_ONE
#undef UOC_FRAG2
#undef UOC_MFMA_E

#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = n0 + wn * WN + 16 * j + 4 * q;
    const float4 bv = bias ? *reinterpret_cast<const float4 *>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * WM + 16 * i + t;
      if (m < M) {
        float4 v = make_float4(acc[j][i][0] + bv.x, acc[j][i][1] + bv.y, acc[j][i][2] + bv.z, acc[j][i][3] + bv.w);
        if (res) {
          const float4 rv = *reinterpret_cast<const float4 *>(res + (size_t)m * p.Cout + co);
          v.x += rv.x;
          v.y += rv.y;
          v.z += rv.z;
          v.w += rv.w;
        }
        if (p.relu) {
          v.x = fmaxf(v.x, 0.f);
          v.y = fmaxf(v.y, 0.f);
          v.z = fmaxf(v.z, 0.f);
          v.w = fmaxf(v.w, 0.f);
        }
        *reinterpret_cast<float4 *>(out + (size_t)m * p.Cout + co) = v;
      }
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool STEM, int VARIANT = 0>
static int launch_glds(const ConvParams &p, hipStream_t st, int kc) {
  const int M = p.B * p.Ho * p.Wo;
  const double taps = (double)p.KH * p.KW;
  const double flops = 2.0 * M * p.Cout * (STEM ? 3.0 : (double)p.Cin) * taps * p.G;
  const double bytes = 4.0 * p.G * ((double)p.B * p.H * p.W * (STEM ? 3 : p.Cin) + taps * p.Cout * (STEM ? 3 : p.Cin) +
                                    (double)M * p.Cout * (p.res ? 2 : 1));
  const ProfTag tag = {{p.prof_kc >= 0 ? p.prof_tag[0] : M, p.prof_kc >= 0 ? p.prof_tag[1] : p.Cin,
                        p.prof_kc >= 0 ? p.prof_tag[2] : p.Cout, p.prof_kc >= 0 ? p.prof_tag[3] : p.dil}};
  ProfScope prof(p.prof_kc >= 0 ? p.prof_kc : kc, st, p.prof_kc >= 0 ? p.prof_flops : flops, bytes, tag);
  const int mtiles = (M + BM - 1) / BM, ntiles = p.Cout / BN;
  const size_t lds = (size_t)3 * (BM + BN) * BK * sizeof(float);
  static DeviceOnce attr_set;
  if (!attr_set.done()) {
    UOC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_glds_kernel<BM, BN, WAVES_M, WAVES_N, STEM, VARIANT>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.mark();
  }
  const int total = mtiles * ntiles * p.G;
  hipLaunchKernelGGL((conv_glds_kernel<BM, BN, WAVES_M, WAVES_N, STEM, VARIANT>), dim3(((total + 7) / 8) * 8),
                     dim3(WAVES_M * WAVES_N * 64), lds, st, p, ntiles, mtiles);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

// ---- tile configurations ---------------------------------------------------------------
struct TileCfg {
  int BM, BN, threads;
  float penalty;  // relative per-flop cost of the tile shape (smaller wave tiles re-read more LDS)
};
static const TileCfg kCfgs[] = {
    {160, 128, 512, 1.00f},  // 0: 2x4 waves, wave tile 80x32
    {80, 128, 512, 1.06f},   // 1: 1x8 waves, wave tile 80x16
    {160, 64, 512, 1.06f},   // 2: 2x4 waves, wave tile 80x16
    {80, 64, 256, 1.12f},    // 3: 1x4 waves, wave tile 80x16
};
constexpr int kNumCfg = sizeof(kCfgs) / sizeof(kCfgs[0]);
constexpr int kNumCU = 256;


static int pick_cfg(const ConvParams &p) {
  const int M = p.B * p.Ho * p.Wo;
  int best = -1;
  double best_cost = 0;
  for (int c = 0; c < kNumCfg; ++c) {
    if (p.Cout % kCfgs[c].BN) continue;
    const long blocks = (long)((M + kCfgs[c].BM - 1) / kCfgs[c].BM) * (p.Cout / kCfgs[c].BN) * p.G;
    const long rounds = (blocks + kNumCU - 1) / kNumCU;
    const double cost = (double)rounds * kCfgs[c].BM * kCfgs[c].BN * kCfgs[c].penalty;
    if (best < 0 || cost < best_cost) {
      best = c;
      best_cost = cost;
    }
  }
  return best;
}

// ---- per-shape choice of (tile configuration, staging variant) --------------------------------
// All candidates accumulate every output element in the same K order, so they are bit-identical;
// the choice is purely a speed matter.  First use of a layer shape times the valid candidates
// (3 launches each, HIP events) and caches the winner; a shape that differs only in batch size
// (stage 2: one crop per ROI) reuses the nearest tuned neighbour instead of re-tuning.
// UOC_CONV_AUTOTUNE=0 falls back to the static cost model (development builds: UOC_CONV_CFG / UOC_CONV_GLDS pin a choice).
struct Choice {
  int cfg;
  int glds;   // 1 = the LDS-DMA kernel (always, in the shipped library); 0 = the register-staged kernel (dev builds)
};

static const int kBmWide[4] = {96, 128, 160, 192};  // 2x4-wave families (wave tile BM/2 x 32|16)
static const int kBmNarrow[3] = {64, 80, 96};       // 1x8 / 1x4-wave families (wave tile BM x 16)

static int pick_bm(int M, int ntiles, int G, int BN, const int *cands, int n) {
  int best = cands[n - 1];
  double best_cost = -1;
  for (int i = 0; i < n; ++i) {
    const int bm = cands[i];
    const long blocks = (long)((M + bm - 1) / bm) * ntiles * G;
    // whole rounds of CUs — also with a second stream on the device: choosing by total work alone ("the other stream
    // fills the tail") measured 4 % slower at two streams x one frame (round 2)
    const long rounds = (blocks + kNumCU - 1) / kNumCU;
    // per-flop cost grows mildly as the tile shrinks (operand re-reads per MFMA)
    const double cost = (double)rounds * bm * BN * (1.0 + 0.05 * ((double)cands[n - 1] / bm - 1.0));
    if (best_cost < 0 || cost <= best_cost) {
      best = bm;
      best_cost = cost;
    }
  }
  return best;
}

static int launch_choice(const ConvParams &p, hipStream_t st, Choice c) {
  if (c.glds) {
    // Each tile family (wave layout x BN) is instantiated for several pixel-tile heights; the one
    // that packs (m-tiles x n-tiles x branches) best into whole rounds of 256 CUs is used, so the
    // stage-2 shapes (784 x #ROIs pixels) do not lose 30-45 % to tile quantisation.
    const int M = p.B * p.Ho * p.Wo;
    switch (c.cfg) {
      case 0:
        switch (pick_bm(M, p.Cout / 128, p.G, 128, kBmWide, 4)) {
          case 96: return launch_glds<96, 128, 2, 4, false>(p, st, KC_GLDS_160x128);
          case 128: return launch_glds<128, 128, 2, 4, false>(p, st, KC_GLDS_160x128);
          case 192: return launch_glds<192, 128, 2, 4, false>(p, st, KC_GLDS_160x128);
          default: return launch_glds<160, 128, 2, 4, false>(p, st, KC_GLDS_160x128);
        }
      case 1:
        switch (pick_bm(M, p.Cout / 128, p.G, 128, kBmNarrow, 3)) {
          case 64: return launch_glds<64, 128, 1, 8, false>(p, st, KC_GLDS_80x128);
          case 96: return launch_glds<96, 128, 1, 8, false>(p, st, KC_GLDS_80x128);
          default: return launch_glds<80, 128, 1, 8, false>(p, st, KC_GLDS_80x128);
        }
      case 2:
        switch (pick_bm(M, p.Cout / 64, p.G, 64, kBmWide, 4)) {
          case 96: return launch_glds<96, 64, 2, 4, false>(p, st, KC_GLDS_160x64);
          case 128: return launch_glds<128, 64, 2, 4, false>(p, st, KC_GLDS_160x64);
          case 192: return launch_glds<192, 64, 2, 4, false>(p, st, KC_GLDS_160x64);
          default: return launch_glds<160, 64, 2, 4, false>(p, st, KC_GLDS_160x64);
        }
      case 3:
        switch (pick_bm(M, p.Cout / 64, p.G, 64, kBmNarrow, 3)) {
          case 64: return launch_glds<64, 64, 1, 4, false>(p, st, KC_GLDS_80x64);
          case 96: return launch_glds<96, 64, 1, 4, false>(p, st, KC_GLDS_80x64);
          default: return launch_glds<80, 64, 1, 4, false>(p, st, KC_GLDS_80x64);
        }
    }
  }
  set_error("conv: bad choice cfg=%d", c.cfg);
  return UOC_EINVAL;
}

struct TuneKey {
  int G, B, H, W, Cin, Cout, K, stride, dil;
  bool same_layer(const TuneKey &o) const {
    return G == o.G && H == o.H && W == o.W && Cin == o.Cin && Cout == o.Cout && K == o.K && stride == o.stride &&
           dil == o.dil;
  }
};
struct TuneEntry {
  TuneKey key;
  Choice choice;
};
static TuneEntry g_tuned[256];
static int g_ntuned = 0;

// Optional persistence (UOC_CONV_TUNE_CACHE=<file>): choices measured by one process are reused by
// the next, e.g. so that a profiled run contains no tuning launches.
static void tune_cache_load() {
  const char *path = getenv("UOC_CONV_TUNE_CACHE");
  if (!path) return;
  FILE *f = fopen(path, "r");
  if (!f) return;
  TuneEntry e;
  while (g_ntuned < 256 && fscanf(f, "%d %d %d %d %d %d %d %d %d %d %d", &e.key.G, &e.key.B, &e.key.H, &e.key.W,
                                  &e.key.Cin, &e.key.Cout, &e.key.K, &e.key.stride, &e.key.dil, &e.choice.cfg,
                                  &e.choice.glds) == 11) {
    e.choice.glds = 1;   // a cache written by a development build may name the register-staged kernel
    if (e.choice.cfg >= 0 && e.choice.cfg < kNumCfg) g_tuned[g_ntuned++] = e;
  }
  fclose(f);
}
static void tune_cache_append(const TuneEntry &e) {
  const char *path = getenv("UOC_CONV_TUNE_CACHE");
  if (!path) return;
  FILE *f = fopen(path, "a");
  if (!f) return;
  fprintf(f, "%d %d %d %d %d %d %d %d %d %d %d\n", e.key.G, e.key.B, e.key.H, e.key.W, e.key.Cin, e.key.Cout, e.key.K,
          e.key.stride, e.key.dil, e.choice.cfg, e.choice.glds);
  fclose(f);
}

static std::mutex g_tune_mutex;   // the tuner's table is process-wide (choices are per layer shape, valid on every
                                  // identical device) and may be reached from several host threads
static Choice choose(const ConvParams &p, hipStream_t st, int glds_default) {
  std::lock_guard<std::mutex> lock(g_tune_mutex);
  static int autotune = -1;
  if (autotune < 0) {
    const char *e = getenv("UOC_CONV_AUTOTUNE");
    autotune = e ? atoi(e) : 1;
    tune_cache_load();
  }
  Choice stat = {pick_cfg(p), 1};   // one kernel family: the tuner only chooses the tile
  if (!autotune) return stat;
  const TuneKey key = {p.G, p.B, p.H, p.W, p.Cin, p.Cout, p.KH, p.stride, p.dil};
  int nearest = -1;
  for (int i = 0; i < g_ntuned; ++i) {
    if (!g_tuned[i].key.same_layer(key)) continue;
    if (g_tuned[i].key.B == key.B) return g_tuned[i].choice;
    if (nearest < 0 || abs(g_tuned[i].key.B - key.B) < abs(g_tuned[nearest].key.B - key.B)) nearest = i;
  }
  if (nearest >= 0) {
    // same layer, other batch: reuse if the tile is still legal, never re-tune mid-stream
    return g_tuned[nearest].choice;
  }
  if (g_ntuned >= 256 || !p.tune || stream_is_capturing(st)) return stat;   // the tuner times launches with events: never inside a capture
  const bool prof_was = g_prof_enabled;
  g_prof_enabled = false;
  hipEvent_t e0, e1;
  Choice best = stat;
  float best_ms = 1e30f;
  if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
    for (int cfg = 0; cfg < kNumCfg; ++cfg) {
      if (p.Cout % kCfgs[cfg].BN) continue;
      for (int gl = 1; gl < 2; ++gl) {
        const Choice c = {cfg, gl};
        if (launch_choice(p, st, c) != UOC_OK) continue;  // warm-up (also sets the LDS attribute)
        (void)hipEventRecord(e0, st);
        for (int r = 0; r < 3; ++r) (void)launch_choice(p, st, c);
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) continue;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best_ms) {
          best_ms = ms;
          best = c;
        }
      }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  g_prof_enabled = prof_was;
  g_tuned[g_ntuned].key = key;
  g_tuned[g_ntuned].choice = best;
  tune_cache_append(g_tuned[g_ntuned]);
  ++g_ntuned;
  if (getenv("UOC_CONV_VERBOSE"))
    fprintf(stderr, "[uoc] conv G%d B%d %dx%d %d->%d k%d s%d d%d : cfg %d glds %d (%.1f us)\n", p.G, p.B, p.H, p.W, p.Cin,
            p.Cout, p.KH, p.stride, p.dil, best.cfg, best.glds, best_ms * 1e3f / 3);
  return best;
}

int launch_conv(const ConvParams &p, hipStream_t st) {
  UOC_REQUIRE(p.in && p.w && p.out && (p.bias || !p.stem), "conv: null pointer");
  UOC_REQUIRE(p.G >= 1 && p.B >= 1 && p.H >= 1 && p.W >= 1, "conv: bad shape");
  UOC_REQUIRE((long)p.B * p.H * p.W * p.Cin < (1l << 31) && (long)p.B * p.Ho * p.Wo * p.Cout < (1l << 31),
              "conv: tensor too large for 32-bit indexing");
  const int use_glds = 1;
  if (p.stem) {
    UOC_REQUIRE(p.Cin == 4 && p.KH == 7 && p.KW == 7 && p.stride == 2 && p.pad == 3 && p.dil == 1 && p.Cout == 64,
                "conv: stem path is 7x7 s2 p3, NHWC4 -> 64 only");
    UOC_REQUIRE((size_t)p.B * p.H * p.W * p.Cin * 4 + (size_t)(p.pad * p.W + p.pad) * p.Cin * 4 < (1ull << 31),
                "conv: the stem's input exceeds the 2 GB a 32-bit buffer offset addresses");
    return launch_glds<160, 64, 2, 4, true>(p, st, KC_CONV_STEM);
  }
  UOC_REQUIRE(p.Cin % BK == 0, "conv: Cin=%d must be a multiple of %d", p.Cin, BK);
  UOC_REQUIRE(p.Cout % 64 == 0, "conv: Cout=%d must be a multiple of 64", p.Cout);
  // the LDS-DMA kernel addresses a group's input through a 32-bit buffer offset: a batch beyond 2 GB per group runs as two
  // launches over halves of the batch (an output pixel never depends on another image: bit-identical)
  const size_t halo = (size_t)(p.pad * p.W + p.pad) * p.Cin * 4;
  static EnvInt limit_mb("UOC_SPLIT_MAX_MB", 0);   // tests: a smaller limit, to exercise the split on small batches (same results)
  const size_t limit = limit_mb.get() > 0 ? (size_t)limit_mb.get() << 20 : (1ull << 31);
  const bool glds_ok = (size_t)p.B * p.H * p.W * p.Cin * 4 + halo < limit;
  if (!glds_ok) {
    UOC_REQUIRE(p.B > 1 || limit_mb.get() > 0, "conv: one image's input exceeds the 2 GB a 32-bit buffer offset addresses");
    if (p.B == 1) {   // (only reachable with the test limit) a single image per group: one launch per group
      ConvParams q = p;
      for (int g = 0; g < p.G; ++g) {
        q.G = 1;
        q.in = p.in + (size_t)g * p.H * p.W * p.Cin;
        q.out = p.out + (size_t)g * p.Ho * p.Wo * p.Cout;
        if (p.res) q.res = p.res + (size_t)g * p.Ho * p.Wo * p.Cout;
        q.w = p.w + (size_t)g * p.KH * p.KW * p.Cout * p.Cin;
        if (p.bias) q.bias = p.bias + (size_t)g * p.Cout;
        // the REAL limit of the 32-bit buffer offset, whatever the test limit above says
        UOC_REQUIRE((size_t)q.B * q.H * q.W * q.Cin * 4 + halo < (1ull << 31),
                    "conv: one image's input exceeds the 2 GB a 32-bit buffer offset addresses");
        const Choice ch1 = choose(q, st, use_glds);
        if (ch1.cfg < 0) {
          set_error("conv: no tile configuration for Cout=%d", p.Cout);
          return UOC_EINVAL;
        }
        if (int rc = launch_choice(q, st, ch1)) return rc;
      }
      return UOC_OK;
    }
    // groups are strided by the FULL batch: a slice keeps the group stride only if it is addressed per group
    for (int g = 0; g < p.G; ++g)
      for (int b0 = 0; b0 < p.B;) {
        ConvParams q = p;
        q.G = 1;
        q.B = (p.B + 1) / 2 < p.B - b0 ? (p.B + 1) / 2 : p.B - b0;
        const size_t ii = ((size_t)g * p.B + b0) * p.H * p.W * p.Cin, oo = ((size_t)g * p.B + b0) * p.Ho * p.Wo * p.Cout;
        q.in = p.in + ii;
        q.out = p.out + oo;
        if (p.res) q.res = p.res + oo;
        q.w = p.w + (size_t)g * p.KH * p.KW * p.Cout * p.Cin;
        if (p.bias) q.bias = p.bias + (size_t)g * p.Cout;
        if (int rc = launch_conv(q, st)) return rc;
        b0 += q.B;
      }
    return UOC_OK;
  }
  UOC_REQUIRE((size_t)p.B * p.H * p.W * p.Cin * 4 + halo < (1ull << 31),
              "conv: a group's input exceeds the 2 GB a 32-bit buffer offset addresses");
  const Choice ch = choose(p, st, use_glds);
  if (ch.cfg < 0) {
    set_error("conv: no tile configuration for Cout=%d", p.Cout);
    return UOC_EINVAL;
  }
  return launch_choice(p, st, ch);
}

// ---- NCHW(3) -> NHWC4 ------------------------------------------------------------------
__global__ __launch_bounds__(256) void nchw3_to_nhwc4_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                             int HW, int total) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / HW, p = i - b * HW;
    const float *src = in + (size_t)b * 3 * HW + p;
    *reinterpret_cast<float4 *>(out + (size_t)i * 4) = make_float4(src[0], src[HW], src[2 * HW], 0.f);
  }
}

int launch_nchw3_to_nhwc4(const float *in, float *out, int B, int H, int W, hipStream_t st) {
  const int total = B * H * W;
  int blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  ProfScope prof(KC_NET_MISC, st, 0.0, 28.0 * total);
  hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3(blocks), dim3(256), 0, st, in, out, H * W, total);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

// ---- MaxPool2d(kernel 3, stride 2, padding 1), NHWC ------------------------------------
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                           int H, int W, int C4, int Ho, int Wo, int total) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int c4 = i % C4;
    int r = i / C4;
    const int ox = r % Wo;
    r /= Wo;
    const int oy = r % Ho;
    const int b = r / Ho;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = 2 * oy - 1 + dy;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = 2 * ox - 1 + dx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const float4 v = *reinterpret_cast<const float4 *>(in + (((size_t)b * H + iy) * W + ix) * C4 * 4 + c4 * 4);
        m.x = fmaxf(m.x, v.x);
        m.y = fmaxf(m.y, v.y);
        m.z = fmaxf(m.z, v.z);
        m.w = fmaxf(m.w, v.w);
      }
    }
    *reinterpret_cast<float4 *>(out + (size_t)i * 4) = m;
  }
}

int launch_maxpool3x3s2(const float *in, float *out, int n_img, int H, int W, int C, int Ho, int Wo, hipStream_t st) {
  UOC_REQUIRE(C % 4 == 0, "maxpool: C %% 4 != 0");
  const int total = n_img * Ho * Wo * (C / 4);
  int blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  ProfScope prof(KC_NET_MISC, st, 0.0, 4.0 * ((double)n_img * H * W * C + (double)total * 4));
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(blocks), dim3(256), 0, st, in, out, H, W, C / 4, Ho, Wo, total);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

// ---- fused head: (branch a + branch b) -> x8 bilinear (align_corners=True) -> L2 normalise ---
// Upsampling is linear, so the two branches are added at 1/8 resolution first
// (resnet_dilated.py:325 per branch, SEG.py:106-108 add, SEG.py:113-114 normalise).
// 16 lanes own one output pixel (float4 of channels each); a wave writes 1 KiB contiguous.
__global__ __launch_bounds__(256) void head_kernel(const float *__restrict__ fa, const float *__restrict__ fb,
                                                   float *__restrict__ embed, int B, int h, int w, int H, int W,
                                                   float sy, float sx) {
  const int lane = threadIdx.x & 63;
  const int t = lane & 15, g = lane >> 4;
  // 32-bit index arithmetic (launch_head checks B * H * W < 2^31): the kernel stores 16 bytes per thread and iteration, and the
  // 64-bit division it used to do for (image, row, column) cost more instructions than everything else in the loop
  const unsigned HW = (unsigned)(H * W);
  const unsigned total = (unsigned)B * HW;
  const unsigned gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const unsigned nw = (gridDim.x * blockDim.x) >> 6;
  for (unsigned p4 = gw * 4; p4 < total; p4 += nw * 4) {
    const unsigned pix = p4 + g;
    if (pix >= total) continue;
    const unsigned b = pix / HW;
    const unsigned r = pix - b * HW;
    const int oy = (int)(r / (unsigned)W), ox = (int)(r - (unsigned)oy * (unsigned)W);
    const float fy = sy * (float)oy, fx = sx * (float)ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const size_t base = (size_t)b * h * w * 64 + 4 * t;
    auto ld = [&](int y, int x) {
      const size_t o = base + ((size_t)y * w + x) * 64;
      const float4 u = *reinterpret_cast<const float4 *>(fa + o);
      if (!fb) return u;  // single-backbone modes (COLOR / DEPTH / early fusion)
      const float4 v = *reinterpret_cast<const float4 *>(fb + o);
      return make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    };
    const float4 v00 = ld(y0, x0), v01 = ld(y0, x1), v10 = ld(y1, x0), v11 = ld(y1, x1);
    float4 o;
    o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
    o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
    o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
    o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
    float ss = o.x * o.x;
    ss = fmaf(o.y, o.y, ss);
    ss = fmaf(o.z, o.z, ss);
    ss = fmaf(o.w, o.w, ss);
    ss = row16_sum(ss);
    const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    o.x *= inv;
    o.y *= inv;
    o.z *= inv;
    o.w *= inv;
    *reinterpret_cast<float4 *>(embed + (size_t)pix * 64 + 4 * t) = o;
  }
}

// 'cat' fusion (SEG.py:109-110,113-114): embed[b][0][p][:] / embed[b][1][p][:] = the two upsampled branches divided
// by the norm of their concatenation.
__global__ __launch_bounds__(256) void head_cat_kernel(const float *__restrict__ fa, const float *__restrict__ fb,
                                                       float *__restrict__ embed, int B, int h, int w, int H, int W,
                                                       float sy, float sx) {
  const int lane = threadIdx.x & 63;
  const int t = lane & 15, g = lane >> 4;
  const int HW = H * W;
  const long total = (long)B * HW;
  const long gw = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nw = ((long)gridDim.x * blockDim.x) >> 6;
  for (long p4 = gw * 4; p4 < total; p4 += nw * 4) {
    const long pix = p4 + g;
    if (pix >= total) continue;
    const int b = (int)(pix / HW);
    const int r = (int)(pix - (long)b * HW);
    const int oy = r / W, ox = r - oy * W;
    const float fy = sy * (float)oy, fx = sx * (float)ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const size_t base = (size_t)b * h * w * 64 + 4 * t;
    float4 o[2];
    float ss = 0.f;
#pragma unroll
    for (int br = 0; br < 2; ++br) {
      const float *f = br ? fb : fa;
      const float4 v00 = *reinterpret_cast<const float4 *>(f + base + ((size_t)y0 * w + x0) * 64);
      const float4 v01 = *reinterpret_cast<const float4 *>(f + base + ((size_t)y0 * w + x1) * 64);
      const float4 v10 = *reinterpret_cast<const float4 *>(f + base + ((size_t)y1 * w + x0) * 64);
      const float4 v11 = *reinterpret_cast<const float4 *>(f + base + ((size_t)y1 * w + x1) * 64);
      o[br].x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
      o[br].y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
      o[br].z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
      o[br].w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
      ss = fmaf(o[br].x, o[br].x, ss);
      ss = fmaf(o[br].y, o[br].y, ss);
      ss = fmaf(o[br].z, o[br].z, ss);
      ss = fmaf(o[br].w, o[br].w, ss);
    }
    ss = row16_sum(ss);
    const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int br = 0; br < 2; ++br) {
      o[br].x *= inv;
      o[br].y *= inv;
      o[br].z *= inv;
      o[br].w *= inv;
      *reinterpret_cast<float4 *>(embed + (((size_t)b * 2 + br) * HW + r) * 64 + 4 * t) = o[br];
    }
  }
}

int launch_head(const float *fa, const float *fb, float *embed, int B, int h, int w, int H, int W, int cat,
                hipStream_t st) {
  const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
  const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const long total = (long)B * H * W;
  // (the kernel is bound by its ~260 instructions per 16 bytes stored — 16 lanes share a pixel's index arithmetic — not by latency:
  // a grid of 32 768 blocks instead of 4 096 measured the same, 122 vs 118 us per four frames)
  constexpr long kMaxBlocks = 4096;
  UOC_REQUIRE(cat || total + 4l * kMaxBlocks * 4 < (1l << 31), "head: B*H*W = %ld does not fit the kernel's 32-bit pixel index", total);
  long blocks = (total / 4 + 3) / 4;  // 4 waves per block, 4 pixels per wave step
  if (blocks > kMaxBlocks) blocks = kMaxBlocks;
  if (blocks < 1) blocks = 1;
  ProfScope prof(KC_HEAD, st, 0.0, 4.0 * 64 * ((cat ? 2.0 : 1.0) * (double)total + (fb ? 2.0 : 1.0) * B * h * w));
  if (cat)
    hipLaunchKernelGGL(head_cat_kernel, dim3((unsigned)blocks), dim3(256), 0, st, fa, fb, embed, B, h, w, H, W, sy, sx);
  else
    hipLaunchKernelGGL(head_kernel, dim3((unsigned)blocks), dim3(256), 0, st, fa, fb, embed, B, h, w, H, W, sy, sx);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

}  // namespace uoc
