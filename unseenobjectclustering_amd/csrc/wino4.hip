// Winograd F(4x4, 3x3) convolution for the 3x3 stride-1 (possibly dilated) layers of the backbone: 36 products per
// 16 outputs instead of the direct form's 144 (and of F(2x2,3x3)'s 64) — 4x fewer matrix-core flops in exact-fp32 MFMA
// arithmetic.  Transform matrices, the choice of interpolation points and the measured rounding error: wino4_math.h.
//
// Unlike csrc/wino.hip (F(2x2): 16 frequencies, output transform folded into the GEMM's registers) the 36 frequency
// planes are kept in HBM and the layer runs as three kernels:
//   wino4_input_kernel    V[(g,xi)][tile][cin]  = B^T d B            elementwise, HBM-bound
//   wino4_gemm_kernel     M[(g,xi)][tile][cout] = V[(g,xi)] U[(g,xi)]^T   72 independent GEMMs, MFMA-bound
//   wino4_output_kernel   out = relu(A^T M A + bias (+ residual))    elementwise, HBM-bound
// (plus wino4_weight_kernel once per layer at uoc_net_finalize).  Per layer the transforms move 2.25x the activation
// in and 2.25x out — less than F(2x2)'s 4x V round trip — and the GEMM is a plain [tiles x Cin] x [Cin x Cout] product
// per frequency, so it gets the large 160x128 block tile of the direct kernel (wave tile 80x32: 0.175 fragment reads
// per MFMA against 0.30 in wino_gemm_kernel, whose 4 output accumulators per product accumulator cap its tile).
//
// The GEMM kernel is persistent over work items (plane = (group, frequency), m-tile, n-tile): a block walks its items
// with ONE software pipeline — the LDS-DMA ring keeps fetching the next item's first chunks while the last chunks of the
// current item are multiplied and its accumulators are stored — because a single item is only Cin/32 = 4..16 K-chunks
// long and a launch-per-plane GEMM would spend a third of its time filling and draining.
#include "conv.h"
#include "dma.h"
#include "prof.h"
#include "wino4_math.h"

#include <stdlib.h>

namespace uoc {

constexpr int W4BK = 32;  // cin chunk (floats): one 128-byte row piece per DMA lane group

// ---- elementwise kernels: thin grid-stride wrappers around the bodies in wino4_math.h ---------------------------
__global__ __launch_bounds__(256) void wino4_weight_kernel(const float *__restrict__ w, float *__restrict__ U, int G,
                                                           int Cout, int Cin) {
  const long total = (long)G * Cout * Cin;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin);
    const int co = (int)((i / Cin) % Cout);
    const int g = (int)(i / ((long)Cin * Cout));
    wino4_weight_body(w, U, G, Cout, Cin, g, co, ci);
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void wino4_input_kernel(const float *__restrict__ in, float *__restrict__ V,
                                                          Wino4Geom geo, int G, int C) {
  const int CV = C / VEC;
  const long total = (long)G * geo.NT * CV;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(idx % CV);
    const int tau = (int)((idx / CV) % geo.NT);
    const int g = (int)(idx / ((long)CV * geo.NT));
    wino4_input_body<VEC>(in, V, geo, C, g, tau, cv);
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void wino4_output_kernel(const float *__restrict__ M, const float *__restrict__ bias,
                                                           const float *__restrict__ res, float *__restrict__ out,
                                                           Wino4Geom geo, int G, int Cout, int relu) {
  const int CV = Cout / VEC;
  const long total = (long)G * geo.NT * CV;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(idx % CV);
    const int tau = (int)((idx / CV) % geo.NT);
    const int g = (int)(idx / ((long)CV * geo.NT));
    wino4_output_body<VEC>(M, bias, res, out, geo, Cout, relu, g, tau, cv);
  }
}


// ---- output transform of layer L fused with the input transform of layer L+1 (same geometry) --------------------------
// Inside a chain of Winograd layers of one resolution and dilation the NHWC activation between two layers is only ever
// read back as 6x6 patches.  One block owns one dilation-phase image of one (group, image) and CS channels: it turns the
// phase image's M tiles into outputs (A^T M A + bias (+ residual), ReLU: wino4_math.h, the very functions of the two
// separate kernels, so the results are bit-identical), keeps them in LDS with the one-pixel zero frame the next
// convolution's padding asks for, and transforms the 6x6 patches straight into the next layer's V planes.  A phase image
// has no neighbours (its halo is padding), so no tile is computed twice.  The activation itself is written to HBM only
// when somebody else needs it (the next block's residual): per pair of layers 4.5 instead of 6.5-7.75 activation sizes
// move.  Layout of the LDS image: [(4 TH + 2) rows][(4 TW + 2) columns][CS / 4] float4.
template <int VEC>
__global__ __launch_bounds__(VEC == 4 ? 256 : 512) void wino4_mid_kernel(const float *__restrict__ M, const float *__restrict__ bias,
                                                        const float *__restrict__ res, float *__restrict__ yout,
                                                        float *__restrict__ V, Wino4Geom geo, int G, int C, int relu, int CS,
                                                        int sib, int units) {
  typedef typename W4Vec<VEC>::type T;
  extern __shared__ __attribute__((aligned(16))) float ysm_raw[];
  T *ysm = reinterpret_cast<T *>(ysm_raw);
  const int Q = CS / VEC, slices = C / CS;
  const int Wl = 4 * geo.TW + 2, Hl = 4 * geo.TH + 2;
  // blocks are dealt to the 8 XCDs round-robin; `sib` sibling slices share 128-byte lines of M and V, so they are
  // mapped to consecutive blocks of ONE XCD (one L2 fetches the line once)
  const int x8 = blockIdx.x & 7, j8 = blockIdx.x >> 3;
  const int unit = ((j8 / sib) * 8 + x8) * sib + j8 % sib;
  if (unit >= units) return;
  const int slice = unit % slices;
  int u = unit / slices;
  const int px = u % geo.d;
  u /= geo.d;
  const int py = u % geo.d;
  u /= geo.d;
  const int b = u % geo.B, g = u / geo.B;
  const int ntile = geo.TH * geo.TW, items = ntile * Q;
  const int tile0 = ((b * geo.d + py) * geo.d + px) * ntile;
  const size_t plane = (size_t)geo.NT * C;
  const size_t gsz = (size_t)geo.Bg * geo.H * geo.W * C;
  const int tid = threadIdx.x, nthr = blockDim.x;

  for (int i = tid; i < Hl * Wl * Q; i += nthr) w4_zero(ysm[i]);
  __syncthreads();
  for (int it = tid; it < items; it += nthr) {
    const int cq = it % Q, t = it / Q, ty = t / geo.TW, tx = t - ty * geo.TW;
    const int ch = slice * CS + VEC * cq;
    const float *src = M + (size_t)g * 36 * plane + (size_t)(tile0 + t) * C + ch;
    T m[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) m[k] = *reinterpret_cast<const T *>(src + (size_t)k * plane);
    T yv[4][4];
    wino4_output_tile(m, yv);
    T bv;
    if (bias)
      bv = *reinterpret_cast<const T *>(bias + (size_t)g * C + ch);
    else
      w4_zero(bv);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int y = py + (4 * ty + a) * geo.d;
      if (y >= geo.H) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int x = px + (4 * tx + e) * geo.d;
        if (x >= geo.W) continue;
        const size_t o = (size_t)g * gsz + (((size_t)b * geo.H + y) * geo.W + x) * C + ch;
        const T v = wino4_epilogue(yv[a][e], bv, res ? res + o : nullptr, relu);
        ysm[((4 * ty + a + 1) * Wl + 4 * tx + e + 1) * Q + cq] = v;
        if (yout) *reinterpret_cast<T *>(yout + o) = v;
      }
    }
  }
  __syncthreads();
  for (int it = tid; it < items; it += nthr) {
    const int cq = it % Q, t = it / Q, ty = t / geo.TW, tx = t - ty * geo.TW;
    T d[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) d[i][j] = ysm[((4 * ty + i) * Wl + 4 * tx + j) * Q + cq];
    T v[36];
    wino4_input_tile(d, v);
    float *dst = V + (size_t)g * 36 * plane + (size_t)(tile0 + t) * C + slice * CS + VEC * cq;
#pragma unroll
    for (int k = 0; k < 36; ++k) *reinterpret_cast<T *>(dst + (size_t)k * plane) = v[k];
  }
}

// ---- the batched GEMM over (group, frequency) planes -------------------------------------------------------------
__device__ __forceinline__ f32x4 w4mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Block tile BM tiles x BN output channels, 8 waves as 2 (m) x 4 (n), wave tile (BM/2) x (BN/4); weights are the MFMA
// "A" operand so a lane ends with 4 consecutive output channels of one tile row (float4 stores into M).
// Operand staging = the LDS-DMA ring of conv_glds_kernel (3 stages, two chunks ahead, counted vmcnt, one raw barrier
// per chunk, XOR swizzle on the source slot and on the fragment read), fed through raw buffer descriptors: per DMA one
// loop-invariant lane offset (row of the block tile) + one scalar offset (plane, cin chunk).
//
// Work items are ordered (plane, m-tile, n-tile).  Each XCD takes a contiguous eighth of the list (so that the rows of
// one plane are pulled into ONE L2) and deals its items round-robin to its blocks: at any time the blocks of an XCD
// work on neighbouring items = the same one or two planes.
// -DUOC_W4_ABLATE=n (dev builds only, scripts/w4_ablate.sh; the results are WRONG, only the time means something): 1 = no
// barriers, 2 = no vmcnt waits, 4 = no LDS-DMA issue (profiles/r04_pmc_mfma.md).  (Dropping the epilogue stores is not an
// ablation: the compiler then removes the MFMAs.)
#ifndef UOC_W4_ABLATE
#define UOC_W4_ABLATE 0
#endif
#define W4_BARRIER() do { if (UOC_W4_ABLATE != 1) __builtin_amdgcn_s_barrier(); } while (0)
template <int N>
__device__ __forceinline__ void w4_wait_vmcnt() {
  if (UOC_W4_ABLATE != 2) wait_vmcnt<N>();
}

// ---- small-K layers (Cin = Cout = 64 / 128): plane GEMMs + output transform in ONE kernel, no M planes -------------------
// With K = 64 a plane GEMM does 32 flop per byte of V + M: the layer is HBM-bound and M (2.25 x the activation, written by
// the GEMM and read back by the output transform) is 40 % of its traffic.  Here a block owns 16 tiles x all output channels
// and walks the 36 planes of its branch: wave w owns output channels 16 w .. 16 w + 15 and keeps ALL 36 plane accumulators
// of its 16 x 16 tile in registers (144 VGPRs), so that after the last plane every lane holds the 36 frequencies of its
// (tile, 4 channels) and applies A^T . A, bias, residual and ReLU itself (wino4_math.h: the functions of the separate
// kernels; MFMA order per accumulator = cin order as in the plane GEMM: bit-identical results).  Operands go straight
// from L2 into registers (one plane ahead), no LDS: U (36 x C x C floats per branch, 0.6 / 2.4 MB) is re-read per 16
// tiles from L2 instead of M making a round trip through HBM.
template <int C>
__global__ __launch_bounds__(C * 4) __attribute__((amdgpu_waves_per_eu(2))) void wino4_small_kernel(const float *__restrict__ V, const float *__restrict__ U,
                                                            const float *__restrict__ bias, const float *__restrict__ res,
                                                            float *__restrict__ out, Wino4Geom geo, int G, int relu) {
  constexpr int KK = C / 16;   // float4 loads per operand row and plane (16 K-steps of 4 per load quartet)
  const int rtiles = (geo.NT + 15) >> 4;
  const int g = blockIdx.x / rtiles, rt = blockIdx.x - g * rtiles;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = lane & 15, q = lane >> 4;
  const int row = min(rt * 16 + t, geo.NT - 1);   // rows beyond the last tile read the last valid row, never stored
  const size_t plane_v = (size_t)geo.NT * C, plane_u = (size_t)C * C;
  const float *vp = V + (size_t)g * 36 * plane_v + (size_t)row * C + 4 * q;
  const float *up = U + (size_t)g * 36 * plane_u + (size_t)(16 * wave + t) * C + 4 * q;
  f32x4 acc[36];
  float4 wf[KK], xf[KK], wn[KK], xn[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    wf[kk] = *reinterpret_cast<const float4 *>(up + 16 * kk);
    xf[kk] = *reinterpret_cast<const float4 *>(vp + 16 * kk);
  }
#pragma unroll
  for (int xi = 0; xi < 36; ++xi) {
    if (xi + 1 < 36) {
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        wn[kk] = *reinterpret_cast<const float4 *>(up + (size_t)(xi + 1) * plane_u + 16 * kk);
        xn[kk] = *reinterpret_cast<const float4 *>(vp + (size_t)(xi + 1) * plane_v + 16 * kk);
      }
    }
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      a = w4mfma(wf[kk].x, xf[kk].x, a);
      a = w4mfma(wf[kk].y, xf[kk].y, a);
      a = w4mfma(wf[kk].z, xf[kk].z, a);
      a = w4mfma(wf[kk].w, xf[kk].w, a);
    }
    acc[xi] = a;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      wf[kk] = wn[kk];
      xf[kk] = xn[kk];
    }
  }
  const int tau = rt * 16 + t;
  if (tau >= geo.NT) return;
  float4 m[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) m[k] = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
  float4 yv[4][4];
  wino4_output_tile(m, yv);
  int b, oy, ox;
  wino4_decode(tau, geo, b, oy, ox);
  const int ch = 16 * wave + 4 * q;
  const size_t gsz = (size_t)geo.Bg * geo.H * geo.W * C;
  float4 bv;
  if (bias)
    bv = *reinterpret_cast<const float4 *>(bias + (size_t)g * C + ch);
  else
    w4_zero(bv);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int y = oy + a * geo.d;
    if (y >= geo.H) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int x = ox + e * geo.d;
      if (x >= geo.W) continue;
      const size_t o = (size_t)g * gsz + (((size_t)b * geo.H + y) * geo.W + x) * C + ch;
      *reinterpret_cast<float4 *>(out + o) = wino4_epilogue(yv[a][e], bv, res ? res + o : nullptr, relu);
    }
  }
}

// PAIR (round 4): a 4-stage ring and ONE barrier per TWO K-chunks.  The barrier costs 5-9 % of the kernel (15 % on 4-chunk
// items: timing ablation, profiles/r04_pmc_mfma.md); a pair of chunks is fetched two chunks ahead into the two stages the
// previous pair has just left, so the barrier at the end of a pair covers both hazards (the DMA'd rows of the next pair are
// visible to every wave; every wave has finished reading the stages the pair after next will overwrite).  The items of a
// block always hold an even number of chunks (Cin / 32 = 2, 4, 8, 16), so a pair never straddles two items.  Same MFMA
// order per accumulator as the single-chunk loop: bit-identical results.
template <int BM, int BN, bool PAIR>
__global__ __launch_bounds__(512) void wino4_gemm_kernel(const float *__restrict__ V, const float *__restrict__ U,
                                                         float *__restrict__ Mo, int NT, int Cin, int Cout, int planes,
                                                         int mtiles, int nt_shift) {
  constexpr int WM = BM / 2, WN = BN / 4;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int R = BM + BN;
  constexpr int RPP = 64;  // rows per DMA pass: 8 waves x 8 rows (1 KiB per wave-instruction)
  constexpr int NPA = (BM + RPP - 1) / RPP, NPW = (BN + RPP - 1) / RPP, NPASS = NPA + NPW;
  constexpr int STAGE = R * W4BK;
  static_assert(WM % 16 == 0 && WN % 16 == 0 && R % 8 == 0 && NPASS <= 8, "tile shape");

  extern __shared__ __attribute__((aligned(16))) float smem[];  // [3 | 4][R][32]

  const int ntiles = 1 << nt_shift;
  const int per_plane = mtiles << nt_shift;
  const int total = planes * per_plane;
  const int S = (total + 7) >> 3;                 // items per XCD slice
  const int xcd = blockIdx.x & 7, jloc = blockIdx.x >> 3, nbl = gridDim.x >> 3;
  const int slice_lo = xcd * S, slice_hi = min(total, slice_lo + S);
  const int first = slice_lo + jloc;
  if (first >= slice_hi) return;
  const int n_items = (slice_hi - first + nbl - 1) / nbl;
  const int cpt = Cin / W4BK;
  const int nchunks = n_items * cpt;

  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) char *)smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int t = lane & 15, q = lane >> 4;

  // The descriptors span ALL planes and no lane ever relies on the range check: measured on gfx950, a descriptor of one
  // plane's size with the plane offset in soffset returned zeros for every plane but the first (the check evidently
  // covers voffset + soffset here).  Tile rows beyond NT read the last valid row instead: their products land in
  // accumulator rows the epilogue never stores.
  const unsigned plane_a_bytes = (unsigned)((size_t)NT * Cin * 4), plane_w_bytes = (unsigned)((size_t)Cout * Cin * 4);
  const v4i srd_a = make_srd(V, (unsigned)planes * plane_a_bytes);
  const v4i srd_w = make_srd(U, (unsigned)planes * plane_w_bytes);

  // per pass: first row of the wave's 8-row group in the stage (compile-time pattern), the lane's row / slot
  int l_r0[NPASS], l_row[NPASS], l_c4[NPASS];
#pragma unroll
  for (int j = 0; j < NPASS; ++j) {
    const bool isw = j >= NPA;
    int r0 = (isw ? BM + (j - NPA) * RPP : j * RPP) + wave * 8;
    if (!isw && r0 >= BM) r0 = BM - 8;  // surplus wave: re-copy the kind's last group (identical bytes)
    if (isw && r0 >= R) r0 = R - 8;
    l_r0[j] = r0;
    const int r = r0 + (lane >> 3);
    l_row[j] = isw ? r - BM : r;
    l_c4[j] = 4 * ((lane & 7) ^ ((r >> 1) & 7));
  }

  // ---- issue-side state: the item / chunk whose DMA is issued next ----
  int iss_plane, iss_rem, iss_cc = 0;
  {
    iss_plane = first / per_plane;
    iss_rem = first - iss_plane * per_plane;
  }
  unsigned l_voff[NPASS];
  unsigned soff_a, soff_w;
#define W4_ITEM_SETUP()                                                                                  \
  {                                                                                                      \
    const int mt_ = iss_rem >> nt_shift, nt_ = iss_rem & (ntiles - 1);                                   \
    _Pragma("unroll") for (int j = 0; j < NPASS; ++j) {                                                  \
      if (j < NPA) {                                                                                     \
        const int tau_ = min(mt_ * BM + l_row[j], NT - 1);                                               \
        l_voff[j] = (unsigned)((tau_ * Cin + l_c4[j]) * 4);                                              \
      } else {                                                                                           \
        l_voff[j] = (unsigned)(((nt_ * BN + l_row[j]) * Cin + l_c4[j]) * 4);                             \
      }                                                                                                  \
    }                                                                                                    \
    soff_a = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)iss_plane * plane_a_bytes));       \
    soff_w = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)iss_plane * plane_w_bytes));       \
  }
#define W4_ISSUE(STG)                                                                                    \
  {                                                                                                      \
    _Pragma("unroll") for (int j = 0; j < NPASS; ++j) {                                                  \
      const unsigned dst_ = lds_base + (unsigned)(((STG)*STAGE + l_r0[j] * W4BK) * sizeof(float));       \
      if (UOC_W4_ABLATE == 4) continue;                                                                  \
      if (j < NPA)                                                                                       \
        blds16(srd_a, l_voff[j], soff_a, dst_);                                                          \
      else                                                                                               \
        blds16(srd_w, l_voff[j], soff_w, dst_);                                                          \
    }                                                                                                    \
  }
  // advance the issue state by one chunk (next cin slice, or the first slice of the block's next item)
#define W4_ADVANCE()                                                                                     \
  {                                                                                                      \
    if (++iss_cc == cpt) {                                                                               \
      iss_cc = 0;                                                                                        \
      iss_rem += nbl;                                                                                    \
      while (iss_rem >= per_plane) {                                                                     \
        iss_rem -= per_plane;                                                                            \
        ++iss_plane;                                                                                     \
      }                                                                                                  \
      W4_ITEM_SETUP()                                                                                    \
    } else {                                                                                             \
      soff_a += W4BK * 4;                                                                                \
      soff_w += W4BK * 4;                                                                                \
    }                                                                                                    \
  }
#define W4_FRAG(STG, HH, WA, XB)                                                                         \
  {                                                                                                      \
    const float *base_ = smem + (STG)*STAGE;                                                             \
    const int slot_ = ((4 * (HH) + q) ^ ((t >> 1) & 7)) * 4;                                             \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) WA[j] =                                               \
        *reinterpret_cast<const float4 *>(base_ + (BM + wn * WN + 16 * j + t) * W4BK + slot_);           \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) XB[i] =                                               \
        *reinterpret_cast<const float4 *>(base_ + (wm * WM + 16 * i + t) * W4BK + slot_);                \
  }
#define W4_MFMA_E(WA, XB, E)                                                                                   \
  {                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) _Pragma("unroll") for (int i = 0; i < TM; ++i) acc[j][i] =  \
        w4mfma(WA[j].E, XB[i].E, acc[j][i]);                                                                   \
  }

  f32x4 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- compute-side state: the item whose chunks are being multiplied ----
  int cmp_plane = iss_plane, cmp_rem = iss_rem, cmp_cc = 0;

  float4 wa0[TN], xb0[TM], wa1[TN], xb1[TM];
  const bool early = wave < 4;         // waves w and w+4 share a SIMD: they issue their DMA bursts at different points
  auto epilogue = [&]() {   // item complete: store its accumulators into the plane, start the next item from zero
    cmp_cc = 0;
    const int mt = cmp_rem >> nt_shift, nt = cmp_rem & (ntiles - 1);
    float *dst = Mo + (size_t)cmp_plane * NT * Cout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int co = nt * BN + wn * WN + 16 * j + 4 * q;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = mt * BM + wm * WM + 16 * i + t;
        if (m < NT)   // plain stores: the non-temporal hint here was measured slower (179-181 -> 184-187 us per launch, round 4)
          *reinterpret_cast<float4 *>(dst + (size_t)m * Cout + co) =
              make_float4(acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]);
        acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    cmp_rem += nbl;
    while (cmp_rem >= per_plane) {
      cmp_rem -= per_plane;
      ++cmp_plane;
    }
  };
  if constexpr (PAIR) {
    W4_ITEM_SETUP()
    W4_ISSUE(0)
    W4_ADVANCE()
    W4_ISSUE(1)
    W4_ADVANCE()
    w4_wait_vmcnt<0>();
    W4_BARRIER();
    W4_FRAG(0, 0, wa0, xb0)
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    for (int kc = 0; kc < nchunks; kc += 2) {
      const int sa = (kc & 2), sb = sa + 1, sc = sa ^ 2, sd = sc + 1;   // stages of this pair / of the next
      const bool more = kc + 2 < nchunks;
      if (more && early) {
        W4_ISSUE(sc)
        W4_ADVANCE()
        W4_ISSUE(sd)
        W4_ADVANCE()
      }
      W4_MFMA_E(wa0, xb0, x)
      W4_FRAG(sa, 1, wa1, xb1)
      __builtin_amdgcn_sched_barrier(0);
      W4_MFMA_E(wa0, xb0, y)
      W4_MFMA_E(wa0, xb0, z)
      W4_MFMA_E(wa0, xb0, w)
      if (more && !early) {
        W4_ISSUE(sc)
        W4_ADVANCE()
        W4_ISSUE(sd)
        W4_ADVANCE()
      }
      W4_FRAG(sb, 0, wa0, xb0)   // the pair's second chunk landed before the previous barrier
      __builtin_amdgcn_sched_barrier(0);
      W4_MFMA_E(wa1, xb1, x)
      W4_MFMA_E(wa1, xb1, y)
      W4_MFMA_E(wa1, xb1, z)
      W4_MFMA_E(wa1, xb1, w)
      __builtin_amdgcn_sched_barrier(0);
      W4_MFMA_E(wa0, xb0, x)
      W4_FRAG(sb, 1, wa1, xb1)
      __builtin_amdgcn_sched_barrier(0);
      W4_MFMA_E(wa0, xb0, y)
      W4_MFMA_E(wa0, xb0, z)
      W4_MFMA_E(wa0, xb0, w)
      if (more) {
        w4_wait_vmcnt<0>();                  // the next pair has landed (issued ~1.5 chunks ago)
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's reads of the pair's stages are done
        W4_BARRIER();
        W4_FRAG(sc, 0, wa0, xb0)
      }
      __builtin_amdgcn_sched_barrier(0);
      W4_MFMA_E(wa1, xb1, x)
      W4_MFMA_E(wa1, xb1, y)
      W4_MFMA_E(wa1, xb1, z)
      W4_MFMA_E(wa1, xb1, w)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_sched_barrier(0);
      cmp_cc += 2;
      if (cmp_cc == cpt) epilogue();
    }
    return;
  }
  W4_ITEM_SETUP()
  W4_ISSUE(0)
  W4_ADVANCE()
  if (nchunks > 1) {
    W4_ISSUE(1)
    W4_ADVANCE()
    w4_wait_vmcnt<NPASS>();
  } else {
    w4_wait_vmcnt<0>();
  }
  W4_BARRIER();
  W4_FRAG(0, 0, wa0, xb0)
  int s_cur = 0, s_nxt = 1, s_nn = 2;  // ring positions of chunks kc, kc+1, kc+2
  // The second-dispatched half of the block loses the issue arbitration against its SIMD partner (priority, then age:
  // MI355X_MICROARCH.md, "Two waves per SIMD", item 4): one static priority bump for it, no per-phase flips.
  // Measured alone: layer4 412 -> 402 us, 28-crop layer4 533 -> 528 us, the short-K shapes unchanged; in the
  // three-stream pipeline neutral (160.9-161.2 frames/s either way, same-box A/B).
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  for (int kc = 0; kc < nchunks; ++kc) {
    if (kc + 2 < nchunks && early) W4_ISSUE(s_nn)
    W4_MFMA_E(wa0, xb0, x)
    W4_FRAG(s_cur, 1, wa1, xb1)
    __builtin_amdgcn_sched_barrier(0);
    W4_MFMA_E(wa0, xb0, y)
    W4_MFMA_E(wa0, xb0, z)
    W4_MFMA_E(wa0, xb0, w)
    if (kc + 2 < nchunks && !early) W4_ISSUE(s_nn)
    if (kc + 2 < nchunks) W4_ADVANCE()
    // chunk kc+1 has landed when at most the NPASS DMAs of chunk kc+2 are outstanding.  The stores of an item's
    // epilogue (below) are counted by vmcnt as well; they only make this wait conservative (loads retire in order
    // among themselves, so "at most NPASS outstanding" always implies chunk kc+1's loads are done).
    if (kc + 2 < nchunks)
      w4_wait_vmcnt<NPASS>();
    else
      w4_wait_vmcnt<0>();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the h=1 fragment reads
    W4_BARRIER();
    if (kc + 1 < nchunks) W4_FRAG(s_nxt, 0, wa0, xb0)
    __builtin_amdgcn_sched_barrier(0);
    W4_MFMA_E(wa1, xb1, x)
    W4_MFMA_E(wa1, xb1, y)
    W4_MFMA_E(wa1, xb1, z)
    W4_MFMA_E(wa1, xb1, w)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    if (++cmp_cc == cpt) epilogue();
    const int tmp = s_cur;
    s_cur = s_nxt;
    s_nxt = s_nn;
    s_nn = tmp;
  }
#undef W4_ITEM_SETUP
#undef W4_ISSUE
#undef W4_ADVANCE
#undef W4_FRAG
#undef W4_MFMA_E
}

template <int BM, int BN, bool PAIR>
static int launch_wino4_gemm_t(const float *V, const float *U, float *Mo, int NT, int Cin, int Cout, int planes, int nblocks,
                               hipStream_t st) {
  const int mtiles = (NT + BM - 1) / BM, ntiles = Cout / BN;
  int nt_shift = 0;
  while ((1 << nt_shift) < ntiles) ++nt_shift;
  UOC_REQUIRE((1 << nt_shift) == ntiles, "winograd F(4x4): Cout / %d = %d is not a power of two", BN, ntiles);
  constexpr int NSTAGE = PAIR ? 4 : 3;
  const size_t lds = (size_t)NSTAGE * (BM + BN) * W4BK * sizeof(float);
  static_assert((size_t)NSTAGE * (BM + BN) * W4BK * sizeof(float) <= 160 * 1024, "LDS ring too large");
  UOC_REQUIRE(!PAIR || (Cin / W4BK) % 2 == 0, "winograd F(4x4): the pair loop needs an even number of 32-channel chunks");
  static DeviceOnce attr_set;
  if (!attr_set.done()) {
    UOC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&wino4_gemm_kernel<BM, BN, PAIR>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.mark();
  }
  hipLaunchKernelGGL((wino4_gemm_kernel<BM, BN, PAIR>), dim3(nblocks), dim3(512), lds, st, V, U, Mo, NT, Cin, Cout, planes, mtiles,
                     nt_shift);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

// Tile choice (static cost model, never timing: every candidate sums each output in the same cin order, so the result
// does not depend on it): steps of the slowest block x tile area, mild preference for large tiles (fewer operand
// bytes per MFMA).  One block per CU (the 110 KB LDS ring), grid = 8 XCDs x 32.
static const int kW4Bm[4] = {96, 128, 160, 192};
static void pick_wino4_tile(int NT, int Cout, int planes, int &bm, int &bn, int &nblocks) {
  const int ncu = device_num_cu() > 0 ? device_num_cu() : 256;
  const int per_xcd = ncu / 8 > 0 ? ncu / 8 : 1;
  double best = -1;
  bm = 160;
  bn = Cout % 128 == 0 ? 128 : 64;
  for (int n = 0; n < 2; ++n) {
    const int BN = n ? 64 : 128;
    if (Cout % BN) continue;
    for (int i = 0; i < 4; ++i) {
      const int BM = kW4Bm[i];
      const long items = (long)planes * ((NT + BM - 1) / BM) * (Cout / BN);
      const long S = (items + 7) / 8;
      const long steps = (S + per_xcd - 1) / per_xcd;
      const double cost = (double)steps * BM * BN * (1.0 + 0.05 * (192.0 / BM - 1.0)) * (BN == 64 ? 1.06 : 1.0);
      if (best < 0 || cost < best) {
        best = cost;
        bm = BM;
        bn = BN;
      }
    }
  }
  const long items = (long)planes * ((NT + bm - 1) / bm) * (Cout / bn);
  const long S = (items + 7) / 8;
  nblocks = 8 * (int)(S < per_xcd ? S : per_xcd);
}

static int launch_wino4_gemm(const float *V, const float *U, float *Mo, int NT, int Cin, int Cout, int planes, hipStream_t st) {
  int bm, bn, nblocks;
  pick_wino4_tile(NT, Cout, planes, bm, bn, nblocks);
  static int env_bm = 0, env_bn = 0, env_seen = -1;   // dev knob UOC_WINO4_TILE="BMxBN", cached (uoc_reload_env re-reads)
  if (env_seen != g_env_epoch.load(std::memory_order_acquire)) {
    env_seen = g_env_epoch.load(std::memory_order_acquire);
    env_bm = env_bn = 0;
    if (const char *e = getenv("UOC_WINO4_TILE")) {
      int a = 0, b = 0;
      if (sscanf(e, "%dx%d", &a, &b) == 2 && (b == 64 || b == 128)) env_bm = a, env_bn = b;
    }
  }
  if (env_bm > 0) {
    const int a = env_bm, b = env_bn;
    if (Cout % b == 0) {
      bm = a;
      bn = b;
      const int ncu = device_num_cu() > 0 ? device_num_cu() : 256;
      const long items = (long)planes * ((NT + bm - 1) / bm) * (Cout / bn), S = (items + 7) / 8;
      nblocks = 8 * (int)(S < ncu / 8 ? S : ncu / 8);
    }
  }
  // UOC_W4_PAIR (A/B): 0 = the 3-stage ring with one barrier per chunk (round 3) everywhere.  Measured per launch shape,
  // same box: layer4 383 -> 377 / 569 -> 551 / 533 -> 516 us, layer3 117.4 -> 115.8 / 165 -> 161 / 150.5 -> 147 us, layer2
  // equal, the 2-chunk items of layer1 45.0 -> 45.6 us (worse: they keep the single-chunk loop); class average 154 -> 151.5 us
  static EnvInt pair_env("UOC_W4_PAIR", 1);
  const int cpt = Cin / W4BK;
  const bool pair = pair_env.get() != 0 && cpt % 2 == 0 && cpt >= 4;
#define W4_CASE(A, B)                                                                                                \
  if (bm == A && bn == B)                                                                                            \
    return pair ? launch_wino4_gemm_t<A, B, true>(V, U, Mo, NT, Cin, Cout, planes, nblocks, st)                       \
                : launch_wino4_gemm_t<A, B, false>(V, U, Mo, NT, Cin, Cout, planes, nblocks, st);
  W4_CASE(96, 128) W4_CASE(128, 128) W4_CASE(160, 128) W4_CASE(192, 128)
  W4_CASE(96, 64) W4_CASE(128, 64) W4_CASE(160, 64) W4_CASE(192, 64)
#undef W4_CASE
  set_error("winograd F(4x4): no GEMM tile %dx%d", bm, bn);
  return UOC_EINVAL;
}

// ---- host side ---------------------------------------------------------------------------------------------------
// Cout / 64 must be a power of two: the plane GEMM splits an item index into (m-tile, n-tile) with a shift (ResNet34: 128,
// 256, 512).  Other widths (192, 320, 384 ...) take the direct kernel.
bool wino4_channels_ok(int Cin, int Cout) {
  const int n64 = Cout / 64;
  return Cin % W4BK == 0 && Cout % 64 == 0 && n64 > 0 && (n64 & (n64 - 1)) == 0;
}

bool wino4_eligible(const ConvParams &p) {
  return !p.stem && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == p.dil && wino4_channels_ok(p.Cin, p.Cout) &&
         p.Ho == p.H && p.Wo == p.W;
}

size_t wino4_ws_floats(int G, int B, int H, int W, int d, int Cin, int Cout) {
  const Wino4Geom geo = make_geom4(B, H, W, d);
  return (size_t)G * 36 * geo.NT * 2 * (size_t)(Cin > Cout ? Cin : Cout);   // two halves: V planes, M planes (chains of layers use them as such)
}

int launch_wino4_weights(const float *w, float *U, int G, int Cout, int Cin, hipStream_t st) {
  const long total = (long)G * Cout * Cin;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wino4_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w, U, G, Cout, Cin);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

// Grid of the two elementwise kernels: every thread owns one (tile, channel quad) item at a time; the grid is capped at
// the blocks that are resident at once (2 per CU at ~200 VGPRs) so that all blocks walk equally long item ranges and
// finish together instead of leaving a half-empty last round (UOC_W4_TGRID = blocks per CU, 0 = one block per 256 items).
static long wino4_elem_blocks(long items) {
  static int per_cu = -1;
  if (per_cu < 0) {
    const char *e = getenv("UOC_W4_TGRID");
    per_cu = e ? atoi(e) : 0;
  }
  long blocks = (items + 255) / 256;
  const long cap = per_cu > 0 ? (long)per_cu * (device_num_cu() > 0 ? device_num_cu() : 256) : 16384;
  return blocks < cap ? blocks : cap;
}

static int launch_wino4_slice(const ConvParams &p0, int Bg, int b0, const float *U, float *ws, hipStream_t st);

int launch_wino4_conv(const ConvParams &p, const float *U, float *ws, hipStream_t st) {
  UOC_REQUIRE(wino4_eligible(p), "winograd F(4x4): layer not eligible");
  UOC_REQUIRE(U && ws && p.in && p.out, "winograd F(4x4): null tensor/weight/workspace pointer");
  const int planes = 36 * p.G;
  UOC_REQUIRE((size_t)planes * p.Cout * p.Cin * 4 < (1ull << 32), "winograd F(4x4): weight planes exceed 4 GB");
  // The plane GEMM addresses V and M with 32-bit buffer offsets: a batch whose frequency planes exceed 4 GB is run as
  // several launches over slices of the batch (same tiles, same arithmetic: a tile never spans two images).
  const size_t per_image = (size_t)planes * make_geom4(1, p.H, p.W, p.dil).NT * (p.Cin > p.Cout ? p.Cin : p.Cout) * 4;
  UOC_REQUIRE(per_image < (1ull << 32), "winograd F(4x4): one image's frequency planes exceed 4 GB");
  static EnvInt limit_mb("UOC_WINO4_MAX_MB", 0);   // dev / tests: a smaller limit, to exercise the split on small batches
  const size_t limit = limit_mb.get() > 0 && ((size_t)limit_mb.get() << 20) > per_image ? (size_t)limit_mb.get() << 20 : (1ull << 32) - 1;
  const int bmax = (int)(limit / per_image);
  if (p.B > bmax) {
    for (int b0 = 0; b0 < p.B; b0 += bmax) {
      ConvParams q = p;
      q.B = p.B - b0 < bmax ? p.B - b0 : bmax;
      if (int rc = launch_wino4_slice(q, p.B, b0, U, ws, st)) return rc;
    }
    return UOC_OK;
  }
  return launch_wino4_slice(p, p.B, 0, U, ws, st);
}

// ---- the three stages of one layer, and the fused stage between two layers --------------------------------------------
static int w4_vec() {   // channels per thread of the two elementwise kernels (UOC_W4_VEC, A/B): 4 (float4) moves the most bytes per instruction
  static EnvInt e("UOC_W4_VEC", 4);
  const int v = e.get();
  return v == 1 || v == 2 ? v : 4;
}

static int w4_stage_input(const ConvParams &p, const Wino4Geom &geo, float *V, hipStream_t st) {
  const double Mpix = (double)p.B * p.H * p.W;
  const ProfTag tag = {{geo.NT, p.Cin, p.Cout, p.dil}};
  const int vec = w4_vec();
  ProfScope prof(KC_WINO4_INPUT, st, 0.0, 4.0 * p.G * (Mpix * p.Cin + 36.0 * geo.NT * p.Cin), tag);
  const long blocks = wino4_elem_blocks((long)p.G * geo.NT * (p.Cin / vec));
  if (vec == 1)
    hipLaunchKernelGGL(wino4_input_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, p.in, V, geo, p.G, p.Cin);
  else if (vec == 2)
    hipLaunchKernelGGL(wino4_input_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, p.in, V, geo, p.G, p.Cin);
  else
    hipLaunchKernelGGL(wino4_input_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, p.in, V, geo, p.G, p.Cin);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

static int w4_stage_gemm(const ConvParams &p, const Wino4Geom &geo, const float *U, float *V, float *Mw, hipStream_t st) {
  const int planes = 36 * p.G;
  const double Mpix = (double)p.B * p.H * p.W;
  const ProfTag tag = {{geo.NT, p.Cin, p.Cout, p.dil}};
  static EnvInt gemm_env("UOC_WINO4_GEMM", 2);   // 2 = the persistent plane-GEMM kernel; 1 = one direct 1x1 "convolution" over 36*G groups (A/B, dev)
  const int gemm_mode = gemm_env.get() == 1 ? 1 : 2;
  // algorithmic flops = the direct 3x3 convolution's; the matrix pipe executes 36/144 of them
  // (+ the padding of partial tiles); bytes: V and U read once, M written once
  const double gflops = 2.0 * Mpix * p.Cout * p.Cin * 9.0 * p.G;
  const double gbytes = 4.0 * planes * ((double)geo.NT * p.Cin + (double)p.Cout * p.Cin + (double)geo.NT * p.Cout);
  if (gemm_mode == 1) {
    ConvParams q;
    q.in = V;
    q.w = U;
    q.bias = nullptr;
    q.res = nullptr;
    q.out = Mw;
    q.G = planes;
    q.B = 1;
    q.H = 1;
    q.W = geo.NT;
    q.Cin = p.Cin;
    q.Ho = 1;
    q.Wo = geo.NT;
    q.Cout = p.Cout;
    q.KH = q.KW = 1;
    q.stride = 1;
    q.dil = 1;
    q.pad = 0;
    q.relu = 0;
    q.stem = 0;
    q.tune = p.tune;
    q.prof_kc = KC_WINO4_GEMM;
    q.prof_flops = gflops;
    q.prof_tag[0] = geo.NT, q.prof_tag[1] = p.Cin, q.prof_tag[2] = p.Cout, q.prof_tag[3] = p.dil;
    return launch_conv(q, st);
  }
  ProfScope prof(KC_WINO4_GEMM, st, gflops, gbytes, tag);
  return launch_wino4_gemm(V, U, Mw, geo.NT, p.Cin, p.Cout, planes, st);
}

static int w4_stage_output(const ConvParams &p, const Wino4Geom &geo, const float *Mw, hipStream_t st) {
  const double Mpix = (double)p.B * p.H * p.W;
  const ProfTag tag = {{geo.NT, p.Cin, p.Cout, p.dil}};
  const int vec = w4_vec();
  ProfScope prof(KC_WINO4_OUTPUT, st, 0.0, 4.0 * p.G * (36.0 * geo.NT * p.Cout + Mpix * p.Cout * (p.res ? 2 : 1)), tag);
  const long blocks = wino4_elem_blocks((long)p.G * geo.NT * (p.Cout / vec));
  if (vec == 1)
    hipLaunchKernelGGL(wino4_output_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, Mw, p.bias, p.res, p.out, geo, p.G, p.Cout, p.relu);
  else if (vec == 2)
    hipLaunchKernelGGL(wino4_output_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, Mw, p.bias, p.res, p.out, geo, p.G, p.Cout, p.relu);
  else
    hipLaunchKernelGGL(wino4_output_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, Mw, p.bias, p.res, p.out, geo, p.G, p.Cout, p.relu);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

// images b0 .. b0 + p.B - 1 of a batch of Bg (activation tensors [g][Bg][H][W][C])
static int launch_wino4_slice(const ConvParams &p0, int Bg, int b0, const float *U, float *ws, hipStream_t st) {
  ConvParams p = p0;
  const size_t img_in = (size_t)p.H * p.W * p.Cin, img_out = (size_t)p.H * p.W * p.Cout;
  p.in += (size_t)b0 * img_in;
  p.out += (size_t)b0 * img_out;
  if (p.res) p.res += (size_t)b0 * img_out;
  Wino4Geom geo = make_geom4(p.B, p.H, p.W, p.dil);
  geo.Bg = Bg;
  float *V = ws, *Mw = ws + (size_t)36 * p.G * geo.NT * p.Cin;
  if (int rc = w4_stage_input(p, geo, V, st)) return rc;
  if (int rc = w4_stage_gemm(p, geo, U, V, Mw, st)) return rc;
  return w4_stage_output(p, geo, Mw, st);
}

// ---- chains of layers (csrc/net.hip): V and M planes in caller-owned halves of the scratch; between two layers of one
// geometry the fused kernel replaces output transform + input transform -------------------------------------------------
// channels per block of the fused kernel: the largest slice whose LDS image (with its zero frame) fits 60 KB, so that two
// blocks share a CU; 0 = the phase image is too large (stage-1 layer2: 62 x 82 pixels), the caller runs the two kernels
static int w4_mid_vec() {   // channels per thread of the fused kernel: 2 (float2: 167 VGPRs, three waves per SIMD) or 4 (280 VGPRs: one wave)
  static EnvInt e("UOC_W4_MID_VEC", 2);
  return e.get() == 4 ? 4 : 2;
}

static int w4_mid_slice(const Wino4Geom &geo, int C) {
  // OFF by default: measured slower end to end (profiles/r04_ab_fused_transforms.md) — the fused kernel needs ~50 KB of LDS
  // per block and so cannot share a CU with another stream's plane GEMM, which the LDS-free separate kernels do.
  static EnvInt on("UOC_WINO4_FUSE", 0);
  if (on.get() == 0) return 0;
  const int vec = w4_mid_vec(), max_threads = vec == 4 ? 256 : 512;
  const long px = (long)(4 * geo.TH + 2) * (4 * geo.TW + 2), tiles = (long)geo.TH * geo.TW;
  // one thread per (tile, VEC channels).  Measured (profiles/r04_ab_fused_transforms.md): the fused kernel only beats the
  // two separate ones when a block's rows are whole 128-byte lines (>= 32 channels); the phase images of stage-1 layer3
  // (30 x 40) and of the crops' layer2 (28 x 28) only fit LDS with 8 channels and lose.  Largest slice with at most 256
  // items (one pass of a 4-wave block) and three blocks per CU (53 KB of LDS each); else the smallest eligible one.
  static EnvInt min_cs("UOC_W4_MID_MIN_CS", 32);
  int best = 0;
  for (int cs = 8; cs <= 128; cs <<= 1) {
    if (C % cs || px * cs * 4 > 53 * 1024) break;
    if (cs < min_cs.get()) continue;
    const long items = tiles * (cs / vec);
    if (items <= 256 || (best == 0 && items <= max_threads)) best = cs;
  }
  return best;
}

bool wino4_chain_ok(const ConvParams &p) {   // no batch split needed (32-bit plane offsets) and eligible
  if (!wino4_eligible(p)) return false;
  const Wino4Geom geo = make_geom4(p.B, p.H, p.W, p.dil);
  return (size_t)36 * p.G * geo.NT * (p.Cin > p.Cout ? p.Cin : p.Cout) * 4 < (1ull << 32) &&
         (size_t)36 * p.G * p.Cout * p.Cin * 4 < (1ull << 32);
}

bool wino4_can_fuse(const ConvParams &prev, const ConvParams &next) {
  return prev.G == next.G && prev.B == next.B && prev.H == next.H && prev.W == next.W && prev.dil == next.dil &&
         prev.Cout == next.Cin && w4_mid_slice(make_geom4(prev.B, prev.H, prev.W, prev.dil), prev.Cout) > 0;
}

int wino4_chain_input(const ConvParams &p, float *V, hipStream_t st) {
  return w4_stage_input(p, make_geom4(p.B, p.H, p.W, p.dil), V, st);
}
int wino4_chain_gemm(const ConvParams &p, const float *U, float *V, float *Mw, hipStream_t st) {
  return w4_stage_gemm(p, make_geom4(p.B, p.H, p.W, p.dil), U, V, Mw, st);
}
int wino4_chain_output(const ConvParams &p, const float *Mw, hipStream_t st) {
  return w4_stage_output(p, make_geom4(p.B, p.H, p.W, p.dil), Mw, st);
}

// plane GEMMs + output transform of a small-K layer in one kernel (wino4_small_kernel): V planes -> p.out
bool wino4_small_ok(const ConvParams &p) {
  // OFF by default: built, bit-identical, measured SLOWER (round 4, same box): 135 us against 45 + 36 us (plane GEMM +
  // output transform) on stage-1 layer1, 137 against 40 + 20 us on layer2; 158.5 vs 166.9 frames/s.  Keeping 36 plane
  // accumulators per lane limits a wave to a 16 x 16 tile: 512 operand bytes per MFMA from L2 (the 160 x 128 block of the
  // plane GEMM needs 57), one plane of prefetch (0.2 us of MFMAs) cannot cover the L2 latency, and the register budget
  // allows neither a larger tile nor a deeper prefetch.
  static EnvInt on("UOC_WINO4_SMALL", 0);
  return on.get() != 0 && p.Cin == p.Cout && (p.Cin == 64 || p.Cin == 128);
}

int wino4_chain_gemm_out(const ConvParams &p, const float *U, const float *V, hipStream_t st) {
  const Wino4Geom geo = make_geom4(p.B, p.H, p.W, p.dil);
  const int grid = p.G * ((geo.NT + 15) / 16);
  const double Mpix = (double)p.B * p.H * p.W;
  const ProfTag tag = {{geo.NT, p.Cin, p.Cout, p.dil}};
  ProfScope prof(KC_WINO4_SMALL, st, 2.0 * Mpix * p.Cout * p.Cin * 9.0 * p.G,
                 4.0 * p.G * (36.0 * geo.NT * p.Cin + 36.0 * p.Cout * p.Cin + Mpix * p.Cout * (p.res ? 2 : 1)), tag);
  if (p.Cin == 64)
    hipLaunchKernelGGL(wino4_small_kernel<64>, dim3(grid), dim3(256), 0, st, V, U, p.bias, p.res, p.out, geo, p.G, p.relu);
  else
    hipLaunchKernelGGL(wino4_small_kernel<128>, dim3(grid), dim3(512), 0, st, V, U, p.bias, p.res, p.out, geo, p.G, p.relu);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

// prev's output transform (bias, residual, ReLU; its NHWC tensor prev.out is written only if write_y) + the next layer's
// input transform: Mw (prev's M planes) -> V (the next layer's V planes)
int wino4_chain_mid(const ConvParams &prev, bool write_y, const float *Mw, float *V, hipStream_t st) {
  const Wino4Geom geo = make_geom4(prev.B, prev.H, prev.W, prev.dil);
  const int C = prev.Cout, cs = w4_mid_slice(geo, C);
  UOC_REQUIRE(cs > 0, "winograd F(4x4): layers cannot be fused");
  const int sib = cs < 32 ? 32 / cs : 1;
  const int units = prev.G * prev.B * prev.dil * prev.dil * (C / cs);
  const int grid = (units + 8 * sib - 1) / (8 * sib) * (8 * sib);
  const int vec = w4_mid_vec();
  const int items = geo.TH * geo.TW * (cs / vec);
  int threads = (items + 63) / 64 * 64;
  if (threads > (vec == 4 ? 256 : 512)) threads = vec == 4 ? 256 : 512;
  const size_t lds = (size_t)(4 * geo.TH + 2) * (4 * geo.TW + 2) * cs * 4;
  const double Mpix = (double)prev.B * prev.H * prev.W;
  const ProfTag tag = {{geo.NT, C, C, prev.dil}};
  ProfScope prof(KC_WINO4_MID, st, 0.0,
                 4.0 * prev.G * (2.0 * 36.0 * geo.NT * C + Mpix * C * ((prev.res ? 1 : 0) + (write_y ? 1 : 0))), tag);
  if (vec == 4)
    hipLaunchKernelGGL(wino4_mid_kernel<4>, dim3((unsigned)grid), dim3(threads), lds, st, Mw, prev.bias, prev.res,
                       write_y ? prev.out : nullptr, V, geo, prev.G, C, prev.relu, cs, sib, units);
  else
    hipLaunchKernelGGL(wino4_mid_kernel<2>, dim3((unsigned)grid), dim3(threads), lds, st, Mw, prev.bias, prev.res,
                       write_y ? prev.out : nullptr, V, geo, prev.G, C, prev.relu, cs, sib, units);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

}  // namespace uoc
