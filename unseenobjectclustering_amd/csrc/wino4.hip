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
#ifndef W4_OUT_MIN_BLOCKS
#define W4_OUT_MIN_BLOCKS 2   // blocks per CU the output transform is compiled for: two waves per SIMD = at most 256 registers per lane (8 of them spill)
#endif

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
#ifndef W4_IN_XCD
#define W4_IN_XCD 1
#endif
  // XCD-aware block order (the grid is a multiple of 8): hardware block b runs on XCD b % 8; virtual block = (b % 8) * (grid / 8)
  // + b / 8 gives every XCD one CONTIGUOUS eighth of the tiles, so the 6x6 patches of neighbouring tiles — which share two of
  // their six rows / columns — are fetched into ONE L2 instead of eight
  const long vblock = W4_IN_XCD ? (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : (long)blockIdx.x;
  for (long idx = vblock * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(idx % CV);
    const int tau = (int)((idx / CV) % geo.NT);
    const int g = (int)(idx / ((long)CV * geo.NT));
    wino4_input_body<VEC>(in, V, geo, C, g, tau, cv);
  }
}

template <int VEC>
__global__ __launch_bounds__(256, W4_OUT_MIN_BLOCKS) void wino4_output_kernel(const float *__restrict__ M, const float *__restrict__ bias,
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

// PAIR (round 4): a 4-stage ring and ONE barrier per TWO K-chunks.  The barrier costs 5-9 % of the kernel (15 % on 4-chunk
// items: timing ablation, profiles/r04_pmc_mfma.md); a pair of chunks is fetched two chunks ahead into the two stages the
// previous pair has just left, so the barrier at the end of a pair covers both hazards (the DMA'd rows of the next pair are
// visible to every wave; every wave has finished reading the stages the pair after next will overwrite).  The items of a
// block always hold an even number of chunks (Cin / 32 = 2, 4, 8, 16), so a pair never straddles two items.  Same MFMA
// order per accumulator as the single-chunk loop: bit-identical results.
//
// NW (round 5): waves per block.  8 = two waves per SIMD as 2 (m) x 4 (n), wave tile (BM/2) x (BN/4) — 0.175 fragment reads
// per MFMA at 160 x 128.  4 = ONE wave per SIMD as 2 x 2 on the SAME block tile and LDS ring, wave tile (BM/2) x (BN/2) —
// 80 x 64: (5 + 4) ds_read_b128 per 80 MFMAs = 0.1125 per MFMA (each read's write-back costs ~24 cycles of the fp32 lanes
// the MFMA runs on, profiles/r02_mfma_microbenchmarks.md) and half as many waves meeting at the barrier.  Same cin order
// per accumulator: bit-identical.
template <int BM, int BN, bool PAIR, int NW>
__global__ __launch_bounds__(NW * 64) void wino4_gemm_kernel(const float *__restrict__ V, const float *__restrict__ U,
                                                             float *__restrict__ Mo, int NT, int Cin, int Cout, int planes,
                                                             int mtiles, int nt_shift) {
  static_assert(NW == 8 || NW == 4, "waves per block");
  constexpr int WAVES_N = NW / 2;   // 2 (m) x WAVES_N (n)
  constexpr int WM = BM / 2, WN = BN / WAVES_N;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int R = BM + BN;
  constexpr int RPP = NW * 8;  // rows per DMA pass: every wave moves 8 rows (1 KiB) per instruction
  constexpr int NPA = (BM + RPP - 1) / RPP, NPW = (BN + RPP - 1) / RPP, NPASS = NPA + NPW;
  constexpr int STAGE = R * W4BK;
  static_assert(WM % 16 == 0 && WN % 16 == 0 && R % 8 == 0 && NPASS <= 12, "tile shape");

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
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
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
  const bool early = NW == 4 || wave < 4;   // NW = 8: waves w and w+4 share a SIMD and issue their DMA bursts at different points
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
    if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
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
  if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
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

template <int BM, int BN, bool PAIR, int NW>
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
    UOC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&wino4_gemm_kernel<BM, BN, PAIR, NW>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.mark();
  }
  hipLaunchKernelGGL((wino4_gemm_kernel<BM, BN, PAIR, NW>), dim3(nblocks), dim3(NW * 64), lds, st, V, U, Mo, NT, Cin, Cout, planes,
                     mtiles, nt_shift);
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
  // Pair loop (4-stage ring, one barrier per two K-chunks) against the 3-stage ring with one barrier per chunk, per launch shape,
  // same box: layer4 383 -> 377 / 569 -> 551 / 533 -> 516 us, layer3 117.4 -> 115.8 / 165 -> 161 / 150.5 -> 147 us, layer2
  // equal, the 2-chunk items of layer1 45.0 -> 45.6 us (worse: they keep the single-chunk loop); class average 154 -> 151.5 us
  const int cpt = Cin / W4BK;
  const bool pair = cpt % 2 == 0 && cpt >= 4;
  // 8 waves per block (two per SIMD: the younger wave's MFMAs cover the older one's vmcnt / barrier wait; the one-wave-per-SIMD
  // variant NW = 4 measured 15-25 % slower in round 5, HISTORY.md)
#define W4_CASE(A, B)                                                                                                \
  if (bm == A && bn == B)                                                                                            \
    return pair ? launch_wino4_gemm_t<A, B, true, 8>(V, U, Mo, NT, Cin, Cout, planes, nblocks, st)                    \
                : launch_wino4_gemm_t<A, B, false, 8>(V, U, Mo, NT, Cin, Cout, planes, nblocks, st);
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
  // V planes + M planes; the split-precision experiment stores V as three bf16 planes (1.5x) -> 2.5 units + alignment slack
  return (size_t)G * 36 * geo.NT * 5 / 2 * (size_t)(Cin > Cout ? Cin : Cout) + 64;
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
// finish together instead of leaving a half-empty last round.
long wino4_elem_blocks(long items) {
  long blocks = (items + 255) / 256;
  const long cap = 16384;
  blocks = blocks < cap ? blocks : cap;
  return (blocks + 7) / 8 * 8;     // a multiple of 8: the input transform deals its blocks to the XCDs in contiguous eighths
}

static int launch_wino4_slice(const ConvParams &p0, int Bg, int b0, const float *U, float *ws, hipStream_t st);

int launch_wino4_conv(const ConvParams &p, const float *U, float *ws, hipStream_t st, const unsigned short *U3) {
  UOC_REQUIRE(wino4_eligible(p), "winograd F(4x4): layer not eligible");
  UOC_REQUIRE(U && ws && p.in && p.out, "winograd F(4x4): null tensor/weight/workspace pointer");
  const int planes = 36 * p.G;
  UOC_REQUIRE((size_t)planes * p.Cout * p.Cin * 4 < (1ull << 32), "winograd F(4x4): weight planes exceed 4 GB");
  // The plane GEMM addresses V and M with 32-bit buffer offsets: a batch whose frequency planes exceed 4 GB is run as
  // several launches over slices of the batch (same tiles, same arithmetic: a tile never spans two images).
  const size_t per_image = (size_t)planes * make_geom4(1, p.H, p.W, p.dil).NT * (p.Cin > p.Cout ? p.Cin : p.Cout) * (U3 ? 6 : 4);
  UOC_REQUIRE(per_image < (1ull << 32), "winograd F(4x4): one image's frequency planes exceed 4 GB");
  static EnvInt limit_mb("UOC_SPLIT_MAX_MB", 0);   // tests: a smaller limit, to exercise the split on small batches (same results)
  const size_t limit = limit_mb.get() > 0 && ((size_t)limit_mb.get() << 20) > per_image ? (size_t)limit_mb.get() << 20 : (1ull << 32) - 1;
  const int bmax = (int)(limit / per_image);
  if (p.B > bmax) {
    for (int b0 = 0; b0 < p.B; b0 += bmax) {
      ConvParams q = p;
      q.B = p.B - b0 < bmax ? p.B - b0 : bmax;
      if (int rc = U3 ? launch_wino4_slice_split(q, p.B, b0, U3, ws, st) : launch_wino4_slice(q, p.B, b0, U, ws, st)) return rc;
    }
    return UOC_OK;
  }
  return U3 ? launch_wino4_slice_split(p, p.B, 0, U3, ws, st) : launch_wino4_slice(p, p.B, 0, U, ws, st);
}

// ---- the three stages of one layer ---------------------------------------------------------------------------------
static int w4_vec() { return 4; }   // channels per thread of the two elementwise kernels: float4 moves the most bytes per instruction

static int w4_stage_input(const ConvParams &p, const Wino4Geom &geo, float *V, hipStream_t st) {
  const double Mpix = (double)p.B * p.H * p.W;
  const ProfTag tag = {{geo.NT, p.Cin, p.Cout, p.dil}};
  const int vec = w4_vec();
  ProfScope prof(KC_WINO4_INPUT, st, 0.0, 4.0 * p.G * (Mpix * p.Cin + 36.0 * geo.NT * p.Cin), tag);
  const long blocks = wino4_elem_blocks((long)p.G * geo.NT * (p.Cin / vec));
    hipLaunchKernelGGL(wino4_input_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, p.in, V, geo, p.G, p.Cin);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

static int w4_stage_gemm(const ConvParams &p, const Wino4Geom &geo, const float *U, float *V, float *Mw, hipStream_t st) {
  const int planes = 36 * p.G;
  const double Mpix = (double)p.B * p.H * p.W;
  const ProfTag tag = {{geo.NT, p.Cin, p.Cout, p.dil}};
  // algorithmic flops = the direct 3x3 convolution's; the matrix pipe executes 36/144 of them
  // (+ the padding of partial tiles); bytes: V and U read once, M written once
  const double gflops = 2.0 * Mpix * p.Cout * p.Cin * 9.0 * p.G;
  const double gbytes = 4.0 * planes * ((double)geo.NT * p.Cin + (double)p.Cout * p.Cin + (double)geo.NT * p.Cout);
  ProfScope prof(KC_WINO4_GEMM, st, gflops, gbytes, tag);
  return launch_wino4_gemm(V, U, Mw, geo.NT, p.Cin, p.Cout, planes, st);
}

int w4_stage_output(const ConvParams &p, const Wino4Geom &geo, const float *Mw, hipStream_t st) {
  const double Mpix = (double)p.B * p.H * p.W;
  const ProfTag tag = {{geo.NT, p.Cin, p.Cout, p.dil}};
  const int vec = w4_vec();
  ProfScope prof(KC_WINO4_OUTPUT, st, 0.0, 4.0 * p.G * (36.0 * geo.NT * p.Cout + Mpix * p.Cout * (p.res ? 2 : 1)), tag);
  const long blocks = wino4_elem_blocks((long)p.G * geo.NT * (p.Cout / vec));
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

}  // namespace uoc
