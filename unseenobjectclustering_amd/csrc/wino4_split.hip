// Split-precision plane GEMM of the Winograd F(4x4,3x3) path — an EXPERIMENT reported under its own key and dtype
// ("bf16x3"), never the headline (the gate was profiles/r06_lds_dma_l2.md: LDS-DMA fills LDS at 31 TB/s from
// L2, five times the HBM-stream figure round 3 had priced this kernel against).
//
// fp32 operand x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (3 x 8 = 24 significand bits: the
// split is exact up to the last rounding of l).  A product a*b then is the sum of nine bf16 x bf16 products, each EXACT in
// fp32; the six largest — hh, hm, mh, mm, hl, lh — carry everything above 2^-24 of the result (ml, lm, ll are <= 2^-24
// relative, the size of one fp32 rounding).  They run on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: 6 instructions of
// 16 cycles for a 16x16x32 block against 8 x 32 cycles of v_mfma_f32_16x16x4_f32 = 2.67x the fp32 matrix rate.
// scripts/wino_f4_error.py (round 3) emulated exactly this through the whole two-branch network: embeddings as close to an
// fp64 evaluation as with fp32 GEMMs (mean |error| 1.06e-8 vs 1.16e-8).
//
// Replaces, when uoc_net_set_split_precision(net, 1) is on, the fp32 plane GEMM of csrc/wino4.hip for the same layers
// (lib/networks/resnet.py:24-73,188-234: the 3x3 stride-1 convolutions): the input transform writes V as three bf16 planes
// [plane][part][tile][Cin], the transformed weights are split once into [plane][part][Cout][Cin], M and the output
// transform are unchanged (fp32).
#include "conv.h"
#include "dma.h"
#include "prof.h"
#include "wino4_math.h"

namespace uoc {

constexpr int W3BK = 32;   // channels per K-chunk = one v_mfma_f32_16x16x32_bf16 step = 64 bytes per row and part

// round-to-nearest-even fp32 -> bf16 (finite inputs; activations and weights are finite)
__host__ __device__ inline unsigned short bf16_rne(float x) {
  unsigned u;
#if defined(__HIP_DEVICE_COMPILE__)
  u = __float_as_uint(x);
#else
  memcpy(&u, &x, 4);
#endif
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__host__ __device__ inline float bf16_to_f32(unsigned short h) {
  const unsigned u = (unsigned)h << 16;
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(u);
#else
  float f;
  memcpy(&f, &u, 4);
  return f;
#endif
}
__device__ __forceinline__ void split3(float x, unsigned short &h, unsigned short &m, unsigned short &l) {
  h = bf16_rne(x);
  const float r1 = x - bf16_to_f32(h);     // exact: at most 17 significant bits
  m = bf16_rne(r1);
  const float r2 = r1 - bf16_to_f32(m);    // exact
  l = bf16_rne(r2);
}
__device__ __forceinline__ void split3x4(const float4 v, uint2 &h, uint2 &m, uint2 &l) {
  unsigned short hh[4], mm[4], ll[4];
  split3(v.x, hh[0], mm[0], ll[0]);
  split3(v.y, hh[1], mm[1], ll[1]);
  split3(v.z, hh[2], mm[2], ll[2]);
  split3(v.w, hh[3], mm[3], ll[3]);
  h = make_uint2((unsigned)hh[0] | ((unsigned)hh[1] << 16), (unsigned)hh[2] | ((unsigned)hh[3] << 16));
  m = make_uint2((unsigned)mm[0] | ((unsigned)mm[1] << 16), (unsigned)mm[2] | ((unsigned)mm[3] << 16));
  l = make_uint2((unsigned)ll[0] | ((unsigned)ll[1] << 16), (unsigned)ll[2] | ((unsigned)ll[3] << 16));
}

// Operand layout of the split GEMM, CHUNK-MAJOR: X3[(plane*3 + part)][Cin/32][rows][32] bf16 — the 32-channel slice of a block
// tile's rows is ONE contiguous piece (128 rows x 64 B = 8 KB), so every LDS-DMA instruction reads 1 KB of whole cache lines
// spread over all L2 channels.  (Row-major [rows][Cin] puts the 64-byte slices of a chunk Cin*2 bytes apart: half-used lines
// camping on a quarter of the channels — measured: the GEMM ran at 28 % of the bf16 matrix rate with that layout.)
// U3 from U[plane][cout][cin]; one thread = 4 consecutive cin
__global__ __launch_bounds__(256) void wino4_split_weights_kernel(const float *__restrict__ U, unsigned short *__restrict__ U3,
                                                                  long planes, int Cout, int Cin) {
  const long per_plane = (long)Cout * Cin, total4 = planes * per_plane / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const long e = 4 * i, pl = e / per_plane, off = e - pl * per_plane;
    const int co = (int)(off / Cin), ci = (int)(off - (long)co * Cin);
    uint2 h, m, l;
    split3x4(*reinterpret_cast<const float4 *>(U + e), h, m, l);
    unsigned short *dst = U3 + (pl * 3) * per_plane + ((long)(ci >> 5) * Cout + co) * 32 + (ci & 31);
    *reinterpret_cast<uint2 *>(dst) = h;
    *reinterpret_cast<uint2 *>(dst + per_plane) = m;
    *reinterpret_cast<uint2 *>(dst + 2 * per_plane) = l;
  }
}

// The input transform of csrc/wino4.hip (same arithmetic: wino4_input_tile) with the split store:
// V3[((g*36 + xi)*3 + part)][cin / 32][tile][32]; a thread = one tile x 4 channels, 8 threads = one tile's 64-byte slice of a
// chunk, a wave = 8 consecutive tiles of one chunk (512 contiguous bytes per store instruction)
__global__ __launch_bounds__(256) void wino4_input3_kernel(const float *__restrict__ in, unsigned short *__restrict__ V3,
                                                           Wino4Geom geo, int G, int C) {
  const int CV = C / 4;
  const long total = (long)G * geo.NT * CV;
  const unsigned grid = gridDim.x, per = grid >> 3;
  const unsigned vblock = per ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : blockIdx.x;   // XCD-aware order, as wino4_input_kernel
  for (long idx = (long)vblock * blockDim.x + threadIdx.x; idx < total; idx += (long)grid * blockDim.x) {
    const int c8 = (int)(idx & 7);
    const int tau = (int)((idx >> 3) % geo.NT);
    const int chunk = (int)(((idx >> 3) / geo.NT) % (CV >> 3));
    const int g = (int)(idx / ((long)CV * geo.NT));
    const int cv = chunk * 8 + c8;
    int b, oy, ox;
    wino4_decode(tau, geo, b, oy, ox);
    const float *src = in + (((size_t)g * geo.Bg + b) * geo.H * geo.W) * C + 4 * cv;
    float4 d[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int x = ox + (j - 1) * geo.d;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int y = oy + (i - 1) * geo.d;
        const bool ok = (unsigned)y < (unsigned)geo.H && (unsigned)x < (unsigned)geo.W;
        if (ok)
          d[i][j] = *reinterpret_cast<const float4 *>(src + ((size_t)y * geo.W + x) * C);
        else
          d[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    float4 v[36];
    wino4_input_tile(d, v);
    const size_t part = (size_t)geo.NT * C;
    unsigned short *dst = V3 + (size_t)g * 36 * 3 * part + ((size_t)chunk * geo.NT + tau) * 32 + 4 * c8;
#pragma unroll
    for (int k = 0; k < 36; ++k) {
      uint2 h, m, l;
      split3x4(v[k], h, m, l);
      unsigned short *p = dst + (size_t)k * 3 * part;
      *reinterpret_cast<uint2 *>(p) = h;
      *reinterpret_cast<uint2 *>(p + part) = m;
      *reinterpret_cast<uint2 *>(p + 2 * part) = l;
    }
  }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma_bf16(const uint4 &a, const uint4 &b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Block tile BM tiles x BN output channels, 8 waves as 2 (m) x 4 (n), weights = the MFMA "A" operand so that a lane ends
// with 4 consecutive output channels of one tile row — the C/D layout of the fp32 kernel (it is dtype-independent on
// gfx950), so the epilogue is the same.  Work items (plane, m-tile, n-tile) are dealt exactly as in wino4_gemm_kernel
// (contiguous eighth of the list per XCD, round-robin inside).
// LDS ring: 3 stages x [3 parts][R = BM + BN rows][64 B].  A DMA instruction lands 16 rows x 64 B (1 KB) of one part;
// the 16-byte slot of a row is XOR-swizzled on the SOURCE side (the LDS image of a DMA is lane-linear) with
// g(row) = (-(row >> 2)) & 3, which makes the four 16-lane groups of a ds_read_b128 fragment read (lanes {0-3, 12-15,
// 20-27}, ...: MI355X_MICROARCH.md, LDS) hit 16 distinct 16-byte slots of the 256-byte bank row.
// Software pipeline (per wave, iteration kc): the 18 fragment reads of chunk kc+1 are issued first (it became visible at
// the barrier that ended iteration kc-1) into the OTHER register set; the 48 MFMAs of chunk kc run from the current set
// with the DMAs of chunk kc+3 — into the stage chunk kc has just vacated — slipped in after every eighth MFMA; then the
// counted wait for chunk kc+2, one barrier.  The loop is unrolled by two so that the register sets swap by name.
// -DW3_ABLATE=n (timing ablations, scripts/w3_ablate.sh; results are WRONG, only the time means something): 1 = no barriers in
// the loop, 2 = no DMA issue in the loop, 3 = no fragment reads in the loop, 4 = one MFMA group instead of six, 5 = DMA only (no
// MFMA, no fragment reads), 6 = 5 without the source-side slot swizzle
#ifndef W3_ABLATE
#define W3_ABLATE 0
#endif
struct W3Frags {
  uint4 a[3][4];   // weights (MFMA A operand): [part][n-tile]   (TN <= 4)
  uint4 b[3][6];   // frequency-domain activations (B operand): [part][m-tile]   (TM <= 6)
};

template <int BM, int BN>
__global__ __launch_bounds__(512) void wino4_gemm3_kernel(const unsigned short *__restrict__ V3,
                                                          const unsigned short *__restrict__ U3, float *__restrict__ Mo,
                                                          int NT, int Cin, int Cout, int planes, int mtiles, int nt_shift) {
  constexpr int WM = BM / 2, WN = BN / 4, TM = WM / 16, TN = WN / 16, R = BM + BN;
  constexpr int NRGA = (BM / 16 + 7) / 8, NRGW = (BN / 16 + 7) / 8, NRG = NRGA + NRGW;   // 16-row groups per wave: V rows, U rows
  constexpr int NPASS = 3 * NRG;
  constexpr int PART_BYTES = R * 64, STAGE_BYTES = 3 * PART_BYTES;
  constexpr int NMMA = 6 * TM * TN;
  static_assert(WM % 16 == 0 && WN % 16 == 0 && BM % 16 == 0 && 3 * STAGE_BYTES <= 160 * 1024 && TM <= 6 && TN <= 4, "tile shape");
  extern __shared__ __attribute__((aligned(16))) char smem3[];

  const int ntiles = 1 << nt_shift;
  const int per_plane = mtiles << nt_shift;
  const int total = planes * per_plane;
  const int S = (total + 7) >> 3;
  const int xcd = blockIdx.x & 7, jloc = blockIdx.x >> 3, nbl = gridDim.x >> 3;
  const int slice_lo = xcd * S, slice_hi = min(total, slice_lo + S);
  const int first = slice_lo + jloc;
  if (first >= slice_hi) return;
  const int n_items = (slice_hi - first + nbl - 1) / nbl;
  const int cpt = Cin / W3BK;
  const int nchunks = n_items * cpt;

  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) char *)smem3);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int t = lane & 15, q = lane >> 4;

  const unsigned part_a_bytes = (unsigned)((size_t)NT * Cin * 2), part_w_bytes = (unsigned)((size_t)Cout * Cin * 2);
  const unsigned chunk_a_bytes = (unsigned)NT * 64u, chunk_w_bytes = (unsigned)Cout * 64u;   // one 32-channel slice of all rows
  const v4i srd_a = make_srd(V3, (unsigned)planes * 3u * part_a_bytes);
  const v4i srd_w = make_srd(U3, (unsigned)planes * 3u * part_w_bytes);

  // DMA geometry of this wave: pass jj < NRGA moves V row group wave + 8 jj, the others U row group wave + 8 (jj - NRGA);
  // a surplus wave re-copies the kind's last group (identical bytes)
  int l_rg[NRG], l_row[NRG];
#pragma unroll
  for (int jj = 0; jj < NRG; ++jj) {
    const bool isw = jj >= NRGA;
    int g = wave + 8 * (isw ? jj - NRGA : jj);
    const int ng = isw ? BN / 16 : BM / 16;
    if (g >= ng) g = ng - 1;
    l_rg[jj] = isw ? BM / 16 + g : g;
    l_row[jj] = g * 16 + (lane >> 2);
  }
  const unsigned l_slot = W3_ABLATE == 6 ? (unsigned)((lane & 3) * 16)
                                         : (unsigned)(((lane & 3) ^ ((-(lane >> 4)) & 3)) * 16);   // logical 16-byte slot this lane fetches

  int iss_plane = first / per_plane, iss_rem = first - iss_plane * per_plane, iss_cc = 0;
  unsigned l_voff[NRG];
  unsigned soff_a, soff_w;
#define W3_ITEM_SETUP()                                                                                   \
  {                                                                                                       \
    const int mt_ = iss_rem >> nt_shift, nt_ = iss_rem & (ntiles - 1);                                    \
    _Pragma("unroll") for (int jj = 0; jj < NRG; ++jj) {                                                  \
      const int row_ = jj >= NRGA ? nt_ * BN + l_row[jj] : min(mt_ * BM + l_row[jj], NT - 1);             \
      l_voff[jj] = (unsigned)row_ * 64u + l_slot;                                                         \
    }                                                                                                     \
    soff_a = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)iss_plane * 3u * part_a_bytes));    \
    soff_w = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)iss_plane * 3u * part_w_bytes));    \
  }
  // DMA number d (0 .. NPASS-1) of a chunk: part d / NRG, row group d % NRG
#define W3_DMA(STG, D)                                                                                    \
  {                                                                                                       \
    constexpr int p_ = (D) / NRG, jj_ = (D) % NRG;                                                        \
    const unsigned dst_ = lds_base + (unsigned)((STG)*STAGE_BYTES + p_ * PART_BYTES + l_rg[jj_] * 1024);  \
    if (W3_ABLATE == 2 && in_loop) {                                                                      \
    } else if (jj_ >= NRGA)                                                                               \
      blds16(srd_w, l_voff[jj_], soff_w + (unsigned)p_ * part_w_bytes, dst_);                             \
    else                                                                                                  \
      blds16(srd_a, l_voff[jj_], soff_a + (unsigned)p_ * part_a_bytes, dst_);                             \
  }
#define W3_ADVANCE()                                                                                      \
  {                                                                                                       \
    if (++iss_cc == cpt) {                                                                                \
      iss_cc = 0;                                                                                         \
      iss_rem += nbl;                                                                                     \
      while (iss_rem >= per_plane) {                                                                      \
        iss_rem -= per_plane;                                                                             \
        ++iss_plane;                                                                                      \
      }                                                                                                   \
      W3_ITEM_SETUP()                                                                                     \
    } else {                                                                                              \
      soff_a += chunk_a_bytes;                                                                            \
      soff_w += chunk_w_bytes;                                                                            \
    }                                                                                                     \
  }

  f32x4 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  int cmp_plane = iss_plane, cmp_rem = iss_rem, cmp_cc = 0;

  auto epilogue = [&]() {
    cmp_cc = 0;
    const int mt = cmp_rem >> nt_shift, nt = cmp_rem & (ntiles - 1);
    float *dst = Mo + (size_t)cmp_plane * NT * Cout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int co = nt * BN + wn * WN + 16 * j + 4 * q;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = mt * BM + wm * WM + 16 * i + t;
        if (m < NT)
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

  // fragment addresses inside a stage: row (tile row t) x 64 B + swizzled slot of k-group q
  const unsigned fslot = (unsigned)((q ^ ((-(t >> 2)) & 3)) * 16);
  const unsigned fu0 = (unsigned)((BM + wn * WN + t) * 64) + fslot, fv0 = (unsigned)((wm * WM + t) * 64) + fslot;
#define W3_READ(F, STG)                                                                                   \
  {                                                                                                       \
    const char *base_ = smem3 + (STG)*STAGE_BYTES;                                                        \
    _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_) {                                                    \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) F.a[p_][j] =                                         \
          *reinterpret_cast<const uint4 *>(base_ + p_ * PART_BYTES + fu0 + j * 1024);                     \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) F.b[p_][i] =                                         \
          *reinterpret_cast<const uint4 *>(base_ + p_ * PART_BYTES + fv0 + i * 1024);                     \
    }                                                                                                     \
  }
  // the six products (a-part, b-part) in the order every output sums them: hh, hm, mh, mm, hl, lh
#define W3_MMA_GROUP(F, PA, PB)                                                                           \
  _Pragma("unroll") for (int j = 0; j < TN; ++j) _Pragma("unroll") for (int i = 0; i < TM; ++i) acc[j][i] = \
      mfma_bf16(F.a[PA][j], F.b[PB][i], acc[j][i]);                                                       \
  __builtin_amdgcn_sched_barrier(0);
  // one chunk: MFMAs of CUR; with ISSUE the NPASS DMAs of the chunk three ahead go out between the product groups
#define W3_COMPUTE(CUR, ISSUE, STG_FREE)                                                                  \
  {                                                                                                       \
    if (W3_ABLATE < 5) { W3_MMA_GROUP(CUR, 0, 0) }                                                        \
    if (ISSUE) { W3_DMA_SLICE(STG_FREE, 0) }                                                              \
    if (W3_ABLATE < 4) { W3_MMA_GROUP(CUR, 0, 1) }                                                        \
    if (ISSUE) { W3_DMA_SLICE(STG_FREE, 1) }                                                              \
    if (W3_ABLATE < 4) { W3_MMA_GROUP(CUR, 1, 0) }                                                        \
    if (ISSUE) { W3_DMA_SLICE(STG_FREE, 2) }                                                              \
    if (W3_ABLATE < 4) { W3_MMA_GROUP(CUR, 1, 1) }                                                        \
    if (W3_ABLATE < 4) { W3_MMA_GROUP(CUR, 0, 2) }                                                        \
    if (W3_ABLATE < 4) { W3_MMA_GROUP(CUR, 2, 0) }                                                        \
  }
  // slice s (0..2) of a chunk's DMAs = part s (NRG instructions)
#define W3_DMA_SLICE(STG, SL)                                                                             \
  {                                                                                                       \
    if constexpr (NRG >= 1) W3_DMA(STG, (SL)*NRG + 0)                                                     \
    if constexpr (NRG >= 2) W3_DMA(STG, (SL)*NRG + (NRG >= 2 ? 1 : 0))                                    \
    if constexpr (NRG >= 3) W3_DMA(STG, (SL)*NRG + (NRG >= 3 ? 2 : 0))                                    \
    if constexpr (NRG >= 4) W3_DMA(STG, (SL)*NRG + (NRG >= 4 ? 3 : 0))                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
  }
#define W3_ISSUE_ALL(STG)                                                                                 \
  {                                                                                                       \
    W3_DMA_SLICE(STG, 0)                                                                                  \
    W3_DMA_SLICE(STG, 1)                                                                                  \
    W3_DMA_SLICE(STG, 2)                                                                                  \
  }

  // ---- prologue: chunks 0, 1, 2 in flight; chunk 0's fragments in set f0 ----
  W3Frags f0, f1;
  bool in_loop = false;
  W3_ITEM_SETUP()
  W3_ISSUE_ALL(0)
  W3_ADVANCE()
  if (nchunks > 1) {
    W3_ISSUE_ALL(1)
    W3_ADVANCE()
  }
  if (nchunks > 2) {
    W3_ISSUE_ALL(2)
    W3_ADVANCE()
  }
  if (nchunks > 2)
    wait_vmcnt<2 * NPASS>();
  else if (nchunks > 1)
    wait_vmcnt<NPASS>();
  else
    wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();            // chunk 0 visible
  W3_READ(f0, 0)
  if (nchunks > 2)
    wait_vmcnt<NPASS>();
  else
    wait_vmcnt<0>();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();            // chunk 1 visible; every wave holds chunk 0 in registers: stage 0 is free
  if (nchunks > 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);

  // iteration kc: registers (CUR) hold chunk kc; chunk kc+1 is visible in stage s_nxt; stage s_cur is free (every wave read
  // chunk kc from it before the last barrier) and takes chunk kc+3.  (A deeper variant — a second barrier behind the first
  // product group frees stage s_nxt for chunk kc+4, the whole ring in flight — measured SLOWER: 379 vs 352 us on layer4; the loop
  // is bound by the DMA path's throughput, not by bytes in flight.)
#define W3_ITER(CUR, NXT)                                                                                 \
  {                                                                                                       \
    const bool more1_ = kc + 1 < nchunks, more3_ = kc + 3 < nchunks;                                      \
    if (more1_ && W3_ABLATE != 3 && W3_ABLATE < 5) W3_READ(NXT, s_nxt)                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    if (more3_) {                                                                                         \
      W3_COMPUTE(CUR, true, s_cur)                                                                        \
      W3_ADVANCE()                                                                                        \
    } else {                                                                                              \
      W3_COMPUTE(CUR, false, s_cur)                                                                       \
    }                                                                                                     \
    if (++cmp_cc == cpt) epilogue();                                                                      \
    if (more3_)                                                                                           \
      wait_vmcnt<NPASS>();   /* chunk kc+2 has landed: only chunk kc+3's DMAs may be in flight */          \
    else                                                                                                  \
      wait_vmcnt<0>();                                                                                    \
    __builtin_amdgcn_s_waitcnt(0xC07F);   /* lgkmcnt(0): this wave's reads of stage s_nxt are done */     \
    if (W3_ABLATE != 1) __builtin_amdgcn_s_barrier();                                                     \
    const int tmp_ = s_cur;                                                                               \
    s_cur = s_nxt;                                                                                        \
    s_nxt = s_nn;                                                                                         \
    s_nn = tmp_;                                                                                          \
  }
  int s_cur = 0, s_nxt = 1, s_nn = 2;
  int kc = 0;
  in_loop = true;
  if (W3_ABLATE == 3) { W3_READ(f1, 1) }
  for (; kc + 1 < nchunks; kc += 2) {
    W3_ITER(f0, f1)
    ++kc;
    W3_ITER(f1, f0)
    --kc;
  }
  if (kc < nchunks) W3_ITER(f0, f1)
#undef W3_ITEM_SETUP
#undef W3_DMA
#undef W3_ADVANCE
#undef W3_READ
#undef W3_MMA_GROUP
#undef W3_COMPUTE
#undef W3_DMA_SLICE
#undef W3_ISSUE_ALL
#undef W3_ITER
}

template <int BM, int BN>
static int launch_gemm3_t(const unsigned short *V3, const unsigned short *U3, float *Mo, int NT, int Cin, int Cout, int planes,
                          int nblocks, hipStream_t st) {
  const int mtiles = (NT + BM - 1) / BM, ntiles = Cout / BN;
  int nt_shift = 0;
  while ((1 << nt_shift) < ntiles) ++nt_shift;
  UOC_REQUIRE((1 << nt_shift) == ntiles, "winograd F(4x4) bf16x3: Cout / %d = %d is not a power of two", BN, ntiles);
  const size_t lds = (size_t)3 * 3 * (BM + BN) * 64;
  static DeviceOnce attr_set;
  if (!attr_set.done()) {
    UOC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&wino4_gemm3_kernel<BM, BN>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.mark();
  }
  hipLaunchKernelGGL((wino4_gemm3_kernel<BM, BN>), dim3(nblocks), dim3(512), lds, st, V3, U3, Mo, NT, Cin, Cout, planes, mtiles,
                     nt_shift);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

// Tile choice: the static cost model of csrc/wino4.hip over the tiles whose 3-stage ring fits 160 KB (BM + BN <= 256)
static const int kW3Tiles[6][2] = {{128, 128}, {96, 128}, {192, 64}, {160, 64}, {128, 64}, {96, 64}};
static int launch_gemm3(const unsigned short *V3, const unsigned short *U3, float *Mo, int NT, int Cin, int Cout, int planes,
                        hipStream_t st) {
  const int ncu = device_num_cu() > 0 ? device_num_cu() : 256;
  const int per_xcd = ncu / 8 > 0 ? ncu / 8 : 1;
  double best = -1;
  int bm = 128, bn = Cout % 128 == 0 ? 128 : 64;
  for (const auto &tl : kW3Tiles) {
    const int BM = tl[0], BN = tl[1];
    if (Cout % BN) continue;
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
  const long items = (long)planes * ((NT + bm - 1) / bm) * (Cout / bn);
  const long S = (items + 7) / 8;
  const int nblocks = 8 * (int)(S < per_xcd ? S : per_xcd);
#define W3_CASE(A, B) \
  if (bm == A && bn == B) return launch_gemm3_t<A, B>(V3, U3, Mo, NT, Cin, Cout, planes, nblocks, st);
  W3_CASE(128, 128) W3_CASE(96, 128) W3_CASE(192, 64) W3_CASE(160, 64) W3_CASE(128, 64) W3_CASE(96, 64)
#undef W3_CASE
  set_error("winograd F(4x4) bf16x3: no GEMM tile %dx%d", bm, bn);
  return UOC_EINVAL;
}

int launch_wino4_split_weights(const float *U, unsigned short *U3, int G, int Cout, int Cin, hipStream_t st) {
  const long planes = 36l * G, per_plane = (long)Cout * Cin;
  long blocks = (planes * per_plane / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wino4_split_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, st, U, U3, planes, Cout, Cin);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

// images b0 .. b0 + p.B - 1 of a batch of Bg: the three stages with the split-precision GEMM.  ws layout: V3 (1.5 halves of
// wino4_ws_floats's unit) then M.
int launch_wino4_slice_split(const ConvParams &p0, int Bg, int b0, const unsigned short *U3, float *ws, hipStream_t st) {
  ConvParams p = p0;
  const size_t img_in = (size_t)p.H * p.W * p.Cin, img_out = (size_t)p.H * p.W * p.Cout;
  p.in += (size_t)b0 * img_in;
  p.out += (size_t)b0 * img_out;
  if (p.res) p.res += (size_t)b0 * img_out;
  Wino4Geom geo = make_geom4(p.B, p.H, p.W, p.dil);
  geo.Bg = Bg;
  const int planes = 36 * p.G;
  unsigned short *V3 = reinterpret_cast<unsigned short *>(ws);
  const size_t v3_floats = ((size_t)planes * 3 * geo.NT * p.Cin + 1) / 2;
  float *Mw = ws + (v3_floats + 63) / 64 * 64;
  UOC_REQUIRE((size_t)planes * 3 * geo.NT * p.Cin * 2 < (1ull << 32) && (size_t)planes * 3 * p.Cout * p.Cin * 2 < (1ull << 32),
              "winograd F(4x4) bf16x3: operand planes exceed the 4 GB a 32-bit buffer offset addresses");
  const double Mpix = (double)p.B * p.H * p.W;
  const ProfTag tag = {{geo.NT, p.Cin, p.Cout, p.dil}};
  {
    ProfScope prof(KC_WINO4_INPUT, st, 0.0, p.G * (4.0 * Mpix * p.Cin + 6.0 * 36.0 * geo.NT * p.Cin), tag);
    const long blocks = wino4_elem_blocks((long)p.G * geo.NT * (p.Cin / 4));
    hipLaunchKernelGGL(wino4_input3_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p.in, V3, geo, p.G, p.Cin);
    UOC_LAUNCH_CHECK();
  }
  {
    const double gflops = 2.0 * Mpix * p.Cout * p.Cin * 9.0 * p.G;
    const double gbytes = (double)planes * (6.0 * geo.NT * p.Cin + 6.0 * p.Cout * p.Cin + 4.0 * geo.NT * p.Cout);
    ProfScope prof(KC_WINO4_GEMM, st, gflops, gbytes, tag);
    if (int rc = launch_gemm3(V3, U3, Mw, geo.NT, p.Cin, p.Cout, planes, st)) return rc;
  }
  return w4_stage_output(p, geo, Mw, st);
}

}  // namespace uoc
