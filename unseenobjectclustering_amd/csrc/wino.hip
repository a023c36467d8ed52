// Winograd F(2x2, 3x3) convolution for the 3x3 stride-1 (possibly dilated) layers of the backbone:
// 2.25x fewer matrix-core flops than the direct implicit GEMM, still exact-fp32 MFMA arithmetic.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A          per 2x2 output tile, 4x4 input patch, 16 "frequencies"
//
// Dilation d is handled by phase decomposition: outputs of one phase (y mod d, x mod d) depend only on
// inputs of the same phase, so a tile covers outputs (oy, ox) + {0,d}x{0,d} and its patch the inputs
// (oy, ox) + {-d,0,d,2d}^2; zero padding is the out-of-range zero fill of the patch.
//
// Three kernels (csrc/net.hip decides per layer; results differ from the direct kernel only by fp32
// rounding, ~1e-6 relative):
//   wino_weight_kernel  U[g][xi][cin/32][cout][32] = G g G^T    once, at uoc_net_finalize
//   wino_input_kernel   V[g][xi][cin/32][tile][32] = B^T d B    elementwise, NHWC float4
//   wino_gemm_kernel    M_xi = U_xi V_xi^T on v_mfma_f32_16x16x4_f32, the OUTPUT transform folded in:
//     8 waves = 2 frequency groups (xi 0-7 / 8-15) x 4 groups of 16 output channels; a group walks its
//     8 frequencies one after the other (K order: frequency outer, cin inner), and after the last cin
//     chunk of a frequency folds M_xi into the four 2x2-output accumulators with the +-1/0 coefficients
//     of A^T (.) A^T.  The two groups' partial outputs meet once in LDS; bias, residual and ReLU are
//     applied in the epilogue.  Operand staging is the LDS-DMA ring of csrc/conv.hip (3 stages, two
//     chunks ahead, counted vmcnt, source-side XOR swizzle, zero page).
// DEVELOPMENT BUILDS ONLY (-DUOC_DEV): the shipped library runs every eligible layer as F(4x4,3x3) (csrc/wino4.hip) since
// round 3; these kernels stay as the measured A/B alternative (profiles/r03_ab_winograd_f{2,4}.json).
#ifdef UOC_DEV
#include "conv.h"
#include "prof.h"

#include <stdlib.h>

namespace uoc {

constexpr int WBK = 32;  // cin chunk
constexpr int WBN = 64;  // output channels per block

struct WinoGeom {
  int B, H, W, d, TH, TW, NT;  // TH x TW tiles per (image, phase); NT = B*d*d*TH*TW
};

static WinoGeom make_geom(int B, int H, int W, int d) {
  WinoGeom g;
  g.B = B;
  g.H = H;
  g.W = W;
  g.d = d;
  g.TH = ((H + d - 1) / d + 1) / 2;
  g.TW = ((W + d - 1) / d + 1) / 2;
  g.NT = B * d * d * g.TH * g.TW;
  return g;
}

__device__ __forceinline__ void wino_decode(int tau, const WinoGeom &g, int &b, int &oy, int &ox) {
  const int tx = tau % g.TW;
  tau /= g.TW;
  const int ty = tau % g.TH;
  tau /= g.TH;
  const int px = tau % g.d;
  tau /= g.d;
  const int py = tau % g.d;
  b = tau / g.d;
  oy = py + 2 * ty * g.d;
  ox = px + 2 * tx * g.d;
}

// ---- weights: U = G g G^T,  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] ---------------------------
__global__ __launch_bounds__(256) void wino_weight_kernel(const float *__restrict__ w, float *__restrict__ U, int G,
                                                          int Cout, int Cin) {
  const long total = (long)G * Cout * Cin;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin);
    const int co = (int)((i / Cin) % Cout);
    const int g = (int)(i / ((long)Cin * Cout));
    float k[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) k[a][b] = w[(((size_t)g * 9 + a * 3 + b) * Cout + co) * Cin + ci];
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      t[0][b] = k[0][b];
      t[1][b] = 0.5f * (k[0][b] + k[1][b] + k[2][b]);
      t[2][b] = 0.5f * (k[0][b] - k[1][b] + k[2][b]);
      t[3][b] = k[2][b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float u0 = t[a][0];
      const float u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]);
      const float u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]);
      const float u3 = t[a][2];
      // chunk-major: U[g][xi][cin / 32][cout][32] — the rows of one (frequency, cin chunk) are contiguous
      const size_t o = ((((size_t)g * 16 + a * 4) * (Cin / WBK) + ci / WBK) * Cout + co) * WBK + ci % WBK;
      const size_t s = (size_t)Cout * Cin;
      U[o] = u0;
      U[o + s] = u1;
      U[o + 2 * s] = u2;
      U[o + 3 * s] = u3;
    }
  }
}

// ---- input: V = B^T d B,  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]] --------------------------
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__global__ __launch_bounds__(256) void wino_input_kernel(const float *__restrict__ in, float *__restrict__ V,
                                                         WinoGeom geo, int G, int C) {
  const int C4 = C >> 2;
  const long total = (long)G * geo.NT * C4;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    const int tau = (int)((idx / C4) % geo.NT);
    const int g = (int)(idx / ((long)C4 * geo.NT));
    int b, oy, ox;
    wino_decode(tau, geo, b, oy, ox);
    const float *src = in + (((size_t)g * geo.B + b) * geo.H * geo.W) * C + 4 * c4;
    float4 dd[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = oy + (i - 1) * geo.d;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int x = ox + (j - 1) * geo.d;
        const bool ok = (unsigned)y < (unsigned)geo.H && (unsigned)x < (unsigned)geo.W;
        dd[i][j] = ok ? *reinterpret_cast<const float4 *>(src + ((size_t)y * geo.W + x) * C) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    float4 t[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[0][j] = f4sub(dd[0][j], dd[2][j]);
      t[1][j] = f4add(dd[1][j], dd[2][j]);
      t[2][j] = f4sub(dd[2][j], dd[1][j]);
      t[3][j] = f4sub(dd[1][j], dd[3][j]);
    }
    // chunk-major: V[g][xi][cin / 32][tile][32]: a block's rows of one (frequency, cin chunk) are one
    // contiguous run, so consecutive rows fall into consecutive L2 channels (with [tile][cin] rows every
    // row of a chunk is Cin*4 bytes apart and a 512-channel layer hits ONE of the 16 channels per XCD)
    const size_t plane = (size_t)geo.NT * C;
    float *dst = V + (size_t)g * 16 * plane + ((size_t)(c4 >> 3) * geo.NT + tau) * WBK + 4 * (c4 & 7);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4 *>(dst + (size_t)(4 * i + 0) * plane) = f4sub(t[i][0], t[i][2]);
      *reinterpret_cast<float4 *>(dst + (size_t)(4 * i + 1) * plane) = f4add(t[i][1], t[i][2]);
      *reinterpret_cast<float4 *>(dst + (size_t)(4 * i + 2) * plane) = f4sub(t[i][2], t[i][1]);
      *reinterpret_cast<float4 *>(dst + (size_t)(4 * i + 3) * plane) = f4sub(t[i][1], t[i][3]);
    }
  }
}

// ---- GEMM over the 16 frequencies with the output transform folded in -----------------------------
__device__ float4 g_wino_zero[8];  // zero page for tile rows beyond NT

// A^T (.) A^T coefficients: cY[xi = 4i+j][k = 2a+b] = At[a][i] * At[b][j],  At = [[1,1,1,0],[0,1,-1,-1]]
__constant__ float c_wino_y[16][4] = {
    {1, 0, 0, 0},  {1, 1, 0, 0},   {1, -1, 0, 0},  {0, -1, 0, 0},   // i = 0
    {1, 0, 1, 0},  {1, 1, 1, 1},   {1, -1, 1, -1}, {0, -1, 0, -1},  // i = 1
    {1, 0, -1, 0}, {1, 1, -1, -1}, {1, -1, -1, 1}, {0, -1, 0, 1},   // i = 2
    {0, 0, -1, 0}, {0, 0, -1, -1}, {0, 0, -1, 1},  {0, 0, 0, 1},    // i = 3
};

// one LDS-DMA (see glds16 in conv.hip) with the address split into a wave-uniform 64-bit base (SGPR pair) and a per-lane 32-bit byte offset:
// no VALU instruction per load (MFMA issue shares its port with the VALU: scripts/mfma_patterns.hip)
__device__ __forceinline__ void wglds16s(const float *base, unsigned voff, unsigned lds_dst) {
  // m0 is written without being saved / restored (two scalar moves per DMA less: scalar instructions in a wave's stream
  // delay its MFMA issue like everything else).  Nothing else in these kernels lives in m0 — gfx9+ LDS instructions do
  // not read it and hipcc sets it itself right before any use of its own (movrel, sendmsg); declaring it clobbered makes
  // hipcc wrap the statement in the very save / restore this avoids.
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wwait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ f32x4 wmfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// VARIANT != 0: timing ablations only (wrong results; UOC_WINO_VARIANT, dev): 1 = no DMA in the loop,
// 4 = full, but every chunk re-reads chunk 0 (same LDS traffic, all cache hits).  Measured on MI355X, layer4 at
// TMT = 5: full 207 us, no DMA 170 us, DMA from cache-resident rows 183 us, MFMA + fold alone 156 us
// (= 70 % matrix-pipe busy in the full kernel, SQ_VALU_MFMA_BUSY_CYCLES).  What did NOT move the 37 us the
// L2-sourced DMA costs: a 4-deep ring (NSTG = 4; so it is not latency), two dedicated producer waves issuing
// all DMA (so it is not MFMA waves stalling on DMA issue), chunk-major U / V (-2 %), fewer L2 misses through
// the XCD mapping (0 %), and loading each wave's own U fragments straight into registers instead of through
// LDS (-44 % DMA bytes, +0 %: the register loads cost what the DMA saved).  What is common to all of them is
// the bytes a CU pulls out of L2 per MFMA (TCP_PENDING_STALL_CYCLES = 35 % of the kernel; the demand, 5.6 TB/s
// chip-wide, is at the 6.4 TB/s MI355X_MICROARCH.md measures for LDS-DMA streams).
// Round 2 tested that reading and it does NOT hold: a variant with a 128-channel block tile (all eight waves on one
// frequency, BM + 128 rows per chunk: 10.4 instead of 14.4 B of operands per matrix-pipe cycle at TMT = 5, 8.6 at TMT = 7)
// measured 573 vs 578 us on three frames of layer4 and 153 vs 168 us on layer3 — within noise — and was removed
// again.  In-kernel counters (-DUOC_WINO_CLOCK): 2.33 GHz, 40.2 / 41.1 / 41.8 / 48.4 cycles per MFMA slot at
// TMT 7 / 5 / 4 / 3 against 32, i.e. an overhead of ~240 + 97*TMT cycles per chunk and SIMD that scales WITH the tile.
// The part that scales is consistent with the 4 fragment reads per 16 MFMAs (a ds_read_b128 costs ~20-24 cycles of the
// fp32 lanes, profiles/r02_mfma_microbenchmarks.md).  A register-blocked variant (wave tile 32 channels x 16*BT tiles,
// waves of a frequency group 2 x 2: (2 + BT) / (8 BT) = 0.19 reads per MFMA at BT = 4 instead of 0.30) was built and
// verified (all Winograd + network goldens) and measured 556 vs 574 us (three frames of layer4), 150 vs 162 us (three
// frames of layer3) — 39.5 cycles per MFMA slot — but 5-40 % slower on every shape whose tile count does not fill
// whole rounds of its fixed 64 / 96 / 128-tile blocks; ~1 % of a frame, not kept (scripts/wino2_bench.py has the table).
template <int TMT, int NSTG, int VARIANT = 0>
__global__ __launch_bounds__(512) void wino_gemm_kernel(const float *__restrict__ V, const float *__restrict__ U,
                                                        const float *__restrict__ bias_, const float *__restrict__ res_,
                                                        float *__restrict__ out_, WinoGeom geo, int G, int Cin, int Cout,
                                                        int relu, int ntiles, int mtiles) {
  constexpr int BM = 16 * TMT;      // winograd tiles per block
  constexpr int SEG = BM + WBN;     // rows of one frequency group in a stage: V rows, then U rows
  constexpr int R = 2 * SEG;        // two groups work on two different frequencies at once
  constexpr int RPP = 64;           // 8 waves x 8 rows per DMA pass
  constexpr int NPASS = (R + RPP - 1) / RPP;
  constexpr int STAGE = R * WBK;
  static_assert(SEG % 16 == 0 && NPASS <= 8 && NSTG >= 3 && NSTG <= 4, "tile shape");

  extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef UOC_WINO_CLOCK
  const unsigned long long dbg_c0 = __builtin_readcyclecounter(), dbg_r0 = wall_clock64();
#endif

  const int total = G * ntiles * mtiles;
  const int per_xcd = (total + 7) >> 3;
  const int work = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (work >= total) return;
  // tile-row outer, channel-tile inner: the blocks that share an XCD's L2 cover a few tile rows x ALL channel
  // tiles, which minimises the distinct U + V rows the XCD pulls in per chunk (layer4: 208 KB instead of 352 KB)
  const int g = work / (ntiles * mtiles);
  const int rem = work - g * (ntiles * mtiles);
  const int mt = rem / ntiles, nt = rem - mt * ntiles;
  const int m0 = mt * BM, n0 = nt * WBN;
  const int NT = geo.NT;
  const int cpt = Cin / WBK;
  const int nit = 8 * cpt;  // >= 8 > NSTG

  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) char *)smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int f = wave >> 2, wn = wave & 3;  // frequency group, 16-cout group
  const int t = lane & 15, q = lane >> 4;

  // ---- DMA descriptors ---------------------------------------------------------------------------
  constexpr int NISS = NPASS;  // DMA instructions per wave and chunk: 8 waves x 8 rows per pass
  static_assert((NSTG - 1) * NISS < 64, "vmcnt range");
  // A pass is 8 rows of ONE kind (SEG and BM are multiples of 16): kind (bit0 = U row, bit1 = frequency group)
  // is wave-uniform, the lane keeps a 32-bit byte offset.  Tile rows beyond NT read the last valid row instead
  // of a zero page: their products land in accumulators of tiles the epilogue never stores.
  int d_r0[NISS], d_kind[NISS];
  unsigned d_off[NISS];
#pragma unroll
  for (int j = 0; j < NISS; ++j) {
    int r0 = j * RPP + wave * 8;
    if (r0 >= R) r0 = R - 8;
    d_r0[j] = r0;
    const int fr0 = r0 >= SEG ? 1 : 0;
    d_kind[j] = ((r0 - fr0 * SEG) >= BM ? 1 : 0) | (fr0 << 1);
    const int r = r0 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    const int rr = r - fr0 * SEG;
    if (rr >= BM) {
      d_off[j] = (unsigned)(((n0 + rr - BM) * WBK + 4 * c) * sizeof(float));
    } else {
      int tau = m0 + rr;
      if (tau >= NT) tau = NT - 1;
      d_off[j] = (unsigned)((tau * WBK + 4 * c) * sizeof(float));
    }
  }
  // Row base of every DMA pass for the chunk being issued.  Chunk index = (g*16 + frequency)*cpt + cin chunk runs
  // contiguously through a frequency group's 8 frequencies, so each pass keeps ONE wave-uniform 64-bit pointer (U or V
  // plane of its frequency group) and advances it by one chunk per issue: NISS scalar 64-bit adds instead of
  // re-deriving four bases and selecting among them (the loop had ~100 SALU per chunk).
  const float *q_ptr[NISS];
  size_t q_step[NISS];
#pragma unroll
  for (int j = 0; j < NISS; ++j) {
    const int kd = d_kind[j];
    const size_t step = (kd & 1) ? (size_t)Cout * WBK : (size_t)NT * WBK;
    q_step[j] = VARIANT == 4 ? 0 : step;
    q_ptr[j] = ((kd & 1) ? U : V) + ((size_t)g * 16 + ((kd & 2) ? 8 : 0)) * cpt * step;
  }
#define W_ISSUE(STG)                                                                                    \
  {                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < NISS; ++j) {                                                  \
      wglds16s(q_ptr[j], d_off[j], lds_base + (unsigned)(((STG)*STAGE + d_r0[j] * WBK) * sizeof(float))); \
      q_ptr[j] += q_step[j];                                                                            \
    }                                                                                                   \
  }
  // before the barrier of step `it`: chunk it+1 must have landed; younger chunks (up to NSTG-2 of them) may fly
#define W_WAIT_NEXT(IT)                                \
  {                                                    \
    if (VARIANT == 1)                                  \
      wwait_vmcnt<0>();                                \
    else if (NSTG == 4 && (IT) + 3 < nit)              \
      wwait_vmcnt<2 * NISS>();                         \
    else if ((IT) + 2 < nit)                           \
      wwait_vmcnt<NISS>();                             \
    else                                               \
      wwait_vmcnt<0>();                                \
  }
#define W_FRAG(STG, HH, WA, XB)                                                                            \
  {                                                                                                        \
    const float *base_ = smem + (STG)*STAGE + f * SEG * WBK;                                               \
    const int slot_ = ((4 * (HH) + q) ^ ((t >> 1) & 7)) * 4;                                               \
    WA = *reinterpret_cast<const float4 *>(base_ + (BM + wn * 16 + t) * WBK + slot_);                      \
    _Pragma("unroll") for (int i = 0; i < TMT; ++i) XB[i] =                                                \
        *reinterpret_cast<const float4 *>(base_ + (16 * i + t) * WBK + slot_);                             \
  }
#define W_MFMA_E(WA, XB, E) \
  { _Pragma("unroll") for (int i = 0; i < TMT; ++i) M[i] = wmfma(WA.E, XB[i].E, M[i]); }

  f32x4 M[TMT], Y[4][TMT];
#pragma unroll
  for (int i = 0; i < TMT; ++i) {
    M[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) Y[k][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  float4 wa0, wa1, xb0[TMT], xb1[TMT];
  // prologue: chunks 0 .. NSTG-2 into stages 0 .. NSTG-2
  W_ISSUE(0)
  W_ISSUE(1)
  if (NSTG == 4) W_ISSUE(2)
  wwait_vmcnt<(NSTG - 2) * NISS>();  // chunk 0 has landed
  // stage of chunk it-1 (= of chunk it+NSTG-1, the one issued during step it), of chunk it, of chunk it+1
  int s_prev = NSTG - 1, s_cur = 0, s_nxt = 1;
#define W_ROTATE()                               \
  {                                              \
    s_prev = s_cur;                              \
    s_cur = s_nxt;                               \
    s_nxt = s_nxt + 1 == NSTG ? 0 : s_nxt + 1;   \
  }
  {
    __builtin_amdgcn_s_barrier();
    W_FRAG(0, 0, wa0, xb0)
    int cc = 0, e_cur = 0;  // cin chunk / frequency-pair of the chunk being multiplied
    const bool early = wave < 4;
    for (int it = 0; it < nit; ++it) {
      const bool more = it + NSTG - 1 < nit && VARIANT != 1;
      if (more && early) W_ISSUE(s_prev)
      W_MFMA_E(wa0, xb0, x)
      W_FRAG(s_cur, 1, wa1, xb1)
      __builtin_amdgcn_sched_barrier(0);
      W_MFMA_E(wa0, xb0, y)
      W_MFMA_E(wa0, xb0, z)
      W_MFMA_E(wa0, xb0, w)
      if (more && !early) W_ISSUE(s_prev)
      W_WAIT_NEXT(it)
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
      if (it + 1 < nit) W_FRAG(s_nxt, 0, wa0, xb0)
      __builtin_amdgcn_sched_barrier(0);
      W_MFMA_E(wa1, xb1, x)
      W_MFMA_E(wa1, xb1, y)
      W_MFMA_E(wa1, xb1, z)
      W_MFMA_E(wa1, xb1, w)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_sched_barrier(0);
      if (++cc == cpt) {  // frequency xi = 8f + e_cur is complete: fold it into the four output accumulators
        cc = 0;
        const int xi = 8 * f + e_cur;
        ++e_cur;
        const float c0 = c_wino_y[xi][0], c1 = c_wino_y[xi][1], c2 = c_wino_y[xi][2], c3 = c_wino_y[xi][3];
#pragma unroll
        for (int i = 0; i < TMT; ++i) {
          Y[0][i] += c0 * M[i];
          Y[1][i] += c1 * M[i];
          Y[2][i] += c2 * M[i];
          Y[3][i] += c3 * M[i];
          M[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      W_ROTATE()
    }
  }
#undef W_ROTATE
#undef W_WAIT_NEXT
#undef W_ISSUE
#undef W_FRAG
#undef W_MFMA_E

#ifdef UOC_WINO_CLOCK
  if (blockIdx.x == 8 && threadIdx.x == 0) {
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    printf("[wino clock] TMT %d nit %d: main loop %.1f us, %llu cycles -> %.0f MHz, %.1f cycles per MFMA slot (2 waves/SIMD x %d MFMAs per chunk)\n",
           TMT, nit, (r1 - dbg_r0) / 100.0, c1 - dbg_c0, (c1 - dbg_c0) / ((r1 - dbg_r0) / 100.0), (double)(c1 - dbg_c0) / ((double)nit * 16 * TMT), 8 * TMT);
  }
#endif
  // ---- the two frequency groups meet in LDS; group 0 writes the 2x2 outputs ---------------------------
  f32x4 *red = reinterpret_cast<f32x4 *>(smem);  // [wn][k][i][lane]
  __syncthreads();
  if (f == 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < TMT; ++i) red[((wn * 4 + k) * TMT + i) * 64 + lane] = Y[k][i];
  }
  __syncthreads();
  if (f == 0) {
    const size_t gsz = (size_t)geo.B * geo.H * geo.W * Cout;
    const float *bias = bias_ + (size_t)g * Cout;
    const float *res = res_ ? res_ + (size_t)g * gsz : nullptr;
    float *out = out_ + (size_t)g * gsz;
    const int co = n0 + wn * 16 + 4 * q;
    const float4 bv = *reinterpret_cast<const float4 *>(bias + co);
#pragma unroll
    for (int i = 0; i < TMT; ++i) {
      const int tau = m0 + 16 * i + t;
      if (tau >= NT) continue;
      int b, oy, ox;
      wino_decode(tau, geo, b, oy, ox);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 yv = Y[k][i] + red[((wn * 4 + k) * TMT + i) * 64 + lane];
        const int y = oy + (k >> 1) * geo.d, x = ox + (k & 1) * geo.d;
        if (y < geo.H && x < geo.W) {
          const size_t o = (((size_t)b * geo.H + y) * geo.W + x) * Cout + co;
          float4 v = make_float4(yv[0] + bv.x, yv[1] + bv.y, yv[2] + bv.z, yv[3] + bv.w);
          if (res) {
            const float4 rv = *reinterpret_cast<const float4 *>(res + o);
            v.x += rv.x;
            v.y += rv.y;
            v.z += rv.z;
            v.w += rv.w;
          }
          if (relu) {
            v.x = fmaxf(v.x, 0.f);
            v.y = fmaxf(v.y, 0.f);
            v.z = fmaxf(v.z, 0.f);
            v.w = fmaxf(v.w, 0.f);
          }
          *reinterpret_cast<float4 *>(out + o) = v;
        }
      }
    }
  }
}

template <int TMT, int NSTG, int VARIANT = 0>
static int launch_wino_gemm(const ConvParams &p, const float *U, const float *V, const WinoGeom &geo, hipStream_t st) {
  constexpr int BM = 16 * TMT;
  const int mtiles = (geo.NT + BM - 1) / BM, ntiles = p.Cout / WBN;
  const size_t lds = (size_t)NSTG * 2 * (BM + WBN) * WBK * sizeof(float);
  static_assert((size_t)NSTG * 2 * (BM + WBN) * WBK * sizeof(float) <= 160 * 1024, "LDS ring too large");
  static DeviceOnce attr_set;
  if (!attr_set.done()) {
    UOC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&wino_gemm_kernel<TMT, NSTG, VARIANT>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.mark();
  }
  const int total = mtiles * ntiles * p.G;
  hipLaunchKernelGGL((wino_gemm_kernel<TMT, NSTG, VARIANT>), dim3(((total + 7) / 8) * 8), dim3(512), lds, st,
                     V, U, p.bias, p.res, p.out, geo, p.G, p.Cin, p.Cout, p.relu, ntiles, mtiles);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

bool wino_eligible(const ConvParams &p) {
  return !p.stem && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == p.dil && p.Cin % WBK == 0 && p.Cout % WBN == 0 &&
         p.Ho == p.H && p.Wo == p.W;
}

size_t wino_v_floats(int G, int B, int H, int W, int d, int Cin) {
  const WinoGeom geo = make_geom(B, H, W, d);
  return (size_t)G * 16 * geo.NT * Cin;
}

int launch_wino_weights(const float *w, float *U, int G, int Cout, int Cin, hipStream_t st) {
  const long total = (long)G * Cout * Cin;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w, U, G, Cout, Cin);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

int launch_wino_conv(const ConvParams &p, const float *U, float *Vws, hipStream_t st) {
  UOC_REQUIRE(wino_eligible(p), "winograd: layer not eligible");
  UOC_REQUIRE(U && Vws, "winograd: null weight/workspace pointer");
  const WinoGeom geo = make_geom(p.B, p.H, p.W, p.dil);
  {
    ProfScope prof(KC_WINO_INPUT, st, 0.0, 4.0 * p.G * ((double)p.B * p.H * p.W * p.Cin + 16.0 * geo.NT * p.Cin),
                   ProfTag{{geo.NT, p.Cin, p.Cout, p.dil}});
    const long total = (long)p.G * geo.NT * (p.Cin / 4);
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p.in, Vws, geo, p.G, p.Cin);
  }
  // pixel-tile height: whole rounds of 256 CUs, mild preference for tall tiles
  const int ntiles = p.Cout / WBN;
  int best = 5;
  double best_cost = -1;
  for (int tmt = 2; tmt <= 7; ++tmt) {
    const long blocks = (long)((geo.NT + 16 * tmt - 1) / (16 * tmt)) * ntiles * p.G;
    const long rounds = (blocks + 255) / 256;
    // measured per-unit cost (MI355X, scripts/wino_microbench.py): flat from 4 to 7 tile rows, +20 % for 2 and 3
    const double cost = (double)rounds * tmt * (tmt <= 3 ? 1.2 : 1.0);
    if (best_cost < 0 || cost <= best_cost) {
      best = tmt;
      best_cost = cost;
    }
  }
  const double M = (double)p.B * p.H * p.W;
  ProfScope prof(KC_WINO_GEMM, st, 2.0 * M * p.Cout * p.Cin * 9.0 * p.G,
                 4.0 * p.G * (16.0 * geo.NT * p.Cin + 16.0 * p.Cout * p.Cin + M * p.Cout * (p.res ? 2 : 1)),
                 ProfTag{{geo.NT, p.Cin, p.Cout, p.dil}});
  if (const char *e = getenv("UOC_WINO_TMT")) {  // dev: force the tile height
    const int v = atoi(e);
    if (v >= 2 && v <= 7) best = v;
  }
#ifdef UOC_DEV   // timing ablations of the TMT=5 kernel (WRONG results): development builds only
  if (const char *e = getenv("UOC_WINO_VARIANT")) {  // dev: timing ablations of the TMT=5 kernel
    switch (atoi(e)) {
      case 1: return launch_wino_gemm<5, 3, 1>(p, U, Vws, geo, st);
      case 4: return launch_wino_gemm<5, 3, 4>(p, U, Vws, geo, st);
      default: break;
    }
  }
#endif
  switch (best) {
    case 2: return launch_wino_gemm<2, 3>(p, U, Vws, geo, st);
    case 3: return launch_wino_gemm<3, 3>(p, U, Vws, geo, st);
    case 4: return launch_wino_gemm<4, 3>(p, U, Vws, geo, st);
    case 6: return launch_wino_gemm<6, 3>(p, U, Vws, geo, st);
    case 7: return launch_wino_gemm<7, 3>(p, U, Vws, geo, st);
    default: return launch_wino_gemm<5, 3>(p, U, Vws, geo, st);
  }
}

}  // namespace uoc

#endif  // UOC_DEV
