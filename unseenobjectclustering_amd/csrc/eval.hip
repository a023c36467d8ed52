// Integer statistics behind multilabel_metrics (lib/utils/evaluation.py:109-257) for one predicted and one
// ground-truth label map, as three small kernels; the float arithmetic (precision / recall / F, Hungarian
// matching) stays on the host where the reference does it in float64.
//
//   cont[g][p]      = #pixels with gt == g and pred == p                               (:188-190 true positives)
//   bnd_pred[p]     = #boundary pixels of mask (pred == p), seg2bmap (:15-73)           (:212-215 denominators)
//   bnd_gt[g]       = same for the ground truth                                         (:216-219)
//   prec_tp[g][p]   = #boundary pixels of pred p that lie in the dilated boundary of gt g   (:98-104, fg_match)
//   rec_tp[g][p]    = #boundary pixels of gt g that lie in the dilated boundary of pred p   (gt_match)
//
// seg2bmap of a binary mask marks pixel (y,x) when the mask differs from its east, south or south-east
// neighbour; the last row only looks east, the last column only south, the corner is never marked.  For a LABEL
// map this means: where the 2x2 neighbourhood is not uniform, (y,x) is a boundary pixel of every distinct label
// in it — at most four, packed into one 32-bit word per pixel (0xFF = empty slot; label 0 is background and is
// never evaluated, ids < 128).  Dilation is with the disk of radius r (x^2 + y^2 <= r^2), out-of-image
// neighbours ignored (cv2.dilate's default border).
#include "common.h"

namespace uoc {

constexpr int EL = 128;  // label ids per map

__device__ __forceinline__ unsigned pack_add(unsigned w, int lab) {
  if (lab <= 0) return w;  // background is never a mask
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned cur = (w >> (8 * k)) & 0xFFu;
    if (cur == (unsigned)lab) return w;
    if (cur == 0xFFu) return (w & ~(0xFFu << (8 * k))) | ((unsigned)lab << (8 * k));
  }
  return w;
}

// grid-stride over pixels: contingency table (LDS-privatised per block) + boundary label packs of both maps
__global__ __launch_bounds__(1024) void eval_stats_kernel(const int *__restrict__ pred, const int *__restrict__ gt,
                                                          int H, int W, int *__restrict__ cont,
                                                          int *__restrict__ bnd_pred, int *__restrict__ bnd_gt,
                                                          unsigned *__restrict__ pk_pred, unsigned *__restrict__ pk_gt,
                                                          int *__restrict__ bad) {
  extern __shared__ int s_cont[];  // [EL][EL]
  for (int i = threadIdx.x; i < EL * EL; i += blockDim.x) s_cont[i] = 0;
  __syncthreads();
  const int n = H * W;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    const int y = p / W, x = p - y * W;
    const int lp = pred[p], lg = gt[p];
    if ((unsigned)lp >= (unsigned)EL || (unsigned)lg >= (unsigned)EL) {
      atomicOr(bad, 1);
      continue;
    }
    atomicAdd(&s_cont[lg * EL + lp], 1);
    const bool last_r = y == H - 1, last_c = x == W - 1;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const int *m = which ? gt : pred;
      const int c = which ? lg : lp;
      unsigned w = 0xFFFFFFFFu;
      if (!(last_r && last_c)) {
        const int e = last_c ? c : m[p + 1];          // last column: the east term is dropped (:55)
        const int s = last_r ? c : m[p + W];          // last row: the south term is dropped (:54)
        const int se = (last_r || last_c) ? c : m[p + W + 1];
        if (e != c || s != c || se != c) {
          w = pack_add(w, c);
          w = pack_add(w, e);
          w = pack_add(w, s);
          w = pack_add(w, se);
        }
      }
      (which ? pk_gt : pk_pred)[p] = w;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned l = (w >> (8 * k)) & 0xFFu;
        if (l != 0xFFu) atomicAdd((which ? bnd_gt : bnd_pred) + l, 1);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < EL * EL; i += blockDim.x)
    if (s_cont[i]) atomicAdd(&cont[i], s_cont[i]);
}

// For every pixel: labels of map A with a boundary here x labels of map B with a boundary within the disk.
__global__ __launch_bounds__(256) void eval_match_kernel(const unsigned *__restrict__ pk_pred,
                                                         const unsigned *__restrict__ pk_gt, int H, int W, int r,
                                                         int *__restrict__ prec_tp, int *__restrict__ rec_tp) {
  const int n = H * W;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    const unsigned here_p = pk_pred[p], here_g = pk_gt[p];
    if (here_p == 0xFFFFFFFFu && here_g == 0xFFFFFFFFu) continue;
    const int y = p / W, x = p - y * W;
    unsigned long long near_g[2] = {0ull, 0ull}, near_p[2] = {0ull, 0ull};
    for (int dy = -r; dy <= r; ++dy) {
      const int yy = y + dy;
      if ((unsigned)yy >= (unsigned)H) continue;
      for (int dx = -r; dx <= r; ++dx) {
        const int xx = x + dx;
        if ((unsigned)xx >= (unsigned)W || dx * dx + dy * dy > r * r) continue;
        const unsigned wg = pk_gt[yy * W + xx], wp = pk_pred[yy * W + xx];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned lg = (wg >> (8 * k)) & 0xFFu, lp = (wp >> (8 * k)) & 0xFFu;
          if (lg != 0xFFu) near_g[lg >> 6] |= 1ull << (lg & 63);
          if (lp != 0xFFu) near_p[lp >> 6] |= 1ull << (lp & 63);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned lp = (here_p >> (8 * k)) & 0xFFu;
      if (lp != 0xFFu)
        for (int h = 0; h < 2; ++h)
          for (unsigned long long m = near_g[h]; m; m &= m - 1)
            atomicAdd(&prec_tp[(64 * h + __ffsll((long long)m) - 1) * EL + lp], 1);
      const unsigned lg = (here_g >> (8 * k)) & 0xFFu;
      if (lg != 0xFFu)
        for (int h = 0; h < 2; ++h)
          for (unsigned long long m = near_p[h]; m; m &= m - 1)
            atomicAdd(&rec_tp[lg * EL + 64 * h + __ffsll((long long)m) - 1], 1);
    }
  }
}

}  // namespace uoc

using namespace uoc;

extern "C" {

size_t uoc_eval_workspace_bytes(int H, int W) {
  if (H < 1 || W < 1) return 0;
  return 2 * align_up((size_t)H * W * sizeof(unsigned), 256);
}

int uoc_eval_pair_stats(const int32_t *d_pred, const int32_t *d_gt, int H, int W, int radius, uoc_eval_tables *d_tables,
                        void *d_ws, size_t ws_bytes, void *stream) {
  UOC_REQUIRE(d_pred && d_gt && d_tables && d_ws, "null pointer");
  UOC_REQUIRE(H >= 1 && W >= 1 && radius >= 0 && radius <= 32, "H=%d W=%d radius=%d out of range", H, W, radius);
  UOC_REQUIRE(ws_bytes >= uoc_eval_workspace_bytes(H, W) && ((uintptr_t)d_ws & 255) == 0,
              "workspace too small or misaligned");
  hipStream_t st = (hipStream_t)stream;
  unsigned *pk_pred = (unsigned *)d_ws;
  unsigned *pk_gt = (unsigned *)((char *)d_ws + align_up((size_t)H * W * sizeof(unsigned), 256));
  UOC_HIP_CHECK(hipMemsetAsync(d_tables, 0, sizeof(uoc_eval_tables), st));
  const int n = H * W;
  int blocks = (n + 1023) / 1024;
  if (blocks > 64) blocks = 64;
  const size_t lds = (size_t)EL * EL * sizeof(int);
  static DeviceOnce attr_set;
  if (!attr_set.done()) {
    UOC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&eval_stats_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.mark();
  }
  hipLaunchKernelGGL(eval_stats_kernel, dim3(blocks), dim3(1024), lds, st, d_pred, d_gt, H, W, d_tables->cont,
                     d_tables->bnd_pred, d_tables->bnd_gt, pk_pred, pk_gt, &d_tables->bad_label);
  int mb = (n + 255) / 256;
  if (mb > 2048) mb = 2048;
  hipLaunchKernelGGL(eval_match_kernel, dim3(mb), dim3(256), 0, st, pk_pred, pk_gt, H, W, radius, d_tables->prec_tp,
                     d_tables->rec_tp);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

}  // extern "C"
