// Two-stage glue on the device: depth-coverage filter, per-object ROI boxes, crop+resize to
// 224x224, crop-cluster matching statistics and paste-back — batched over ROIs, no per-object
// host round trips.
//
// Replaces /root/reference/lib/fcn/test_dataset.py:
//   filter_labels_depth :183-198   -> label_stats_kernel + roi_build_kernel + apply_lut_kernel
//   crop_rois           :62-112    -> (same stats pass) + roi_crop_kernel   (mask.py:180-187 tight box)
//   match_label_crop    :116-179   -> crop_stats_kernel + crop_meanz_kernel (+ host ordering) + paste_kernel
//
// Interpolation index arithmetic follows ATen's float formulas so that crops are the ones the
// reference's F.upsample_bilinear (align_corners=True) / F.upsample_nearest produce:
//   bilinear: src = dst * (in-1)/(out-1), i0 = (int)src, lambda = src - i0
//   nearest : src = min((int)floorf(dst * (float)in/out), in-1)
#include "common.h"

#include <limits.h>
#include <math.h>

namespace uoc {

constexpr int NL = UOC_MAX_SEEDS;  // label ids are < 128

// stats layout per label: [cnt, zpos, maxx+1, W-minx, maxy+1, H-miny]  (all "max" accumulators, zero = empty)
constexpr int NSTAT = 6;

__global__ __launch_bounds__(256) void label_stats_kernel(const int *__restrict__ labels, const float *__restrict__ z,
                                                          int H, int W, int *__restrict__ stats) {
  __shared__ int s[NL * NSTAT];
  for (int i = threadIdx.x; i < NL * NSTAT; i += blockDim.x) s[i] = 0;
  __syncthreads();
  const int n = H * W;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    const int l = labels[p];
    if ((unsigned)l >= (unsigned)NL) continue;
    const int y = p / W, x = p - y * W;
    atomicAdd(&s[l * NSTAT + 0], 1);
    if (z && z[p] > 0.f) atomicAdd(&s[l * NSTAT + 1], 1);
    atomicMax(&s[l * NSTAT + 2], x + 1);
    atomicMax(&s[l * NSTAT + 3], W - x);
    atomicMax(&s[l * NSTAT + 4], y + 1);
    atomicMax(&s[l * NSTAT + 5], H - y);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NL * NSTAT; i += blockDim.x) {
    const int v = s[i];
    if (v == 0) continue;
    if (i % NSTAT < 2)
      atomicAdd(&stats[i], v);
    else
      atomicMax(&stats[i], v);
  }
}

// One block of 128 threads: thread l owns label l.
__global__ __launch_bounds__(NL) void roi_build_kernel(const int *__restrict__ stats, int use_filter, float thr,
                                                       float pad_frac, int H, int W, int *__restrict__ lut,
                                                       uoc_roi_table *__restrict__ table) {
  __shared__ int alive[NL];
  const int l = threadIdx.x;
  const int cnt = stats[l * NSTAT + 0], zpos = stats[l * NSTAT + 1];
  bool present = cnt > 0;
  bool filtered = false;
  if (use_filter && present && l != 0) {
    // torch.sum(roi_depth > 0).float() / torch.sum(mask)  <  threshold   (test_dataset.py:194-196)
    filtered = ((float)zpos / (float)cnt) < thr;
  }
  lut[l] = filtered ? 0 : l;
  alive[l] = (present && !filtered && l != 0) ? 1 : 0;
  __syncthreads();
  if (!table) return;
  int rank = 0;
  for (int i = 0; i < l; ++i) rank += alive[i];
  if (alive[l]) {
    const int x1 = stats[l * NSTAT + 2] - 1, x0 = W - stats[l * NSTAT + 3];
    const int y1 = stats[l * NSTAT + 4] - 1, y0 = H - stats[l * NSTAT + 5];
    // int(torch.round((x_max - x_min).float() * 0.25))  — round half to even (:83-84)
    const int px = (int)rintf((float)(x1 - x0) * pad_frac);
    const int py = (int)rintf((float)(y1 - y0) * pad_frac);
    table->label[rank] = l;
    table->box[rank][0] = max(x0 - px, 0);
    table->box[rank][1] = max(y0 - py, 0);
    table->box[rank][2] = min(x1 + px, W - 1);
    table->box[rank][3] = min(y1 + py, H - 1);
  }
  if (l == NL - 1) table->K = rank + alive[l];
}

__global__ __launch_bounds__(256) void apply_lut_kernel(int *__restrict__ labels, int n, const int *__restrict__ lut) {
  __shared__ int s[NL];
  if (threadIdx.x < NL) s[threadIdx.x] = lut[threadIdx.x];
  __syncthreads();
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    const int l = labels[p];
    if ((unsigned)l < (unsigned)NL) labels[p] = s[l];
  }
}

// grid (ceil(S*S/256), K).  rgb/xyz: [3][H][W] planes; outputs NCHW crops [K][3][S][S], mask [K][S][S].
__global__ __launch_bounds__(256) void roi_crop_kernel(const float *__restrict__ rgb, const float *__restrict__ xyz,
                                                       const int *__restrict__ labels, int H, int W,
                                                       const uoc_roi_table *__restrict__ table, int S,
                                                       float *__restrict__ out_rgb, float *__restrict__ out_xyz,
                                                       float *__restrict__ out_mask) {
  const int k = blockIdx.y;
  const int x0 = table->box[k][0], y0 = table->box[k][1], x1 = table->box[k][2], y1 = table->box[k][3];
  const int lab = table->label[k];
  const int cw = x1 - x0 + 1, ch = y1 - y0 + 1;
  const float sy = S > 1 ? (float)(ch - 1) / (float)(S - 1) : 0.f;
  const float sx = S > 1 ? (float)(cw - 1) / (float)(S - 1) : 0.f;
  const float ny = (float)ch / (float)S, nx = (float)cw / (float)S;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S * S) return;
  const int oy = i / S, ox = i - oy * S;
  // bilinear, align_corners=True (:104,109)
  const float fy = sy * (float)oy, fx = sx * (float)ox;
  int iy0 = (int)fy, ix0 = (int)fx;
  if (iy0 > ch - 1) iy0 = ch - 1;
  if (ix0 > cw - 1) ix0 = cw - 1;
  const int iy1 = iy0 + (iy0 < ch - 1 ? 1 : 0), ix1 = ix0 + (ix0 < cw - 1 ? 1 : 0);
  const float ly = fminf(fmaxf(fy - (float)iy0, 0.f), 1.f), lx = fminf(fmaxf(fx - (float)ix0, 0.f), 1.f);
  const float hy = 1.f - ly, hx = 1.f - lx;
  const size_t o00 = (size_t)(y0 + iy0) * W + (x0 + ix0), o01 = (size_t)(y0 + iy0) * W + (x0 + ix1);
  const size_t o10 = (size_t)(y0 + iy1) * W + (x0 + ix0), o11 = (size_t)(y0 + iy1) * W + (x0 + ix1);
  const size_t HW = (size_t)H * W, SS = (size_t)S * S;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float *a = rgb + c * HW;
    out_rgb[((size_t)k * 3 + c) * SS + i] = (a[o00] * hx + a[o01] * lx) * hy + (a[o10] * hx + a[o11] * lx) * ly;
    if (xyz) {  // COLOR input has no XYZ planes (crop_rois :73-76)
      const float *d = xyz + c * HW;
      out_xyz[((size_t)k * 3 + c) * SS + i] = (d[o00] * hx + d[o01] * lx) * hy + (d[o10] * hx + d[o11] * lx) * ly;
    }
  }
  // nearest (:106)
  int my = (int)floorf((float)oy * ny), mx = (int)floorf((float)ox * nx);
  if (my > ch - 1) my = ch - 1;
  if (mx > cw - 1) mx = cw - 1;
  out_mask[(size_t)k * SS + i] = labels[(size_t)(y0 + my) * W + (x0 + mx)] == lab ? 1.f : 0.f;
}

// per ROI / per crop-cluster id: pixel count and overlap with the stage-1 mask (:118-125)
__global__ __launch_bounds__(256) void crop_stats_kernel(const int *__restrict__ labels_crop,
                                                         const float *__restrict__ mask_crop, int SS,
                                                         int *__restrict__ cnt, int *__restrict__ ov) {
  __shared__ int sc[NL], so[NL];
  const int k = blockIdx.y;
  if (threadIdx.x < NL) {
    sc[threadIdx.x] = 0;
    so[threadIdx.x] = 0;
  }
  __syncthreads();
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < SS; i += gridDim.x * blockDim.x) {
    const int l = labels_crop[(size_t)k * SS + i];
    if ((unsigned)l >= (unsigned)NL) continue;
    atomicAdd(&sc[l], 1);
    if (mask_crop[(size_t)k * SS + i] != 0.f) atomicAdd(&so[l], 1);
  }
  __syncthreads();
  if (threadIdx.x < NL) {
    if (sc[threadIdx.x]) atomicAdd(&cnt[k * NL + threadIdx.x], sc[threadIdx.x]);
    if (so[threadIdx.x]) atomicAdd(&ov[k * NL + threadIdx.x], so[threadIdx.x]);
  }
}

// keep[k][c] = cluster c of ROI k overlaps its stage-1 mask by >= 50 %; mean z of kept pixels
// with z > 0 (all pixels if nothing is kept) (:129-136).  One block per ROI.
__global__ __launch_bounds__(1024) void crop_meanz_kernel(const int *__restrict__ labels_crop,
                                                          const float *__restrict__ xyz_crop, int SS,
                                                          const int *__restrict__ cnt, const int *__restrict__ ov,
                                                          int *__restrict__ keep, float *__restrict__ meanz) {
  __shared__ int sk[NL];
  __shared__ int s_any;
  __shared__ double rs[1024 / 64];
  __shared__ int rc[1024 / 64];
  const int k = blockIdx.x;
  if (threadIdx.x == 0) s_any = 0;
  __syncthreads();
  if (threadIdx.x < NL) {
    const int c = cnt[k * NL + threadIdx.x], o = ov[k * NL + threadIdx.x];
    const int kp = (c > 0 && !(((float)o / (float)c) < 0.5f)) ? 1 : 0;
    sk[threadIdx.x] = kp;
    keep[k * NL + threadIdx.x] = kp;
    if (kp) atomicOr(&s_any, 1);
  }
  __syncthreads();
  if (!xyz_crop) return;  // no depth: the host orders ROIs by box area instead (:138-146)
  const int any = s_any;
  const float *z = xyz_crop + ((size_t)k * 3 + 2) * SS;
  const int *lab = labels_crop + (size_t)k * SS;
  double sum = 0.0;
  int n = 0;
  // Seven elements of a thread's strided walk are loaded together and then added IN ORDER (the same additions as an
  // element-at-a-time loop: bit-identical mean) — one block per ROI walks 49 elements per thread at 224x224, and with two
  // dependent loads per element the loop was 49 serial memory round trips (44 us; 153 us beside other streams' kernels).
  constexpr int UN = 7;
  for (int i0 = threadIdx.x; i0 < SS; i0 += UN * blockDim.x) {
    int l[UN];
    float v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int i = i0 + u * blockDim.x;
      l[u] = i < SS ? lab[i] : -1;
      v[u] = i < SS ? z[i] : 0.f;          // 0 is never selected (v > 0)
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const bool sel = any ? ((unsigned)l[u] < (unsigned)NL && sk[l[u]]) : true;
      if (sel && v[u] > 0.f) {
        sum += (double)v[u];
        ++n;
      }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    sum += __shfl_xor(sum, off);
    n += __shfl_xor(n, off);
  }
  if ((threadIdx.x & 63) == 0) {
    rs[threadIdx.x >> 6] = sum;
    rc[threadIdx.x >> 6] = n;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s2 = 0.0;
    int n2 = 0;
    for (int w = 0; w < 1024 / 64; ++w) {
      s2 += rs[w];
      n2 += rc[w];
    }
    meanz[k] = (float)(s2 / (double)n2);  // 0/0 -> NaN like torch.mean of an empty tensor
  }
}

// ROI paint order and global renumbering on the device (match_label_crop :129-163) — round 6: replaces the host's
// sorted() between a device->host read of the statistics and a host->device upload of the plan.
//   key[k]   = mean depth of ROI k (:129-136), or its box area as float32 without depth (:138-146)
//   order    = sorted(range(K), key=key, reverse=True): Python's stable sort — equal keys keep their index order.
//              Without NaN keys `<` is a strict weak order and every stable sort gives the same list: a rank sort.  With
//              NaN keys (`torch.mean` of an empty selection, :135) every comparison is false and the result is whatever
//              CPython's list.sort does; for K < 64 that is reverse, count_run, binary insertion sort, reverse
//              (Objects/listobject.c, minrun = n below 64), restated here step by step.  NaN keys AND K >= 64 (merges with
//              galloping) raise bit 0 of *status instead and the caller orders on the host.
//   map[k][c] = running count over the kept clusters c of the ROIs in paint order (:156-163), 0 = dropped.
// One block of 128 threads.
struct KeyIdx {
  float key;
  int idx;
};
__device__ __forceinline__ bool key_lt(const KeyIdx &a, const KeyIdx &b) { return a.key < b.key; }   // the 0-dim tensors' `<`

__device__ void cpython_sort_small(KeyIdx *a, int n) {   // list.sort(reverse=True) for n < 64, comparisons `<` only
  auto rev = [&](int lo, int hi) {   // reverse a[lo:hi]
    for (--hi; lo < hi; ++lo, --hi) {
      const KeyIdx t = a[lo];
      a[lo] = a[hi];
      a[hi] = t;
    }
  };
  rev(0, n);
  if (n >= 2) {
    int run = 2, lo;   // count_run
    const bool desc = key_lt(a[1], a[0]);
    if (desc) {
      for (lo = 2; lo < n; ++lo, ++run)
        if (!key_lt(a[lo], a[lo - 1])) break;
    } else {
      for (lo = 2; lo < n; ++lo, ++run)
        if (key_lt(a[lo], a[lo - 1])) break;
    }
    if (desc) rev(0, run);
    for (int start = run; start < n; ++start) {   // binarysort(lo = 0, hi = n, start = run)
      int l = 0, r = start;
      const KeyIdx pivot = a[start];
      do {
        const int p = l + ((r - l) >> 1);
        if (key_lt(pivot, a[p]))
          r = p;
        else
          l = p + 1;
      } while (l < r);
      for (int p = start; p > l; --p) a[p] = a[p - 1];
      a[l] = pivot;
    }
  }
  rev(0, n);
}

__global__ __launch_bounds__(NL) void roi_order_kernel(const int *__restrict__ keep, const float *__restrict__ meanz,
                                                       const uoc_roi_table *__restrict__ table, int K,
                                                       int *__restrict__ order, int *__restrict__ map,
                                                       int *__restrict__ status) {
  __shared__ KeyIdx a[NL];
  __shared__ int ord[NL], nk[NL], base[NL];
  __shared__ int s_nan;
  const int t = threadIdx.x;
  if (t == 0) s_nan = 0;
  __syncthreads();
  float key = 0.f;
  if (t < K) {
    if (meanz) {
      key = meanz[t];
    } else {  // roi_size = (y_max - y_min + 1) * (x_max - x_min + 1) on the float32 rois (:139-146)
      const float x0 = (float)table->box[t][0], y0 = (float)table->box[t][1];
      const float x1 = (float)table->box[t][2], y1 = (float)table->box[t][3];
      key = __fmul_rn(y1 - y0 + 1.f, x1 - x0 + 1.f);
    }
    a[t].key = key;
    a[t].idx = t;
    if (key != key) atomicOr(&s_nan, 1);
    int c = 0;
    for (int l = 0; l < NL; ++l) c += keep[t * NL + l] != 0 ? 1 : 0;
    nk[t] = c;
  }
  __syncthreads();
  if (!s_nan) {
    if (t < K) {
      int pos = 0;
      for (int j = 0; j < K; ++j) {
        const float kj = a[j].key;
        pos += (kj > key || (kj == key && j < t)) ? 1 : 0;
      }
      ord[pos] = t;
    }
  } else if (t == 0) {
    if (K < 64) {
      cpython_sort_small(a, K);
      for (int j = 0; j < K; ++j) ord[j] = a[j].idx;
    } else {
      if (status) atomicOr(status, 1);
      for (int j = 0; j < K; ++j) ord[j] = j;
    }
  }
  __syncthreads();
  if (t == 0) {
    int running = 0;
    for (int j = 0; j < K; ++j) {
      const int i = ord[j];
      base[i] = running;
      running += nk[i];
      order[j] = i;
    }
  }
  __syncthreads();
  // thread t = crop cluster id t: its rank among the kept clusters of ROI i (ascending id, like torch.unique :156)
  __shared__ int wave0[NL];
  for (int i = 0; i < K; ++i) {
    const int kp = keep[i * NL + t] != 0 ? 1 : 0;
    const unsigned long long m = __ballot(kp);
    const int lane = t & 63;
    const int below = __popcll(m & ((1ull << lane) - 1ull));
    if (t == 0) wave0[i] = __popcll(m);   // kept clusters with id < 64
    __syncthreads();
    const int rank = below + (t >= 64 ? wave0[i] : 0);
    map[i * NL + t] = kp ? base[i] + rank + 1 : 0;
  }
}

// refined[p] = relabelled crop cluster of the LAST ROI (in paint order) covering p with a kept
// cluster; 0 otherwise (:165-177; nearest resize back to the ROI size).
__global__ __launch_bounds__(256) void paste_kernel(const int *__restrict__ labels_crop,
                                                    const uoc_roi_table *__restrict__ table,
                                                    const int *__restrict__ map, const int *__restrict__ order, int K,
                                                    int S, int H, int W, int *__restrict__ refined) {
  const int n = H * W, SS = S * S;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    const int y = p / W, x = p - y * W;
    int val = 0;
    for (int j = K - 1; j >= 0; --j) {
      const int k = order[j];
      const int x0 = table->box[k][0], y0 = table->box[k][1], x1 = table->box[k][2], y1 = table->box[k][3];
      if (x < x0 || x > x1 || y < y0 || y > y1) continue;
      const int w = x1 - x0 + 1, h = y1 - y0 + 1;
      int sy = (int)floorf((float)(y - y0) * ((float)S / (float)h));
      int sx = (int)floorf((float)(x - x0) * ((float)S / (float)w));
      if (sy > S - 1) sy = S - 1;
      if (sx > S - 1) sx = S - 1;
      const int l = labels_crop[(size_t)k * SS + sy * S + sx];
      const int v = ((unsigned)l < (unsigned)NL) ? map[k * NL + l] : 0;
      if (v != 0) {
        val = v;
        break;
      }
    }
    refined[p] = val;
  }
}

// Input preparation on the device (tools/test_images.py:96-133): uint8 BGR [H][W][3] + uint16 depth in
// millimetres -> the two float NCHW tensors the path consumes.  Same float32 operations, in the same
// order, as the reference's numpy/torch code (true divisions, no FMA contraction possible).
__global__ __launch_bounds__(256) void prep_rgbd_kernel(const unsigned char *__restrict__ bgr,
                                                        const unsigned short *__restrict__ depth_mm, int H, int W,
                                                        float fx, float fy, float px, float py, float m0, float m1,
                                                        float m2, float *__restrict__ image, float *__restrict__ xyz) {
  const int n = H * W;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    const int v = p / W, u = p - v * W;
    image[p] = (float)bgr[3 * p + 0] / 255.0f - m0;          // im / 255.0 - PIXEL_MEANS / 255.0  (:125-127)
    image[n + p] = (float)bgr[3 * p + 1] / 255.0f - m1;
    image[2 * n + p] = (float)bgr[3 * p + 2] / 255.0f - m2;
    const float z = (float)depth_mm[p] / 1000.0f;            // :113
    xyz[p] = ((float)u - px) * z / fx;                       // :99
    xyz[n + p] = ((float)v - py) * z / fy;                   // :100
    xyz[2 * n + p] = z;
  }
}

struct RoiWs {
  int *stats;  // [128][6]
  int *lut;    // [128]
  int *cnt;    // [127][128]
  int *ov;     // [127][128]
  int *keep;   // [127][128]   (uoc_roi_match)
  float *meanz;  // [128]
  int *plan;   // [128] paint order + [127][128] id map
  size_t total;
};
static RoiWs carve_roi(void *base) {
  RoiWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    void *p = base ? (void *)((char *)base + off) : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  w.stats = (int *)take(NL * NSTAT * sizeof(int));
  w.lut = (int *)take(NL * sizeof(int));
  w.cnt = (int *)take((size_t)NL * NL * sizeof(int));
  w.ov = (int *)take((size_t)NL * NL * sizeof(int));     // directly behind cnt (the statistics' single fill relies on it)
  w.keep = (int *)take((size_t)NL * NL * sizeof(int));
  w.meanz = (float *)take(NL * sizeof(float));
  w.plan = (int *)take((size_t)(NL + NL * NL) * sizeof(int));
  w.total = off;
  return w;
}

static int grid_for(int n) {
  int b = (n + 255) / 256;
  if (b > 1024) b = 1024;
  return b < 1 ? 1 : b;
}

}  // namespace uoc

using namespace uoc;

// ---- label map -> uint8 block row + running maximum (the frame-parallel runner's per-frame output step) ----------
__global__ __launch_bounds__(256) void labels_to_u8_kernel(const int32_t *__restrict__ lab, long n, uint8_t *__restrict__ out,
                                                           int32_t *__restrict__ top) {
  int mx = 0;
  const long n4 = n >> 2;   // four labels per thread: one 16-byte load, one 4-byte store (both pointers are 16-byte aligned rows)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int4 v = reinterpret_cast<const int4 *>(lab)[i];
    mx = max(max(mx, max(v.x, v.y)), max(v.z, v.w));
    reinterpret_cast<uchar4 *>(out)[i] = make_uchar4((uint8_t)v.x, (uint8_t)v.y, (uint8_t)v.z, (uint8_t)v.w);
  }
  for (long i = 4 * n4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int v = lab[i];
    mx = max(mx, v);
    out[i] = (uint8_t)v;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mx = max(mx, __shfl_xor(mx, off));
  __shared__ int s_mx[4];            // one atomic per block (4 096 same-address atomics cost ~35 us, measured)
  if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3]));
    if (mx > 0) atomicMax(top, mx);
  }
}

extern "C" {

size_t uoc_roi_workspace_bytes(void) { return carve_roi(nullptr).total; }

int uoc_prep_rgbd(const uint8_t *d_bgr, const uint16_t *d_depth_mm, int H, int W, float fx, float fy, float px,
                  float py, float mean_b, float mean_g, float mean_r, float *d_image, float *d_xyz, void *stream) {
  UOC_REQUIRE(d_bgr && d_depth_mm && d_image && d_xyz, "null pointer");
  UOC_REQUIRE(H >= 1 && W >= 1, "bad shape");
  hipLaunchKernelGGL(prep_rgbd_kernel, dim3(grid_for(H * W)), dim3(256), 0, (hipStream_t)stream, d_bgr, d_depth_mm, H, W,
                     fx, fy, px, py, mean_b, mean_g, mean_r, d_image, d_xyz);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

int uoc_filter_labels_depth(int32_t *d_labels, const float *d_z, long z_batch_stride, int B, int H, int W,
                            float threshold, void *d_ws, size_t ws_bytes, void *stream) {
  UOC_REQUIRE(d_labels && d_z && d_ws, "null pointer");
  UOC_REQUIRE(B >= 1 && H >= 1 && W >= 1, "bad shape");
  RoiWs w = carve_roi(d_ws);
  UOC_REQUIRE(ws_bytes >= w.total, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int n = H * W;
  for (int b = 0; b < B; ++b) {
    int *lab = d_labels + (size_t)b * n;
    UOC_HIP_CHECK(hipMemsetAsync(w.stats, 0, NL * NSTAT * sizeof(int), st));
    hipLaunchKernelGGL(label_stats_kernel, dim3(grid_for(n)), dim3(256), 0, st, lab, d_z + (size_t)b * z_batch_stride, H,
                       W, w.stats);
    hipLaunchKernelGGL(roi_build_kernel, dim3(1), dim3(NL), 0, st, w.stats, 1, threshold, 0.f, H, W, w.lut,
                       (uoc_roi_table *)nullptr);
    hipLaunchKernelGGL(apply_lut_kernel, dim3(grid_for(n)), dim3(256), 0, st, lab, n, w.lut);
  }
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

int uoc_roi_build(int32_t *d_labels, const float *d_z, int H, int W, float threshold, float pad_fraction,
                  uoc_roi_table *d_table, void *d_ws, size_t ws_bytes, void *stream) {
  UOC_REQUIRE(d_labels && d_table && d_ws, "null pointer");
  RoiWs w = carve_roi(d_ws);
  UOC_REQUIRE(ws_bytes >= w.total, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int n = H * W;
  UOC_HIP_CHECK(hipMemsetAsync(w.stats, 0, NL * NSTAT * sizeof(int), st));
  UOC_HIP_CHECK(hipMemsetAsync(d_table, 0, sizeof(uoc_roi_table), st));
  hipLaunchKernelGGL(label_stats_kernel, dim3(grid_for(n)), dim3(256), 0, st, d_labels, d_z, H, W, w.stats);
  hipLaunchKernelGGL(roi_build_kernel, dim3(1), dim3(NL), 0, st, w.stats, d_z ? 1 : 0, threshold, pad_fraction, H, W,
                     w.lut, d_table);
  if (d_z) hipLaunchKernelGGL(apply_lut_kernel, dim3(grid_for(n)), dim3(256), 0, st, d_labels, n, w.lut);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

int uoc_roi_crop(const float *d_rgb, const float *d_xyz, const int32_t *d_labels, int H, int W,
                 const uoc_roi_table *d_table, int K, int S, float *d_rgb_crops, float *d_xyz_crops,
                 float *d_mask_crops, void *stream) {
  UOC_REQUIRE(d_rgb && d_labels && d_table && d_rgb_crops && d_mask_crops, "null pointer");
  UOC_REQUIRE((d_xyz == nullptr) == (d_xyz_crops == nullptr), "d_xyz and d_xyz_crops must be given together");
  UOC_REQUIRE(K >= 1 && K < NL && S >= 1, "K=%d S=%d out of range", K, S);
  hipLaunchKernelGGL(roi_crop_kernel, dim3((S * S + 255) / 256, K), dim3(256), 0, (hipStream_t)stream, d_rgb, d_xyz,
                     d_labels, H, W, d_table, S, d_rgb_crops, d_xyz_crops, d_mask_crops);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

int uoc_roi_match_stats(const int32_t *d_labels_crop, const float *d_mask_crops, const float *d_xyz_crops, int K,
                        int S, int32_t *d_keep, float *d_meanz, void *d_ws, size_t ws_bytes, void *stream) {
  UOC_REQUIRE(d_labels_crop && d_mask_crops && d_keep && d_ws, "null pointer");
  UOC_REQUIRE(d_xyz_crops == nullptr || d_meanz != nullptr, "d_meanz is null");
  UOC_REQUIRE(K >= 1 && K < NL && S >= 1, "K=%d S=%d out of range", K, S);
  RoiWs w = carve_roi(d_ws);
  UOC_REQUIRE(ws_bytes >= w.total, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  // cnt and ov are neighbours in the workspace (NL * NL ints each): one fill clears all of cnt and the K rows of ov in use
  UOC_HIP_CHECK(hipMemsetAsync(w.cnt, 0, ((size_t)NL * NL + (size_t)K * NL) * sizeof(int), st));
  int gb = (S * S + 255) / 256;
  if (gb > 64) gb = 64;
  hipLaunchKernelGGL(crop_stats_kernel, dim3(gb, K), dim3(256), 0, st, d_labels_crop, d_mask_crops, S * S, w.cnt, w.ov);
  hipLaunchKernelGGL(crop_meanz_kernel, dim3(K), dim3(1024), 0, st, d_labels_crop, d_xyz_crops, S * S, w.cnt, w.ov,
                     d_keep, d_meanz);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

int uoc_roi_paste(const int32_t *d_labels_crop, const uoc_roi_table *d_table, const int32_t *d_map,
                  const int32_t *d_order, int K, int S, int H, int W, int32_t *d_refined, void *stream) {
  UOC_REQUIRE(d_labels_crop && d_table && d_map && d_order && d_refined, "null pointer");
  UOC_REQUIRE(K >= 1 && K < NL, "K=%d out of range", K);
  hipLaunchKernelGGL(paste_kernel, dim3(grid_for(H * W)), dim3(256), 0, (hipStream_t)stream, d_labels_crop, d_table,
                     d_map, d_order, K, S, H, W, d_refined);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

int uoc_roi_match(const int32_t *d_labels_crop, const float *d_mask_crops, const float *d_xyz_crops,
                  const uoc_roi_table *d_table, int K, int S, int H, int W, int32_t *d_refined, int32_t *d_keep,
                  int32_t *d_plan, int32_t *d_status, void *d_ws, size_t ws_bytes, void *stream) {
  UOC_REQUIRE(d_labels_crop && d_mask_crops && d_table && d_refined && d_ws, "null pointer");
  UOC_REQUIRE(K >= 1 && K < NL && S >= 1 && H >= 1 && W >= 1, "K=%d S=%d out of range", K, S);
  RoiWs w = carve_roi(d_ws);
  UOC_REQUIRE(ws_bytes >= w.total, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  int *keep = d_keep ? d_keep : w.keep;
  int *plan = d_plan ? d_plan : w.plan;
  // cnt and ov are neighbours in the workspace (NL * NL ints each): one fill clears all of cnt and the K rows of ov in use
  UOC_HIP_CHECK(hipMemsetAsync(w.cnt, 0, ((size_t)NL * NL + (size_t)K * NL) * sizeof(int), st));
  int gb = (S * S + 255) / 256;
  if (gb > 64) gb = 64;
  hipLaunchKernelGGL(crop_stats_kernel, dim3(gb, K), dim3(256), 0, st, d_labels_crop, d_mask_crops, S * S, w.cnt, w.ov);
  hipLaunchKernelGGL(crop_meanz_kernel, dim3(K), dim3(1024), 0, st, d_labels_crop, d_xyz_crops, S * S, w.cnt, w.ov, keep,
                     w.meanz);
  hipLaunchKernelGGL(roi_order_kernel, dim3(1), dim3(NL), 0, st, keep, d_xyz_crops ? w.meanz : (const float *)nullptr,
                     d_table, K, plan, plan + K, d_status);
  hipLaunchKernelGGL(paste_kernel, dim3(grid_for(H * W)), dim3(256), 0, st, d_labels_crop, d_table, plan + K, plan, K, S,
                     H, W, d_refined);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

int uoc_labels_to_u8(const int32_t *d_labels, long n, uint8_t *d_out, int32_t *d_top, void *stream) {
  UOC_REQUIRE(d_labels && d_out && d_top && n >= 1, "null pointer / empty map");
  UOC_REQUIRE(((uintptr_t)d_labels & 15) == 0 && ((uintptr_t)d_out & 3) == 0, "labels_to_u8: misaligned pointer");
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(labels_to_u8_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, d_labels, n, d_out, d_top);
  UOC_LAUNCH_CHECK();
  return UOC_OK;
}

}  // extern "C"
