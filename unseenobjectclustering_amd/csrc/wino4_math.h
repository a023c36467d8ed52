// Winograd F(4x4, 3x3) transforms: geometry and the three small matrix products, written as host + device inline
// functions so that tests/csrc/wino4_host_test.cpp can run the very same arithmetic on the CPU against a direct
// convolution (no GPU in the build container).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      4x4 outputs from a 6x6 input patch, 36 products per (cin, cout)
//                                              instead of the direct form's 144: 4x fewer matrix-core flops
//
// Interpolation points (0, 1, -1, 2, -1/2, inf).  The textbook set (0, +-1, +-2) and three others were compared on
// the CPU through the whole two-branch network (scripts/wino_f4_error.py, calibrated weights, 480x640): embeddings
// differ from an fp64 evaluation by 1.16e-8 on average with this set against 1.05e-8 for F(2x2,3x3), 1.25e-8 for the
// textbook F(4x4) set and ~1.0e-8 for the direct fp32 sum; the largest difference to the direct fp32 result is 8.8e-7
// (F(2x2): 6.6e-7).  B^T and A^T contain only binary fractions (exact multiplications, the additions round); G has
// thirds and fifteenths and is applied once, in double, when the weights are loaded.
#pragma once
#include <hip/hip_runtime.h>

namespace uoc {

struct Wino4Geom {
  int B, H, W, d, TH, TW, NT;  // TH x TW tiles of 4x4 outputs per (image, dilation phase); NT = B*d*d*TH*TW
  int Bg;                      // images per group in the activation tensors [g][Bg][H][W][C] (>= B: a launch may cover a slice of the batch)
};

__host__ __device__ inline Wino4Geom make_geom4(int B, int H, int W, int d) {
  Wino4Geom g;
  g.B = B;
  g.H = H;
  g.W = W;
  g.d = d;
  g.TH = ((H + d - 1) / d + 3) / 4;
  g.TW = ((W + d - 1) / d + 3) / 4;
  g.NT = B * d * d * g.TH * g.TW;
  g.Bg = B;
  return g;
}

// tile index -> image and the first output pixel of the tile; outputs are (oy + a*d, ox + e*d), a, e in 0..3, the
// input patch is (oy + (i-1)*d, ox + (j-1)*d), i, j in 0..5 (dilation by phase decomposition as in csrc/wino.hip)
__host__ __device__ inline void wino4_decode(int tau, const Wino4Geom &g, int &b, int &oy, int &ox) {
  const int tx = tau % g.TW;
  tau /= g.TW;
  const int ty = tau % g.TH;
  tau /= g.TH;
  const int px = tau % g.d;
  tau /= g.d;
  const int py = tau % g.d;
  b = tau / g.d;
  oy = py + 4 * ty * g.d;
  ox = px + 4 * tx * g.d;
}

// ---- scalar / float4 arithmetic used by the templates below ----------------------------------------------------
__host__ __device__ inline float w4_fma(float c, float a, float b) { return fmaf(c, a, b); }
__host__ __device__ inline float w4_add(float a, float b) { return a + b; }
__host__ __device__ inline float w4_sub(float a, float b) { return a - b; }
__host__ __device__ inline float w4_neg(float a) { return -a; }
__host__ __device__ inline float2 w4_fma(float c, float2 a, float2 b) { return make_float2(fmaf(c, a.x, b.x), fmaf(c, a.y, b.y)); }
__host__ __device__ inline float2 w4_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__host__ __device__ inline float2 w4_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__host__ __device__ inline float2 w4_neg(float2 a) { return make_float2(-a.x, -a.y); }
__host__ __device__ inline float4 w4_fma(float c, float4 a, float4 b) {
  return make_float4(fmaf(c, a.x, b.x), fmaf(c, a.y, b.y), fmaf(c, a.z, b.z), fmaf(c, a.w, b.w));
}
__host__ __device__ inline float4 w4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__host__ __device__ inline float4 w4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__host__ __device__ inline float4 w4_neg(float4 a) { return make_float4(-a.x, -a.y, -a.z, -a.w); }

// Streaming accesses of the frequency planes (each element is written once and read once, by another kernel): on the
// device they carry the non-temporal hint so that they do not displace other kernels' working sets from L2 (round 4,
// same-box A/B of two builds: 161.0 -> 162.9 frames/s sustained, input transform 36 -> 32 us, output 47 -> 45 us;
// -DUOC_W4_NO_NT restores plain accesses); on the host they are plain accesses.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(UOC_W4_NO_NT)
typedef float w4_nv2 __attribute__((ext_vector_type(2)));
typedef float w4_nv4 __attribute__((ext_vector_type(4)));
__device__ inline float w4_nt_load(const float *p) { return __builtin_nontemporal_load(p); }
__device__ inline float2 w4_nt_load(const float2 *p) {
  const w4_nv2 v = __builtin_nontemporal_load(reinterpret_cast<const w4_nv2 *>(p));
  return make_float2(v.x, v.y);
}
__device__ inline float4 w4_nt_load(const float4 *p) {
  const w4_nv4 v = __builtin_nontemporal_load(reinterpret_cast<const w4_nv4 *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ inline void w4_nt_store(float *p, float v) { __builtin_nontemporal_store(v, p); }
__device__ inline void w4_nt_store(float2 *p, float2 v) {
  w4_nv2 n = {v.x, v.y};
  __builtin_nontemporal_store(n, reinterpret_cast<w4_nv2 *>(p));
}
__device__ inline void w4_nt_store(float4 *p, float4 v) {
  w4_nv4 n = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(n, reinterpret_cast<w4_nv4 *>(p));
}
#define W4_STREAM_LOAD(T, ptr) w4_nt_load(reinterpret_cast<const T *>(ptr))
#define W4_STREAM_STORE(T, ptr, val) w4_nt_store(reinterpret_cast<T *>(ptr), (val))
#else
#define W4_STREAM_LOAD(T, ptr) (*reinterpret_cast<const T *>(ptr))
#define W4_STREAM_STORE(T, ptr, val) (*reinterpret_cast<T *>(ptr) = (val))
#endif

// channel vectors of 1, 2 or 4 floats: the elementwise kernels are instantiated for all three widths (fewer channels per
// thread = fewer registers per thread = waves that fit beside another stream's matrix kernel on the same SIMD)
template <int VEC> struct W4Vec;
template <> struct W4Vec<1> { typedef float type; };
template <> struct W4Vec<2> { typedef float2 type; };
template <> struct W4Vec<4> { typedef float4 type; };
__host__ __device__ inline void w4_zero(float &v) { v = 0.f; }
__host__ __device__ inline void w4_zero(float2 &v) { v = make_float2(0.f, 0.f); }
__host__ __device__ inline void w4_zero(float4 &v) { v = make_float4(0.f, 0.f, 0.f, 0.f); }
__host__ __device__ inline float w4_relu(float v) { return fmaxf(v, 0.f); }
__host__ __device__ inline float2 w4_relu(float2 v) { return make_float2(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f)); }
__host__ __device__ inline float4 w4_relu(float4 v) { return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)); }

// o = B^T i,   B^T = [[1, 1.5, -2, -1.5, 1, 0], [0, -1, -2.5, -0.5, 1, 0], [0, 1, 0.5, -2.5, 1, 0],
//                     [0, -0.5, -1, 0.5, 1, 0], [0, 2, -1, -2, 1, 0],      [0, 1, 1.5, -2, -1.5, 1]]
template <typename T>
__host__ __device__ inline void wino4_bt(const T (&i)[6], T (&o)[6]) {
  o[0] = w4_fma(-1.5f, i[3], w4_fma(-2.0f, i[2], w4_fma(1.5f, i[1], w4_add(i[0], i[4]))));
  o[1] = w4_fma(-0.5f, i[3], w4_fma(-2.5f, i[2], w4_sub(i[4], i[1])));
  o[2] = w4_fma(-2.5f, i[3], w4_fma(0.5f, i[2], w4_add(i[4], i[1])));
  o[3] = w4_fma(0.5f, i[3], w4_fma(-0.5f, i[1], w4_sub(i[4], i[2])));
  o[4] = w4_fma(-2.0f, i[3], w4_fma(2.0f, i[1], w4_sub(i[4], i[2])));
  o[5] = w4_fma(-1.5f, i[4], w4_fma(-2.0f, i[3], w4_fma(1.5f, i[2], w4_add(i[1], i[5]))));
}

// y = A^T m,   A^T = [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -0.5, 0], [0, 1, 1, 4, 0.25, 0], [0, 1, -1, 8, -0.125, 1]]
template <typename T>
__host__ __device__ inline void wino4_at(const T (&m)[6], T (&y)[4]) {
  const T s12 = w4_add(m[1], m[2]), d12 = w4_sub(m[1], m[2]);
  y[0] = w4_add(w4_add(m[0], s12), w4_add(m[3], m[4]));
  y[1] = w4_fma(-0.5f, m[4], w4_fma(2.0f, m[3], d12));
  y[2] = w4_fma(0.25f, m[4], w4_fma(4.0f, m[3], s12));
  y[3] = w4_add(w4_fma(-0.125f, m[4], w4_fma(8.0f, m[3], d12)), m[5]);
}

// u = G k (double),   G = [[1, 0, 0], [-1/3, -1/3, -1/3], [1/3, -1/3, 1/3], [1/15, 2/15, 4/15], [-16/15, 8/15, -4/15], [0, 0, 1]]
__host__ __device__ inline void wino4_g(const double (&k)[3], double (&u)[6]) {
  u[0] = k[0];
  u[1] = -(k[0] + k[1] + k[2]) / 3.0;
  u[2] = (k[0] - k[1] + k[2]) / 3.0;
  u[3] = (k[0] + 2.0 * k[1] + 4.0 * k[2]) / 15.0;
  u[4] = (-16.0 * k[0] + 8.0 * k[1] - 4.0 * k[2]) / 15.0;
  u[5] = k[2];
}

// ---- per-element bodies of the three elementwise kernels (one (group, tile / weight, channel quad) each) --------

// U[(g*36 + xi)][cout][cin] = (G g G^T)[xi] from w [g][9][cout][cin]; one (g, cout, cin) per call
__host__ __device__ inline void wino4_weight_body(const float *w, float *U, int G, int Cout, int Cin, int g, int co, int ci) {
  double t[6][3];
  for (int b = 0; b < 3; ++b) {
    const double col[3] = {(double)w[(((size_t)g * 9 + 0 + b) * Cout + co) * Cin + ci],
                           (double)w[(((size_t)g * 9 + 3 + b) * Cout + co) * Cin + ci],
                           (double)w[(((size_t)g * 9 + 6 + b) * Cout + co) * Cin + ci]};
    double u[6];
    wino4_g(col, u);
    for (int i = 0; i < 6; ++i) t[i][b] = u[i];
  }
  const size_t plane = (size_t)Cout * Cin;
  for (int i = 0; i < 6; ++i) {
    double u[6];
    wino4_g(t[i], u);
    for (int j = 0; j < 6; ++j) U[((size_t)g * 36 + 6 * i + j) * plane + (size_t)co * Cin + ci] = (float)u[j];
  }
  (void)G;
}

// v[6*i + j] = (B^T d B)[i][j] of one 6x6 patch d[row][col]  (columns first, then rows: the order every caller shares)
template <typename T>
__host__ __device__ inline void wino4_input_tile(const T (&d)[6][6], T (&v)[36]) {
  T t[6][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    T col[6], o[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) col[i] = d[i][j];
    wino4_bt(col, o);
#pragma unroll
    for (int i = 0; i < 6; ++i) t[i][j] = o[i];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    T o[6];
    wino4_bt(t[i], o);
#pragma unroll
    for (int j = 0; j < 6; ++j) v[6 * i + j] = o[j];
  }
}

// V[(g*36 + xi)][tile][cin] = (B^T d B)[xi] for channels VEC*cv .. VEC*cv+VEC-1 of tile tau; in: [g][Bg][H][W][C]
template <int VEC>
__host__ __device__ inline void wino4_input_body(const float *in, float *V, const Wino4Geom &geo, int C, int g, int tau, int cv) {
  typedef typename W4Vec<VEC>::type T;
  int b, oy, ox;
  wino4_decode(tau, geo, b, oy, ox);
  const float *src = in + (((size_t)g * geo.Bg + b) * geo.H * geo.W) * C + VEC * cv;
  T d[6][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int x = ox + (j - 1) * geo.d;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int y = oy + (i - 1) * geo.d;
      const bool ok = (unsigned)y < (unsigned)geo.H && (unsigned)x < (unsigned)geo.W;
      if (ok)
        d[i][j] = *reinterpret_cast<const T *>(src + ((size_t)y * geo.W + x) * C);
      else
        w4_zero(d[i][j]);
    }
  }
  T v[36];
  wino4_input_tile(d, v);
  const size_t plane = (size_t)geo.NT * C;
  float *dst = V + (size_t)g * 36 * plane + (size_t)tau * C + VEC * cv;
#pragma unroll
  for (int k = 0; k < 36; ++k) W4_STREAM_STORE(T, dst + (size_t)k * plane, v[k]);
}

// y[a][e] = (A^T m A)[a][e] of one tile's 36 frequencies m[6*i + j]  (columns first, then rows)
template <typename T>
__host__ __device__ inline void wino4_output_tile(const T (&m)[36], T (&y)[4][4]) {
  T s[4][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    T col[6], o[4];
#pragma unroll
    for (int i = 0; i < 6; ++i) col[i] = m[6 * i + j];
    wino4_at(col, o);
#pragma unroll
    for (int a = 0; a < 4; ++a) s[a][j] = o[a];
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) wino4_at(s[a], y[a]);
}

// out[g][b][y][x][cout] = relu?(A^T M A + bias (+ res)) for channels VEC*cv .. of tile tau; M: [(g*36 + xi)][tile][cout]
template <int VEC>
__host__ __device__ inline void wino4_output_body(const float *M, const float *bias, const float *res, float *out,
                                                  const Wino4Geom &geo, int Cout, int relu, int g, int tau, int cv) {
  typedef typename W4Vec<VEC>::type T;
  const size_t plane = (size_t)geo.NT * Cout;
  const float *src = M + (size_t)g * 36 * plane + (size_t)tau * Cout + VEC * cv;
  T m[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) m[k] = W4_STREAM_LOAD(T, src + (size_t)k * plane);
  int b, oy, ox;
  wino4_decode(tau, geo, b, oy, ox);
  const size_t gsz = (size_t)geo.Bg * geo.H * geo.W * Cout;
  const size_t o0 = (size_t)g * gsz + (((size_t)b * geo.H + oy) * geo.W + ox) * Cout + VEC * cv;
  const size_t oy_step = (size_t)geo.d * geo.W * Cout, ox_step = (size_t)geo.d * Cout;
  // The 16 residual values are loaded UP FRONT, next to the 36 frequencies (round 5).  Loaded where they are used — one load,
  // one wait, one store, sixteen times in a row, because the compiler may not move a load of `res` across a store to `out` — the
  // kernel paid sixteen dependent memory round trips per thread: 4.0-4.7 TB/s on the residual layers against 5.5 without.
#ifndef W4_OUT_RES_MODE
#define W4_OUT_RES_MODE 2   // 2 = residual loads after the column pass (fewest live registers; measured best), 1 = up front, 0 = at the stores (round 4)
#endif
  T rv[4][4];
#define W4_LOAD_RES()                                                                        \
  if (res) {                                                                                 \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) _Pragma("unroll") for (int e = 0; e < 4; ++e) { \
      if (oy + a * geo.d < geo.H && ox + e * geo.d < geo.W)                                  \
        rv[a][e] = *reinterpret_cast<const T *>(res + o0 + a * oy_step + e * ox_step);       \
      else                                                                                   \
        w4_zero(rv[a][e]);                                                                   \
    }                                                                                        \
  }
  if (W4_OUT_RES_MODE == 1) { W4_LOAD_RES() }
  T yv[4][4];
  {
    T s[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      T col[6], o[4];
#pragma unroll
      for (int i = 0; i < 6; ++i) col[i] = m[6 * i + j];
      wino4_at(col, o);
#pragma unroll
      for (int a = 0; a < 4; ++a) s[a][j] = o[a];
    }
    if (W4_OUT_RES_MODE == 2) { W4_LOAD_RES() }
#pragma unroll
    for (int a = 0; a < 4; ++a) wino4_at(s[a], yv[a]);
  }
#undef W4_LOAD_RES
  T bv;
  if (bias)
    bv = *reinterpret_cast<const T *>(bias + (size_t)g * Cout + VEC * cv);
  else
    w4_zero(bv);   // a null bias is zero, as in the direct kernels
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int y = oy + a * geo.d;
    if (y >= geo.H) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int x = ox + e * geo.d;
      if (x >= geo.W) continue;
      T v = w4_add(yv[a][e], bv);                 // + bias, + residual, ReLU — in this order
      if (res) v = w4_add(v, W4_OUT_RES_MODE == 0 ? *reinterpret_cast<const T *>(res + o0 + a * oy_step + e * ox_step) : rv[a][e]);
      if (relu) v = w4_relu(v);
      *reinterpret_cast<T *>(out + o0 + a * oy_step + e * ox_step) = v;
    }
  }
}

}  // namespace uoc
