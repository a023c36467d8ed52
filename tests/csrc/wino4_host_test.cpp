// Host-side check of the Winograd F(4x4,3x3) arithmetic (csrc/wino4_math.h): the same inline functions the HIP
// kernels call, driven by plain loops on the CPU, against a direct 3x3 convolution in double.
// Build + run: tests/test_wino4_host.py (hipcc, no GPU needed).  Exit code 0 = all cases within tolerance.
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

#include "../../unseenobjectclustering_amd/csrc/wino4_math.h"

using namespace uoc;

static unsigned long long s_rng = 88172645463325252ull;
static float frand() {  // xorshift, uniform in [-1, 1)
  s_rng ^= s_rng << 13;
  s_rng ^= s_rng >> 7;
  s_rng ^= s_rng << 17;
  return (float)((double)(s_rng >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}

template <int VEC>
static int run_case(int G, int B, int H, int W, int Cin, int Cout, int d, bool use_res, int relu) {
  const Wino4Geom geo = make_geom4(B, H, W, d);
  std::vector<float> in((size_t)G * B * H * W * Cin), w((size_t)G * 9 * Cout * Cin), bias((size_t)G * Cout),
      res((size_t)G * B * H * W * Cout), out((size_t)G * B * H * W * Cout, -1e30f);
  for (auto &v : in) v = frand();
  for (auto &v : w) v = frand() / sqrtf(9.f * Cin);
  for (auto &v : bias) v = frand();
  for (auto &v : res) v = frand();
  std::vector<float> U((size_t)G * 36 * Cout * Cin), V((size_t)G * 36 * geo.NT * Cin), M((size_t)G * 36 * geo.NT * Cout);
  for (int g = 0; g < G; ++g)
    for (int co = 0; co < Cout; ++co)
      for (int ci = 0; ci < Cin; ++ci) wino4_weight_body(w.data(), U.data(), G, Cout, Cin, g, co, ci);
  for (int g = 0; g < G; ++g)
    for (int tau = 0; tau < geo.NT; ++tau)
      for (int cv = 0; cv < Cin / VEC; ++cv) wino4_input_body<VEC>(in.data(), V.data(), geo, Cin, g, tau, cv);
  // the batched GEMM: M[g*36+xi][tile][cout] = sum_cin V[g*36+xi][tile][cin] * U[g*36+xi][cout][cin]  (fp32 accumulate)
  for (int gx = 0; gx < G * 36; ++gx)
    for (int tau = 0; tau < geo.NT; ++tau)
      for (int co = 0; co < Cout; ++co) {
        float acc = 0.f;
        const float *v = &V[((size_t)gx * geo.NT + tau) * Cin], *u = &U[((size_t)gx * Cout + co) * Cin];
        for (int ci = 0; ci < Cin; ++ci) acc = fmaf(v[ci], u[ci], acc);
        M[((size_t)gx * geo.NT + tau) * Cout + co] = acc;
      }
  for (int g = 0; g < G; ++g)
    for (int tau = 0; tau < geo.NT; ++tau)
      for (int cv = 0; cv < Cout / VEC; ++cv)
        wino4_output_body<VEC>(M.data(), bias.data(), use_res ? res.data() : nullptr, out.data(), geo, Cout, relu, g, tau, cv);
  double worst = 0.0, scale = 1.0;
  for (int g = 0; g < G; ++g)
    for (int b = 0; b < B; ++b)
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
          for (int co = 0; co < Cout; ++co) {
            double acc = bias[(size_t)g * Cout + co];
            for (int kh = 0; kh < 3; ++kh)
              for (int kw = 0; kw < 3; ++kw) {
                const int iy = y + (kh - 1) * d, ix = x + (kw - 1) * d;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                const float *a = &in[((((size_t)g * B + b) * H + iy) * W + ix) * Cin];
                const float *k = &w[(((size_t)g * 9 + kh * 3 + kw) * Cout + co) * Cin];
                for (int ci = 0; ci < Cin; ++ci) acc += (double)a[ci] * (double)k[ci];
              }
            const size_t o = ((((size_t)g * B + b) * H + y) * W + x) * Cout + co;
            if (use_res) acc += res[o];
            if (relu && acc < 0) acc = 0;
            const double e = fabs(acc - (double)out[o]);
            if (e > worst) worst = e;
            if (fabs(acc) > scale) scale = fabs(acc);
          }
  const bool ok = worst < 2e-5 * scale;
  printf("vec%d G%d B%d %dx%d %d->%d d%d res%d relu%d: tiles %d, max abs err %.3e (scale %.2f) %s\n", VEC, G, B, H, W, Cin, Cout, d,
         (int)use_res, relu, geo.NT, worst, scale, ok ? "ok" : "FAIL");
  return ok ? 0 : 1;
}

int main() {
  int bad = 0;
  bad += run_case<4>(1, 1, 8, 8, 8, 8, 1, false, 0);
  bad += run_case<4>(2, 1, 15, 20, 8, 12, 1, true, 1);    // partial tiles
  bad += run_case<2>(1, 2, 14, 14, 8, 8, 2, true, 1);     // 7x7 phase images
  bad += run_case<1>(1, 1, 15, 20, 16, 8, 4, false, 1);   // phases of 4x5 / 3x5 pixels
  bad += run_case<2>(2, 2, 13, 9, 4, 8, 2, false, 0);     // ragged
  bad += run_case<1>(1, 1, 3, 5, 4, 4, 4, true, 0);       // image smaller than the dilation
  return bad ? 1 : 0;
}
