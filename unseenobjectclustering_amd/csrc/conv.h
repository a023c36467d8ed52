// Internal interface of the conv / pooling / head kernels (csrc/conv.hip, csrc/head.hip).
#pragma once
#include "common.h"

namespace uoc {

// One convolution over G independent groups (= the RGB and the XYZ branch, which have their
// own weights).  Activations are NHWC fp32; weights are [G][taps][Cout][Kc] with BatchNorm
// already folded in (Kc = Cin, or 32 for the stem's 8-pixel x 4-channel rows).
struct ConvParams {
  const float *in;    // [G][B][H][W][Cin]
  const float *w;     // [G][T][Cout][Kc]
  const float *bias;  // [G][Cout]; nullptr = no bias (the Winograd F(4x4) plane GEMMs)
  const float *res;   // [G][B][Ho][Wo][Cout] or nullptr
  float *out;         // [G][B][Ho][Wo][Cout]
  int G, B, H, W, Cin, Ho, Wo, Cout;
  int KH, KW, stride, dil, pad, relu;
  int stem;           // 1: 7x7 s2 p3 conv over NHWC4 input, K-chunk = one kernel row
  int tune = 1;       // 0: never autotune for this call (generic C entry: the tuner re-launches into `out` and syncs)
  // profiling overrides (csrc/prof.h): a caller that uses this convolution as a building block books it under its own
  // kernel class, algorithmic flop count and shape tag
  int prof_kc = -1;
  double prof_flops = 0.0;
  int prof_tag[4] = {0, 0, 0, 0};
};

int launch_conv(const ConvParams &p, hipStream_t st);


// Winograd F(4x4,3x3) path (csrc/wino4.hip): U = transformed weights [G*36][Cout][Cin] (launch_wino4_weights, computed
// in double), ws = scratch of wino4_ws_floats() floats (the V and M frequency planes).
bool wino4_eligible(const ConvParams &p);
bool wino4_channels_ok(int Cin, int Cout);
size_t wino4_ws_floats(int G, int B, int H, int W, int d, int Cin, int Cout);
int launch_wino4_weights(const float *w, float *U, int G, int Cout, int Cin, hipStream_t st);
// U3 != nullptr: the split-precision ("bf16x3") plane GEMM of csrc/wino4_split.hip with the weights split by
// launch_wino4_split_weights ([G*36][3][Cout][Cin] bf16) — an experiment, never the default
int launch_wino4_conv(const ConvParams &p, const float *U, float *ws, hipStream_t st, const unsigned short *U3 = nullptr);
int launch_wino4_split_weights(const float *U, unsigned short *U3, int G, int Cout, int Cin, hipStream_t st);
int launch_wino4_slice_split(const ConvParams &p0, int Bg, int b0, const unsigned short *U3, float *ws, hipStream_t st);
struct Wino4Geom;
int w4_stage_output(const ConvParams &p, const Wino4Geom &geo, const float *Mw, hipStream_t st);
long wino4_elem_blocks(long items);
// NCHW [B][3][H][W] -> NHWC4 [B][H][W][4] (4th channel = 0)
int launch_nchw3_to_nhwc4(const float *in, float *out, int B, int H, int W, hipStream_t st);
// 3x3 s2 p1 max pooling, NHWC, C % 4 == 0; `n_img` images
int launch_maxpool3x3s2(const float *in, float *out, int n_img, int H, int W, int C, int Ho, int Wo, hipStream_t st);
// embed[b][p][:] = normalize(bilinear_up(a[b] + c[b]))  (align_corners=True), a,c: [B][h][w][64]; c may be null
// cat != 0: no sum; embed [B][2][H*W][64] = the two branches as planes, normalised over all 128 channels
int launch_head(const float *fa, const float *fb, float *embed, int B, int h, int w, int H, int W, int cat,
                hipStream_t st);

}  // namespace uoc
