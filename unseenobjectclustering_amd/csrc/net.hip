// Two-branch RGB-D ResNet34-8s embedding network on gfx950: parameter intake (reference
// state-dict names), BatchNorm folding, weight re-layout, and the forward pass as a fixed
// sequence of hand-written kernels (csrc/conv.hip).
//
// Structure follows /root/reference/lib/networks:
//   SEG.py:69-71,105-108,113-114   two Resnet34_8s (fcn on BGR image, fcn_depth on XYZ), add, L2-normalise
//   resnet_dilated.py:287-327      resnet34(fully_conv, output_stride=8, no avgpool), fc=Conv2d(512,64,1)+bias,
//                                  upsample_bilinear to the input size (align_corners=True)
//   resnet.py:116-270              stem 7x7 s2 + BN + ReLU + maxpool; layers [3,4,6,3] of BasicBlocks;
//                                  :188-234 stride->dilation once the stride reaches 8 (layer3 d=2, layer4 d=4)
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "conv.h"

#include <stdlib.h>

namespace uoc {

struct ConvLayer {
  std::string conv, bn;  // parameter prefixes relative to "<branch>.resnet34_8s."
  int Cin, Cout, K, stride, dil, pad, relu;
  bool has_bn, has_bias, stem;
  float *d_w = nullptr, *d_b = nullptr;  // [G][...]
  float *d_U4 = nullptr;                 // Winograd F(4x4) weights [G*36][Cout][Cin] (eligible layers, wino_f == 4)
  unsigned short *d_U3 = nullptr;        // the same split into three bf16 planes (uoc_net_set_split_precision; csrc/wino4_split.hip)
  size_t w_per_group = 0;
};

struct Block {
  int conv1, conv2, down;  // indices into layers, down = -1 if identity shortcut
};

}  // namespace uoc

struct uoc_net {
  std::map<std::string, std::vector<float>> params;
  std::vector<uoc::ConvLayer> layers;
  std::vector<uoc::Block> blocks;
  int stem = -1, fc = -1;
  bool finalized = false;
  int device = -1;
  // Which layers run as Winograd convolutions is a compile-time rule (the algorithms round differently; no environment
  // variable may change a result).
  const int wino_min_cin = 64;   // 3x3 stride-1 layers with Cin >= this run as Winograd convolutions (round 4: 64, was 128)
  const int wino_f = 4;          // output tile of the Winograd path: F(4x4,3x3) (csrc/wino4.hip)
  bool split = false;      // EXPERIMENT: plane GEMMs in split precision (three bf16 terms per fp32 operand, fp32 accumulation)
  int mode = UOC_NET_RGBD_ADD;
  int G = 2;  // backbones evaluated side by side (2 for RGBD 'add' and 'cat')
};

namespace uoc {

static const char *kBranch[2] = {"fcn", "fcn_depth"};

static int add_conv(uoc_net *n, const std::string &conv, const std::string &bn, int Cin, int Cout, int K, int stride,
                    int dil, int relu, bool has_bn, bool has_bias, bool stem = false) {
  ConvLayer L;
  L.conv = conv;
  L.bn = bn;
  L.Cin = Cin;
  L.Cout = Cout;
  L.K = K;
  L.stride = stride;
  L.dil = dil;
  L.pad = stem ? 3 : (K == 3 ? dil : 0);
  L.relu = relu;
  L.has_bn = has_bn;
  L.has_bias = has_bias;
  L.stem = stem;
  n->layers.push_back(L);
  return (int)n->layers.size() - 1;
}

static void build_graph(uoc_net *n) {
  // early fusion feeds cat(img, xyz) to ONE backbone with a 6-channel stem (SEG.py:103-105,178-181)
  n->stem = add_conv(n, "conv1", "bn1", n->mode == UOC_NET_RGBD_EARLY ? 6 : 3, 64, 7, 2, 1, 1, true, false, true);
  const int nblocks[4] = {3, 4, 6, 3};
  const int planes[4] = {64, 128, 256, 512};
  int inpl = 64, cur_stride = 4, cur_dil = 1;
  for (int li = 0; li < 4; ++li) {
    int stride = li == 0 ? 1 : 2;
    const bool need_down = stride != 1 || inpl != planes[li];
    if (need_down) {
      if (cur_stride == 8) {  // output_stride reached: trade the stride for dilation (resnet.py:201-206)
        cur_dil *= stride;
        stride = 1;
      } else {
        cur_stride *= stride;
      }
    }
    for (int bi = 0; bi < nblocks[li]; ++bi) {
      const std::string pfx = "layer" + std::to_string(li + 1) + "." + std::to_string(bi) + ".";
      Block b;
      const int s = bi == 0 ? stride : 1;
      const int cin = bi == 0 ? inpl : planes[li];
      b.conv1 = add_conv(n, pfx + "conv1", pfx + "bn1", cin, planes[li], 3, s, cur_dil, 1, true, false);
      b.conv2 = add_conv(n, pfx + "conv2", pfx + "bn2", planes[li], planes[li], 3, 1, cur_dil, 1, true, false);
      b.down = -1;
      if (bi == 0 && need_down)
        b.down = add_conv(n, pfx + "downsample.0", pfx + "downsample.1", cin, planes[li], 1, s, 1, 0, true, false);
      n->blocks.push_back(b);
    }
    inpl = planes[li];
  }
  n->fc = add_conv(n, "fc", "", 512, 64, 1, 1, 1, 0, false, true);
}

static const std::vector<float> *find(const uoc_net *n, const std::string &key, size_t numel) {
  auto it = n->params.find(key);
  if (it == n->params.end()) {
    set_error("missing parameter '%s'", key.c_str());
    return nullptr;
  }
  if (it->second.size() != numel) {
    set_error("parameter '%s' has %zu elements, expected %zu", key.c_str(), it->second.size(), numel);
    return nullptr;
  }
  return &it->second;
}

static int finalize_layer(uoc_net *n, ConvLayer &L) {
  const int G = n->G;
  const int T = L.stem ? 7 : L.K * L.K;
  const int Kc = L.stem ? 32 : L.Cin;
  // The stem kernel consumes 4-channel pixels (3 used).  A 6-channel stem is stored as two 3-channel
  // weight sets [set][kh][cout][kw(8)][ch(4)] and evaluated as conv(xyz, set 1) then conv(img, set 0) + that.
  const int sets = L.stem ? L.Cin / 3 : 1;
  L.w_per_group = (size_t)T * L.Cout * Kc;
  // one extra all-zero bias row (used by the partial stem pass)
  std::vector<float> hw((size_t)G * sets * L.w_per_group, 0.f), hb((size_t)(G + 1) * L.Cout, 0.f);
  for (int g = 0; g < G; ++g) {
    const std::string base = std::string(kBranch[g]) + ".resnet34_8s.";
    const auto *w = find(n, base + L.conv + ".weight", (size_t)L.Cout * L.Cin * L.K * L.K);
    if (!w) return UOC_ENOENT;
    std::vector<double> scale(L.Cout, 1.0), shift(L.Cout, 0.0);
    if (L.has_bn) {
      const auto *ga = find(n, base + L.bn + ".weight", L.Cout);
      const auto *be = find(n, base + L.bn + ".bias", L.Cout);
      const auto *mu = find(n, base + L.bn + ".running_mean", L.Cout);
      const auto *va = find(n, base + L.bn + ".running_var", L.Cout);
      if (!ga || !be || !mu || !va) return UOC_ENOENT;
      for (int c = 0; c < L.Cout; ++c) {
        const double inv = 1.0 / sqrt((double)(*va)[c] + 1e-5);  // BatchNorm2d eps (resnet.py:143)
        scale[c] = (double)(*ga)[c] * inv;
        shift[c] = (double)(*be)[c] - (double)(*mu)[c] * scale[c];
      }
    }
    if (L.has_bias) {
      const auto *bi = find(n, base + L.conv + ".bias", L.Cout);
      if (!bi) return UOC_ENOENT;
      for (int c = 0; c < L.Cout; ++c) shift[c] += (double)(*bi)[c];
    }
    float *dst = hw.data() + (size_t)g * sets * L.w_per_group;
    for (int co = 0; co < L.Cout; ++co)
      for (int ci = 0; ci < L.Cin; ++ci)
        for (int kh = 0; kh < L.K; ++kh)
          for (int kw = 0; kw < L.K; ++kw) {
            const double v = (double)(*w)[(((size_t)co * L.Cin + ci) * L.K + kh) * L.K + kw] * scale[co];
            size_t o;
            if (L.stem)  // [set][kh][cout][kw(8)][ch(4)]
              o = (size_t)(ci / 3) * L.w_per_group + ((size_t)kh * L.Cout + co) * 32 + kw * 4 + ci % 3;
            else  // [tap][cout][cin]
              o = ((size_t)(kh * L.K + kw) * L.Cout + co) * L.Cin + ci;
            dst[o] = (float)v;
          }
    for (int c = 0; c < L.Cout; ++c) hb[(size_t)g * L.Cout + c] = (float)shift[c];
  }
  UOC_HIP_CHECK(hipMalloc(&L.d_w, hw.size() * sizeof(float)));
  UOC_HIP_CHECK(hipMalloc(&L.d_b, hb.size() * sizeof(float)));
  UOC_HIP_CHECK(hipMemcpy(L.d_w, hw.data(), hw.size() * sizeof(float), hipMemcpyHostToDevice));
  UOC_HIP_CHECK(hipMemcpy(L.d_b, hb.data(), hb.size() * sizeof(float), hipMemcpyHostToDevice));
  // Which layers run as Winograd is a STATIC rule (Cin >= wino_min_cin), not autotuned: the two algorithms round
  // differently, and results must not depend on timing.  F(4x4): every 3x3 stride-1 layer from 64 channels up since
  // round 4 (layer1 too: with the non-temporal frequency-plane accesses its transforms stopped costing more than the 4x
  // fewer MFMAs save — same-box A/B 165.3 / 165.9 -> 167.4 / 167.6 frames/s; round 3 had measured 160.3 vs 160.2).
  if (!L.stem && L.K == 3 && L.stride == 1 && n->wino_min_cin > 0 && L.Cin >= n->wino_min_cin && L.Cin % 32 == 0 &&
      L.Cout % 64 == 0) {
    if (n->wino_f == 4 && !wino4_channels_ok(L.Cin, L.Cout)) {
      // the direct kernel runs this layer (run_conv asks wino4_eligible again)
    } else if (n->wino_f == 4) {
      UOC_HIP_CHECK(hipMalloc(&L.d_U4, (size_t)G * 36 * L.Cout * L.Cin * sizeof(float)));
      if (int rc = launch_wino4_weights(L.d_w, L.d_U4, G, L.Cout, L.Cin, nullptr)) return rc;
    }
    UOC_HIP_CHECK(hipDeviceSynchronize());
  }
  return UOC_OK;
}

struct Dims {
  int H1, W1, H2, W2, H3, W3;
};
static Dims dims(int H, int W) {
  Dims d;
  d.H1 = (H - 1) / 2 + 1;  // conv 7x7 s2 p3
  d.W1 = (W - 1) / 2 + 1;
  d.H2 = (d.H1 - 1) / 2 + 1;  // maxpool 3x3 s2 p1
  d.W2 = (d.W1 - 1) / 2 + 1;
  d.H3 = (d.H2 - 1) / 2 + 1;  // layer2 stride 2
  d.W3 = (d.W2 - 1) / 2 + 1;
  return d;
}

struct NetWs {
  float *in4, *stem, *stem_part, *buf[4], *fc, *wino;
  size_t total;
};
static NetWs carve_net(void *base, int mode, int wino_f, int B, int H, int W) {
  const Dims d = dims(H, W);
  const int G = (mode == UOC_NET_RGBD_ADD || mode == UOC_NET_RGBD_CAT) ? 2 : 1;
  const int n_in = (G == 2 || mode == UOC_NET_RGBD_EARLY) ? 2 : 1;
  NetWs w;
  size_t off = 0;
  auto take = [&](size_t floats) {
    float *p = base ? (float *)((char *)base + off) : nullptr;
    off += align_up(floats * sizeof(float), 256);
    return p;
  };
  w.in4 = take((size_t)n_in * B * H * W * 4);
  w.stem = take((size_t)G * B * d.H1 * d.W1 * 64);
  w.stem_part = mode == UOC_NET_RGBD_EARLY ? take((size_t)B * d.H1 * d.W1 * 64) : nullptr;
  size_t act = (size_t)G * B * d.H2 * d.W2 * 64;
  const size_t a3 = (size_t)G * B * d.H3 * d.W3 * 512;
  if (a3 > act) act = a3;
  for (int i = 0; i < 4; ++i) w.buf[i] = take(act);
  w.fc = take((size_t)G * B * d.H3 * d.W3 * 64);
  // Winograd scratch: worst case over the layers that may use it (1/8 resolution, dilation 2 with 256 channels or
  // dilation 4 with 512 channels; also sized for 1/4 resolution x 64)
  size_t wv = 0;
  if (wino_f == 4) {  // F(4x4): V and M frequency planes [G*36][tiles][Cin + Cout]
    const int cand[6][4] = {{3, 4, 512, 512}, {3, 4, 256, 512}, {3, 2, 256, 256}, {3, 2, 128, 256}, {3, 1, 128, 128}, {2, 1, 64, 64}};
    for (const auto &c : cand) {
      const size_t v = wino4_ws_floats(G, B, c[0] == 3 ? d.H3 : d.H2, c[0] == 3 ? d.W3 : d.W2, c[1], c[2], c[3]);
      if (v > wv) wv = v;
    }
  }
  w.wino = take(wv);
  w.total = off;
  return w;
}

static ConvParams conv_params(int G, const ConvLayer &L, const float *in, const float *res, float *out, int B, int H, int W,
                              int Ho, int Wo) {
  ConvParams p;
  p.in = in;
  p.w = L.d_w;
  p.bias = L.d_b;
  p.res = res;
  p.out = out;
  p.G = G;
  p.B = B;
  p.H = H;
  p.W = W;
  p.Cin = L.stem ? 4 : L.Cin;
  p.Ho = Ho;
  p.Wo = Wo;
  p.Cout = L.Cout;
  p.KH = p.KW = L.K;
  p.stride = L.stride;
  p.dil = L.dil;
  p.pad = L.pad;
  p.relu = L.relu;
  p.stem = L.stem ? 1 : 0;
  return p;
}

static int run_conv(int G, const ConvLayer &L, const float *in, const float *res, float *out, int B, int H, int W,
                    int Ho, int Wo, hipStream_t st, float *wino_ws = nullptr, bool split = false) {
  const ConvParams p = conv_params(G, L, in, res, out, B, H, W, Ho, Wo);
  if (L.d_U4 && wino_ws && wino4_eligible(p))
    return launch_wino4_conv(p, L.d_U4, wino_ws, st, split ? L.d_U3 : nullptr);
  return launch_conv(p, st);
}

}  // namespace uoc

using namespace uoc;

extern "C" {

int uoc_net_create(uoc_net **out) { return uoc_net_create_mode(out, UOC_NET_RGBD_ADD); }

int uoc_net_create_mode(uoc_net **out, int mode) {
  UOC_REQUIRE(out != nullptr, "out is null");
  UOC_REQUIRE(mode >= UOC_NET_RGBD_ADD && mode <= UOC_NET_RGBD_CAT, "unknown network mode %d", mode);
  uoc_net *n = new (std::nothrow) uoc_net();
  if (!n) {
    set_error("out of host memory");
    return UOC_ENOMEM;
  }
  n->mode = mode;
  n->G = (mode == UOC_NET_RGBD_ADD || mode == UOC_NET_RGBD_CAT) ? 2 : 1;
  build_graph(n);
  *out = n;
  return UOC_OK;
}

int uoc_net_set_split_precision(uoc_net *n, int on) {
  UOC_REQUIRE(n && n->finalized, "uoc_net_set_split_precision: the network must be finalized");
  if (on) {
    for (auto &L : n->layers) {
      if (!L.d_U4 || L.d_U3) continue;
      const size_t elems = (size_t)n->G * 36 * 3 * L.Cout * L.Cin;
      UOC_HIP_CHECK(hipMalloc(&L.d_U3, elems * sizeof(unsigned short)));
      if (int rc = launch_wino4_split_weights(L.d_U4, L.d_U3, n->G, L.Cout, L.Cin, nullptr)) return rc;
    }
    UOC_HIP_CHECK(hipDeviceSynchronize());
  }
  n->split = on != 0;
  return UOC_OK;
}

int uoc_net_embed_dim(const uoc_net *n) { return n && n->mode == UOC_NET_RGBD_CAT ? 128 : 64; }

int uoc_net_destroy(uoc_net *n) {
  if (!n) return UOC_OK;
  for (auto &L : n->layers) {
    if (L.d_w) (void)hipFree(L.d_w);
    if (L.d_b) (void)hipFree(L.d_b);
    if (L.d_U4) (void)hipFree(L.d_U4);
    if (L.d_U3) (void)hipFree(L.d_U3);
  }
  delete n;
  return UOC_OK;
}

int uoc_net_load_param(uoc_net *n, const char *name, const float *host, size_t numel) {
  UOC_REQUIRE(n && name && host, "null argument");
  UOC_REQUIRE(!n->finalized, "network already finalized");
  n->params[std::string(name)] = std::vector<float>(host, host + numel);
  return UOC_OK;
}

int uoc_net_finalize(uoc_net *n) {
  UOC_REQUIRE(n != nullptr, "net is null");
  UOC_REQUIRE(!n->finalized, "network already finalized");
  UOC_HIP_CHECK(hipGetDevice(&n->device));
  for (auto &L : n->layers)
    if (int rc = finalize_layer(n, L)) return rc;
  n->params.clear();
  n->finalized = true;
  return UOC_OK;
}

size_t uoc_net_workspace_bytes(const uoc_net *n, int B, int H, int W) {
  if (!n || B < 1 || H < 8 || W < 8) return 0;
  return carve_net(nullptr, n->mode, n->wino_f, B, H, W).total;
}

int uoc_net_forward(uoc_net *n, const float *d_rgb, const float *d_xyz, int B, int H, int W, float *d_embed,
                    void *d_ws, size_t ws_bytes, void *stream) {
  UOC_REQUIRE(n && n->finalized, "network not finalized");
  UOC_REQUIRE(d_embed && (d_rgb || n->mode == UOC_NET_DEPTH) && (d_xyz || n->mode == UOC_NET_COLOR),
              "null tensor pointer");
  UOC_REQUIRE(B >= 1 && H >= 8 && W >= 8, "bad input shape B=%d H=%d W=%d", B, H, W);
  const int G = n->G;
  const NetWs w = carve_net(d_ws, n->mode, n->wino_f, B, H, W);
  UOC_REQUIRE(d_ws && ws_bytes >= w.total && ((uintptr_t)d_ws & 255) == 0, "workspace too small or misaligned (%zu < %zu)",
              ws_bytes, w.total);
  hipStream_t st = (hipStream_t)stream;
  const Dims d = dims(H, W);

  const ConvLayer &stem = n->layers[n->stem];
  const size_t in_stride = (size_t)B * H * W * 4;
  if (n->mode == UOC_NET_RGBD_EARLY) {
    // 6-channel stem as two 3-channel passes: XYZ part first (no bias, no ReLU), then the image part
    // with the folded-BN bias, the first pass as its residual, and the ReLU.
    if (int rc = launch_nchw3_to_nhwc4(d_rgb, w.in4, B, H, W, st)) return rc;
    if (int rc = launch_nchw3_to_nhwc4(d_xyz, w.in4 + in_stride, B, H, W, st)) return rc;
    ConvLayer part = stem;
    part.d_w = stem.d_w + stem.w_per_group;
    part.d_b = stem.d_b + (size_t)G * stem.Cout;  // the all-zero row
    part.relu = 0;
    if (int rc = run_conv(1, part, w.in4 + in_stride, nullptr, w.stem_part, B, H, W, d.H1, d.W1, st)) return rc;
    if (int rc = run_conv(1, stem, w.in4, w.stem_part, w.stem, B, H, W, d.H1, d.W1, st)) return rc;
  } else {
    // inputs -> NHWC4; RGBD 'add': group 0 = BGR image, group 1 = XYZ
    const float *first = n->mode == UOC_NET_DEPTH ? d_xyz : d_rgb;
    if (int rc = launch_nchw3_to_nhwc4(first, w.in4, B, H, W, st)) return rc;
    if (G == 2)
      if (int rc = launch_nchw3_to_nhwc4(d_xyz, w.in4 + in_stride, B, H, W, st)) return rc;
    if (int rc = run_conv(G, stem, w.in4, nullptr, w.stem, B, H, W, d.H1, d.W1, st)) return rc;
  }
  if (int rc = launch_maxpool3x3s2(w.stem, w.buf[0], G * B, d.H1, d.W1, 64, d.H2, d.W2, st)) return rc;

  int cur = 0, h = d.H2, wd = d.W2;
  for (const Block &b : n->blocks) {
    const ConvLayer &c1 = n->layers[b.conv1], &c2 = n->layers[b.conv2];
    const int ho = (h - 1) / c1.stride + 1, wo = (wd - 1) / c1.stride + 1;
    float *x = w.buf[cur], *tmp = w.buf[(cur + 1) & 3], *sc = w.buf[(cur + 2) & 3], *y = w.buf[(cur + 3) & 3];
    if (int rc = run_conv(G, c1, x, nullptr, tmp, B, h, wd, ho, wo, st, w.wino, n->split)) return rc;
    const float *res = x;
    if (b.down >= 0) {   // 1x1 (possibly strided) shortcut + BN (resnet.py:215-219)
      if (int rc = run_conv(G, n->layers[b.down], x, nullptr, sc, B, h, wd, ho, wo, st)) return rc;
      res = sc;
    }
    if (int rc = run_conv(G, c2, tmp, res, y, B, ho, wo, ho, wo, st, w.wino, n->split)) return rc;
    cur = (cur + 3) & 3;
    h = ho;
    wd = wo;
  }
  if (int rc = run_conv(G, n->layers[n->fc], w.buf[cur], nullptr, w.fc, B, h, wd, h, wd, st)) return rc;
  return launch_head(w.fc, G == 2 ? w.fc + (size_t)B * h * wd * 64 : nullptr, d_embed, B, h, wd, H, W,
                     n->mode == UOC_NET_RGBD_CAT, st);
}

/* Generic NHWC convolution entry (unit tests / integration): G independent groups stacked on the
 * leading dimension; per group: weights [T][Cout][Cin] with BN folded.  `algo` names the algorithm; nothing in the
 * environment decides it. */
int uoc_conv2d_nhwc_algo(const float *d_in, const float *d_w, const float *d_bias, const float *d_res, float *d_out, int G,
                         int B, int H, int W, int Cin, int Cout, int K, int stride, int dil, int pad, int relu, int algo,
                         void *stream) {
  UOC_REQUIRE(K == 1 || K == 3, "K=%d (only 1 or 3)", K);
  UOC_REQUIRE(d_res == nullptr || d_res != d_out, "conv2d: the residual must not alias the output");
  ConvParams p;
  p.tune = 0;   // static tile choice: no timing launches into the caller's buffers, no host-side sync
  p.in = d_in;
  p.w = d_w;
  p.bias = d_bias;
  p.res = d_res;
  p.out = d_out;
  p.G = G;
  p.B = B;
  p.H = H;
  p.W = W;
  p.Cin = Cin;
  p.Cout = Cout;
  p.KH = p.KW = K;
  p.stride = stride;
  p.dil = dil;
  p.pad = pad;
  p.relu = relu;
  p.stem = 0;
  p.Ho = (H + 2 * pad - dil * (K - 1) - 1) / stride + 1;
  p.Wo = (W + 2 * pad - dil * (K - 1) - 1) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  if (algo == UOC_CONV_DIRECT) return launch_conv(p, st);
  // scratch owned by this entry (kept between calls, grown on demand; the network path gets its scratch from the caller):
  // one caller thread at a time, as the header says — the mutex makes a violation slow instead of wrong
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (algo == UOC_CONV_WINOGRAD4 || algo == UOC_CONV_WINOGRAD4_BF16X3) {
    UOC_REQUIRE(wino4_eligible(p), "conv2d: shape not eligible for Winograd F(4x4,3x3)");
    static float *U4 = nullptr, *ws4 = nullptr;
    static unsigned short *U3 = nullptr;
    static size_t ucap4 = 0, wcap4 = 0, ucap3 = 0;
    const size_t un = (size_t)G * 36 * Cout * Cin, wn = wino4_ws_floats(G, B, H, W, dil, Cin, Cout);
    if (un > ucap4) {
      if (U4) (void)hipFree(U4);
      UOC_HIP_CHECK(hipMalloc(&U4, un * sizeof(float)));
      ucap4 = un;
    }
    if (wn > wcap4) {
      if (ws4) (void)hipFree(ws4);
      UOC_HIP_CHECK(hipMalloc(&ws4, wn * sizeof(float)));
      wcap4 = wn;
    }
    // always re-transform: callers reuse device addresses with new weights
    if (int rc = launch_wino4_weights(d_w, U4, G, Cout, Cin, st)) return rc;
    if (algo == UOC_CONV_WINOGRAD4_BF16X3) {
      if (3 * un > ucap3) {
        if (U3) (void)hipFree(U3);
        UOC_HIP_CHECK(hipMalloc(&U3, 3 * un * sizeof(unsigned short)));
        ucap3 = 3 * un;
      }
      if (int rc = launch_wino4_split_weights(U4, U3, G, Cout, Cin, st)) return rc;
      return launch_wino4_conv(p, U4, ws4, st, U3);
    }
    return launch_wino4_conv(p, U4, ws4, st);
  }
  set_error("conv2d: unknown algorithm %d", algo);
  return UOC_EINVAL;
}

int uoc_conv2d_nhwc(const float *d_in, const float *d_w, const float *d_bias, const float *d_res, float *d_out, int G,
                    int B, int H, int W, int Cin, int Cout, int K, int stride, int dil, int pad, int relu,
                    void *stream) {
  return uoc_conv2d_nhwc_algo(d_in, d_w, d_bias, d_res, d_out, G, B, H, W, Cin, Cout, K, stride, dil, pad, relu,
                              UOC_CONV_DIRECT, stream);
}

}  // extern "C"
