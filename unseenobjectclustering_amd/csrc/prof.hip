#include "prof.h"

#include <stdio.h>
#include <string.h>

#include <vector>

#include "common.h"

namespace uoc {

bool g_prof_enabled = false;

static const char *kNames[KC_COUNT] = {
    "conv_mfma_160x128", "conv_mfma_80x128", "conv_mfma_160x64", "conv_mfma_80x64", "conv_stem",
    "conv_glds_160x128", "conv_glds_80x128", "conv_glds_160x64", "conv_glds_80x64",
    "wino_input",        "wino_gemm",
    "net_misc",          "head",             "fps_step",         "hc_iter",         "hc_finalize",
    "seed_cc",           "assign",           "relabel",          "roi",
    "wino4_input",       "wino4_gemm",       "wino4_output"};

struct Rec {
  int kc;
  hipEvent_t a, b;
  double flops, bytes;
  ProfTag tag;
};
static std::vector<Rec> g_recs;
static std::vector<hipEvent_t> g_pool;

static hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

void prof_begin(int kc, hipStream_t st, double flops, double bytes, const ProfTag &tag) {
  Rec r;
  r.kc = kc;
  r.a = get_event();
  r.b = get_event();
  r.flops = flops;
  r.bytes = bytes;
  r.tag = tag;
  (void)hipEventRecord(r.a, st);
  g_recs.push_back(r);
}

void prof_end(hipStream_t st) { (void)hipEventRecord(g_recs.back().b, st); }

}  // namespace uoc

using namespace uoc;

extern "C" {

int uoc_prof_enable(int on) {
  g_prof_enabled = on != 0;
  return UOC_OK;
}

int uoc_prof_reset(void) {
  for (auto &r : g_recs) {
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  return UOC_OK;
}

/* JSON array: one object per kernel class with launches > 0; classes whose launches carry a shape tag also list the
 * per-shape totals under "shapes" (tag = [rows, Cin, Cout, dilation] for the convolution classes). */
int uoc_prof_report(char *buf, size_t cap) {
  UOC_REQUIRE(buf && cap > 2, "bad buffer");
  struct Agg {
    int kc;
    ProfTag tag;
    double ms = 0, fl = 0, by = 0;
    long cnt = 0;
  };
  double ms[KC_COUNT] = {0}, fl[KC_COUNT] = {0}, by[KC_COUNT] = {0};
  long cnt[KC_COUNT] = {0};
  std::vector<Agg> shapes;
  for (auto &r : g_recs) {
    UOC_HIP_CHECK(hipEventSynchronize(r.b));
    float t = 0.f;
    UOC_HIP_CHECK(hipEventElapsedTime(&t, r.a, r.b));
    ms[r.kc] += t;
    fl[r.kc] += r.flops;
    by[r.kc] += r.bytes;
    cnt[r.kc] += 1;
    if (r.tag.v[0] | r.tag.v[1] | r.tag.v[2] | r.tag.v[3]) {
      Agg *a = nullptr;
      for (auto &s : shapes)
        if (s.kc == r.kc && !memcmp(s.tag.v, r.tag.v, sizeof(r.tag.v))) a = &s;
      if (!a) {
        shapes.push_back(Agg());
        a = &shapes.back();
        a->kc = r.kc;
        a->tag = r.tag;
      }
      a->ms += t;
      a->fl += r.flops;
      a->by += r.bytes;
      a->cnt += 1;
    }
  }
  size_t off = 0;
  off += snprintf(buf + off, cap - off, "[");
  bool first = true;
  for (int k = 0; k < KC_COUNT; ++k) {
    if (!cnt[k]) continue;
    if (off + 320 >= cap) break;
    off += snprintf(buf + off, cap - off,
                    "%s{\"kernel\":\"%s\",\"launches\":%ld,\"total_ms\":%.6f,\"flops\":%.6e,\"bytes\":%.6e,\"shapes\":[",
                    first ? "" : ",", kNames[k], cnt[k], ms[k], fl[k], by[k]);
    bool f2 = true;
    for (auto &s : shapes) {
      if (s.kc != k || off + 320 >= cap) continue;
      off += snprintf(buf + off, cap - off,
                      "%s{\"tag\":[%d,%d,%d,%d],\"launches\":%ld,\"total_ms\":%.6f,\"flops\":%.6e,\"bytes\":%.6e}",
                      f2 ? "" : ",", s.tag.v[0], s.tag.v[1], s.tag.v[2], s.tag.v[3], s.cnt, s.ms, s.fl, s.by);
      f2 = false;
    }
    off += snprintf(buf + off, cap - off, "]}");
    first = false;
  }
  snprintf(buf + off, cap - off, "]");
  return UOC_OK;
}

}  // extern "C"
