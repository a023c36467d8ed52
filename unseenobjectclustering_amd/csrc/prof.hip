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
    "seed_cc",           "assign",           "relabel",          "roi"};

struct Rec {
  int kc;
  hipEvent_t a, b;
  double flops, bytes;
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

void prof_begin(int kc, hipStream_t st, double flops, double bytes) {
  Rec r;
  r.kc = kc;
  r.a = get_event();
  r.b = get_event();
  r.flops = flops;
  r.bytes = bytes;
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

/* JSON array: one object per kernel class with launches > 0. */
int uoc_prof_report(char *buf, size_t cap) {
  UOC_REQUIRE(buf && cap > 2, "bad buffer");
  double ms[KC_COUNT] = {0}, fl[KC_COUNT] = {0}, by[KC_COUNT] = {0};
  long cnt[KC_COUNT] = {0};
  for (auto &r : g_recs) {
    UOC_HIP_CHECK(hipEventSynchronize(r.b));
    float t = 0.f;
    UOC_HIP_CHECK(hipEventElapsedTime(&t, r.a, r.b));
    ms[r.kc] += t;
    fl[r.kc] += r.flops;
    by[r.kc] += r.bytes;
    cnt[r.kc] += 1;
  }
  size_t off = 0;
  off += snprintf(buf + off, cap - off, "[");
  bool first = true;
  for (int k = 0; k < KC_COUNT; ++k) {
    if (!cnt[k]) continue;
    if (off + 256 >= cap) break;
    off += snprintf(buf + off, cap - off,
                    "%s{\"kernel\":\"%s\",\"launches\":%ld,\"total_ms\":%.6f,\"flops\":%.6e,\"bytes\":%.6e}",
                    first ? "" : ",", kNames[k], cnt[k], ms[k], fl[k], by[k]);
    first = false;
  }
  snprintf(buf + off, cap - off, "]");
  return UOC_OK;
}

}  // extern "C"
