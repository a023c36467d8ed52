// Opt-in per-kernel-class timing with HIP events recorded on the launch stream.
// Off by default (zero overhead); bench.py switches it on for a dedicated profiled pass and
// reads the per-class launch count, total duration and algorithmic flops/bytes.
// Single-threaded use only.
#pragma once
#include <hip/hip_runtime.h>

namespace uoc {

enum KernelClass {
  KC_CONV_160x128 = 0,
  KC_CONV_80x128,
  KC_CONV_160x64,
  KC_CONV_80x64,
  KC_CONV_STEM,
  KC_GLDS_160x128,
  KC_GLDS_80x128,
  KC_GLDS_160x64,
  KC_GLDS_80x64,
  KC_WINO_INPUT,
  KC_WINO_GEMM,
  KC_NET_MISC,  // layout conversion, max-pool
  KC_HEAD,
  KC_FPS_STEP,
  KC_HC_ITER,
  KC_HC_FINALIZE,
  KC_SEED_CC,
  KC_ASSIGN,
  KC_RELABEL,
  KC_ROI,
  KC_WINO4_INPUT,
  KC_WINO4_GEMM,
  KC_WINO4_OUTPUT,
  KC_COUNT
};

// Optional launch-shape tag: the report also aggregates per (class, tag), so that one kernel class can be broken down
// by layer shape (convolutions: rows M of the GEMM, Cin, Cout, dilation).
struct ProfTag {
  int v[4] = {0, 0, 0, 0};
};

extern bool g_prof_enabled;
void prof_begin(int kc, hipStream_t st, double flops, double bytes, const ProfTag &tag);
void prof_end(hipStream_t st);

struct ProfScope {
  hipStream_t st;
  bool on;
  ProfScope(int kc, hipStream_t s, double flops, double bytes, const ProfTag &tag = ProfTag()) : st(s), on(g_prof_enabled) {
    if (on) prof_begin(kc, s, flops, bytes, tag);
  }
  ~ProfScope() {
    if (on) prof_end(st);
  }
};

}  // namespace uoc
