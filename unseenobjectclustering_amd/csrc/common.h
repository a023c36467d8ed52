// Shared helpers for libuoc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "../../include/uoc_hip.h"

namespace uoc {

void set_error(const char *fmt, ...);

#define UOC_HIP_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      ::uoc::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return UOC_EHIP;                                                                   \
    }                                                                                    \
  } while (0)

#define UOC_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      ::uoc::set_error(__VA_ARGS__);  \
      return UOC_EINVAL;              \
    }                                 \
  } while (0)

#define UOC_LAUNCH_CHECK() UOC_HIP_CHECK(hipGetLastError())

// Development knobs from the environment, read ONCE (not per launch): the value is cached until uoc_reload_env() bumps
// the epoch (tests / micro-benchmarks that switch a knob inside one process call that).
extern std::atomic<int> g_env_epoch;
struct EnvInt {
  const char *name;
  int def;
  std::atomic<int> seen{-1};
  std::atomic<int> val{0};
  EnvInt(const char *n, int d) : name(n), def(d) {}
  int get();
};

// The library has ONE implementation per step and no knob that selects a kernel or changes a rounding (the measured-and-
// rejected alternates of rounds 1-5 live in the git history, HISTORY.md names the commits).  What it still reads from
// the environment is speed-only and listed in INTEGRATION.md.

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// True while `st` is being captured into a hipGraph (fcn/graph_replay.py): nothing that synchronises, times or launches
// cooperatively may run then.
static inline bool stream_is_capturing(hipStream_t st) {
  hipStreamCaptureStatus s = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(st, &s) == hipSuccess && s != hipStreamCaptureStatusNone;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Per-device one-time state.  hipFuncSetAttribute (dynamic LDS above 64 KB) applies to the CURRENT device's copy of a
// kernel, and a process may drive several GPUs (the reference wraps its nets in DataParallel), so "done once" flags
// are kept per device ordinal, not per process.
constexpr int kMaxDevices = 64;
static inline int current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) d = 0;
  return d;
}
struct DeviceOnce {
  std::atomic<unsigned long long> bits{0};
  bool done() const { return (bits.load(std::memory_order_acquire) >> current_device()) & 1ull; }
  void mark() { bits.fetch_or(1ull << current_device(), std::memory_order_release); }
};
// Compute-unit count of the current device (cached per device).
static inline int device_num_cu() {
  static std::atomic<int> cached[kMaxDevices];
  const int d = current_device();
  int v = cached[d].load(std::memory_order_relaxed);
  if (v == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d) != hipSuccess) return 0;
    v = prop.multiProcessorCount;
    cached[d].store(v, std::memory_order_relaxed);
  }
  return v;
}

// ---- wave64 cross-lane helpers (DPP; one VALU op each, no LDS) -------------------------
// Row = 16 lanes.  quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141,
// row_mirror = 0x140.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
// Sum over each 16-lane row; every lane of the row ends with the bitwise-identical total.
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);
  v += dpp_f<0x140>(v);
  return v;
}

}  // namespace uoc
