// libuoc_hip.so — error channel and version of the C ABI (include/uoc_hip.h).
#include "common.h"

#include <stdlib.h>
#include <string.h>

namespace uoc {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

std::atomic<int> g_env_epoch{0};
int EnvInt::get() {
  const int e = g_env_epoch.load(std::memory_order_acquire);
  if (seen.load(std::memory_order_acquire) != e) {
    const char *s = getenv(name);
    val.store(s ? atoi(s) : def, std::memory_order_relaxed);
    seen.store(e, std::memory_order_release);
  }
  return val.load(std::memory_order_relaxed);
}
}  // namespace uoc

extern "C" {
int uoc_version(void) { return 102; }
int uoc_is_dev_build(void) {
  return 0;
}
/* What could make two ranks compute different bits: the library version, a development build, and (development builds
 * only) the values of the knobs that change rounding.  runner.run_sharded all-reduces it with the error flag. */
unsigned long long uoc_config_fingerprint(void) {
  unsigned long long h = 1469598103934665603ull;
  auto mix = [&](long long v) {
    for (int i = 0; i < 8; ++i) {
      h ^= (unsigned long long)(v >> (8 * i)) & 0xff;
      h *= 1099511628211ull;
    }
  };
  mix(uoc_version());
  mix(uoc_is_dev_build());
  return h;
}
int uoc_reload_env(void) {
  uoc::g_env_epoch.fetch_add(1, std::memory_order_acq_rel);
  return UOC_OK;
}
const char *uoc_last_error(void) { return uoc::g_err; }
}
