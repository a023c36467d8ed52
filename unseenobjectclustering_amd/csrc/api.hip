// libuoc_hip.so — error channel and version of the C ABI (include/uoc_hip.h).
#include "common.h"

#include <string.h>

namespace uoc {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace uoc

extern "C" {
int uoc_version(void) { return 100; }
const char *uoc_last_error(void) { return uoc::g_err; }
}
