// Error reporting and version of libsage_hip.so.
#include "common.h"

namespace sage {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace sage

extern "C" const char* sage_last_error(void) { return sage::g_err; }
extern "C" int sage_abi_version(void) { return 2; }
