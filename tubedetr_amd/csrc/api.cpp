// Error plumbing of the C ABI (thread-local last-error string, launch checks).
#include <stdarg.h>

#include "td_common.h"

namespace td {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return TD_ERR_LAUNCH;
  }
  return TD_OK;
}

}  // namespace td

extern "C" const char* td_last_error(void) { return td::g_err; }
extern "C" int td_abi_version(void) { return 1; }
