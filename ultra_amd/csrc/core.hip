// Library-level plumbing: version, error text, launch check.
#include "common.hpp"

#include <string.h>

namespace ultr {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return ULTR_EHIP;
  }
  return ULTR_OK;
}

}  // namespace ultr

extern "C" int ultr_version(void) { return ULTR_VERSION; }
extern "C" const char* ultr_last_error(void) { return ultr::g_err; }
