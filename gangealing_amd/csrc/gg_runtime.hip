// Error plumbing + build identification for the C ABI (include/gangealing_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/gangealing_hip.h"
#include "gg_common.h"

namespace gg {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code ? code : -1;
}

int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
  return 0;
}

}  // namespace gg

extern "C" int gg_abi_version(void) { return 1; }
extern "C" const char* gg_last_error(void) { return gg::g_err; }
extern "C" const char* gg_build_arch(void) { return "gfx950"; }
