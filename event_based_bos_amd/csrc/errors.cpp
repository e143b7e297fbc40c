// errors.cpp -- thread-local error string + version/build info of libebos_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "ebos_hip.h"

namespace ebos {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace ebos

extern "C" {
int ebos_version(void) { return EBOS_ABI_VERSION; }
const char* ebos_last_error(void) { return ebos::g_err; }
const char* ebos_build_info(void) {
  return "libebos_hip gfx950 (CDNA4, wave64) hipcc " __VERSION__ " built " __DATE__;
}
}
