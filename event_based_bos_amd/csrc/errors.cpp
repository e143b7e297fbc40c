// errors.cpp -- thread-local error string + version/build info of libebos_hip.so.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include <vector>

#include "ebos_hip.h"

namespace ebos {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- optional in-library kernel timing (bench.py's roofline leg) -------------------------------
// When enabled, the launcher of the dominant kernel attaches a HIP event pair to the dispatch itself
// (hipExtLaunchKernelGGL start/stop events on the stream it launches on).  Off by default: zero cost.
struct ProfileState {
  bool on = false;
  int which = 0;  // ebos_profile_kernel: which launcher's dispatches are stamped
  std::vector<hipEvent_t> ev;  // pairs
  int used = 0;
};
static ProfileState g_prof;

// Next free (start, stop) event pair, or false when profiling is off / the pairs are used up.  The launcher hands the
// pair to hipExtLaunchKernelGGL, which stamps the events with the dispatch's own begin / end timestamps.
bool profile_next_pair(hipEvent_t* start, hipEvent_t* stop, int which) {
  if (!g_prof.on || which != g_prof.which) return false;
  const int pairs = (int)g_prof.ev.size() / 2;
  if (g_prof.used >= pairs) return false;
  *start = g_prof.ev[2 * g_prof.used];
  *stop = g_prof.ev[2 * g_prof.used + 1];
  ++g_prof.used;
  return true;
}
}  // namespace ebos

extern "C" {
int ebos_profile_start(int max_records) { return ebos_profile_start_kernel(EBOS_PROFILE_SLAB_ACCUMULATE, max_records); }
int ebos_profile_start_kernel(int which, int max_records) {
  using namespace ebos;
  if (max_records <= 0 || g_prof.on || which < 0 || which > EBOS_PROFILE_GRADMAG_FUSED) {
    set_error("ebos_profile_start: bad kernel selector / max_records, or profiling already on");
    return EBOS_ERR_INVALID_ARG;
  }
  g_prof.which = which;
  g_prof.ev.resize(2 * (size_t)max_records);
  for (auto& e : g_prof.ev)
    if (hipEventCreate(&e) != hipSuccess) {
      set_error("ebos_profile_start: hipEventCreate failed");
      return EBOS_ERR_LAUNCH;
    }
  g_prof.used = 0;
  g_prof.on = true;
  return EBOS_OK;
}
int ebos_profile_stop(float* ms, int cap) {
  using namespace ebos;
  if (!g_prof.on) return 0;
  g_prof.on = false;
  int n = 0;
  for (int i = 0; i < g_prof.used; ++i) {
    float t = 0.f;
    if (hipEventSynchronize(g_prof.ev[2 * i + 1]) == hipSuccess &&
        hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) == hipSuccess && ms != nullptr && n < cap)
      ms[n++] = t;
  }
  for (auto& e : g_prof.ev) (void)hipEventDestroy(e);
  g_prof.ev.clear();
  g_prof.used = 0;
  return n;
}
int ebos_version(void) { return EBOS_ABI_VERSION; }
const char* ebos_last_error(void) { return ebos::g_err; }
const char* ebos_build_info(void) {
  return "libebos_hip gfx950 (CDNA4, wave64) hipcc " __VERSION__ " built " __DATE__;
}
}
