// cmax_resident_45x80.hip -- the resident solver kernels of the patch-grid problem (one per contrast: variance, blurred variance, gradient
// magnitude) for source tiles of 45 x 80 pixels with a 32 px largest window: see cmax_resident_core.h.
// One translation unit per tile shape and problem: a kernel takes about a minute to compile, the units build side by side.
#include "cmax_resident_core.h"

namespace ebos {

int resident_launch_45x80(const ebos_cmax_patch_problem* q, int n_iter, void* mailbox, double spin_timeout_s, hipStream_t s) {
  return resident_patch_launch<45, 80, 32>(q, n_iter, mailbox, spin_timeout_s, s);
}

}  // namespace ebos

#ifdef EBOS_STAMPS
extern "C" {
int ebos_debug_read_stamps_resident(unsigned long long* host, int count) {  // diagnostic builds only (the 45 x 80 kernels' stamps)
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ebos::g_rstamps), sizeof(unsigned long long) * count);
}
int ebos_debug_read_stamps_resident_bwd(unsigned long long* host, int count) {  // (this unit's copy of the shared backward stamps)
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ebos::g_stamps_bwd), sizeof(unsigned long long) * count);
}
}
#endif
