// cmax_resident_32x64.hip -- the resident solver kernels of the patch-grid problem (one per contrast: variance, blurred variance, gradient
// magnitude) for source tiles of 32 x 64 pixels with a 32 px largest window  (720 x 640: the ROI of configs/hot_plate1.yaml): see cmax_resident_core.h.
// One translation unit per tile shape and problem: a kernel takes about a minute to compile, the units build side by side.
#include "cmax_resident_core.h"

namespace ebos {

int resident_launch_32x64(const ebos_cmax_patch_problem* q, int n_iter, void* mailbox, double spin_timeout_s, hipStream_t s) {
  return resident_patch_launch<32, 64, 32>(q, n_iter, mailbox, spin_timeout_s, s);
}

}  // namespace ebos
