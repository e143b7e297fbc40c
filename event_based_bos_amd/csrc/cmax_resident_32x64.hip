// cmax_resident_32x64.hip -- the resident solver kernels (patch grid and 2-DoF) for source tiles of 32 x 64 pixels with a 32 px largest
// window: see cmax_resident_core.h.  One translation unit per tile shape: the kernel takes minutes to compile, the units build side by side.
#include "cmax_resident_core.h"

namespace ebos {

int resident_launch_32x64(const ebos_cmax_patch_problem* q, const ebos_cmax_2dof_problem* q2, float w_variance2, int n_iter, void* mailbox,
                          double spin_timeout_s, hipStream_t s) {
  return resident_tile_launch<32, 64, 32>(q, q2, w_variance2, n_iter, mailbox, spin_timeout_s, s);
}

}  // namespace ebos
