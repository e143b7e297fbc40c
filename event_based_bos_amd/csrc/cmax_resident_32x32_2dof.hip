// cmax_resident_32x32_2dof.hip -- the resident solver kernels of the 2-DoF problem (plain / blurred variance x integer / fractional
// source coordinates) for source tiles of 32 x 32 pixels with a 32 px largest window: see cmax_resident_core.h.
#include "cmax_resident_core.h"

namespace ebos {

int resident_launch_2dof_32x32(const ebos_cmax_2dof_problem* q, float w_variance, int n_iter, void* mailbox, double spin_timeout_s,
                             hipStream_t s) {
  return resident_2dof_launch<32, 32, 32>(q, w_variance, n_iter, mailbox, spin_timeout_s, s);
}

}  // namespace ebos
