// iwe_tiled_45x80x32.hip -- the tile-private pipeline's kernels and launchers for source tiles of 45 x 80 pixels with a 32 px largest
// window: see iwe_tiled_launch.h (one translation unit per built configuration; the C entry points are in iwe_tiled.hip).
#define EBOS_SLAB_OPS_UNIT
#include "iwe_tiled_launch.h"

namespace ebos {
EBOS_DEFINE_SLAB_OPS(45, 80, 32)
}  // namespace ebos

// diagnostic builds only (tools/build_stamps_lib.sh): the in-kernel stamps of THIS unit's kernels
extern "C" {
#ifdef EBOS_STAMPS
int ebos_debug_read_stamps(unsigned long long* host, int count) {  // diagnostic builds only
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ebos::g_stamps), sizeof(unsigned long long) * count);
}
int ebos_debug_read_stamps_bwd(unsigned long long* host, int count) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ebos::g_stamps_bwd), sizeof(unsigned long long) * count);
}
#endif
}
