// iwe_tiled_32x32x32.hip -- the tile-private pipeline's kernels and launchers for source tiles of 32 x 32 pixels with a 32 px largest
// window: see iwe_tiled_launch.h (one translation unit per built configuration; the C entry points are in iwe_tiled.hip).
#define EBOS_SLAB_OPS_UNIT
#include "iwe_tiled_launch.h"

namespace ebos {
EBOS_DEFINE_SLAB_OPS(32, 32, 32)
}  // namespace ebos
