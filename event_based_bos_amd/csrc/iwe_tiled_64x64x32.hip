// iwe_tiled_64x64x32.hip -- the tile-private pipeline's kernels and launchers for source tiles of 64 x 64 pixels with a 32 px largest
// window: see iwe_tiled_launch.h (one translation unit per built configuration; the C entry points are in iwe_tiled.hip).
#define EBOS_SLAB_OPS_UNIT
#include "iwe_tiled_launch.h"

namespace ebos {
EBOS_DEFINE_SLAB_OPS(64, 64, 32)
}  // namespace ebos
