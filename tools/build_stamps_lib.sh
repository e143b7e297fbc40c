#!/bin/bash
# Diagnostic twin of the product library: the 45 x 80 units (iwe_tiled_45x80x32.hip: the tile-private pipeline's kernels;
# cmax_resident_45x80.hip: the resident patch-grid kernels) compiled with -DEBOS_STAMPS (in-kernel phase stamps), every other object
# as built by `python -m event_based_bos_amd.build`.  -> event_based_bos_amd/lib/libebos_stamps.so (git-ignored)
# EBOS_STAMPS_MORE (default -DEBOS_STAMPS_EPI: slots 1, 2, 7 of the backward stamps take the sub-steps of the tile adjoint, for
# tools/stamp_resident.py): "" for tools/stamp_phases_bwd.py's table, -DEBOS_STAMPS_SETUP for its --setup table.
set -e
cd "$(dirname "$0")/.."
python -m event_based_bos_amd.build > /dev/null
mkdir -p /tmp/ebos_stamps
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -fPIC -fno-gpu-rdc -Wno-unused-function -Iinclude -Ievent_based_bos_amd/csrc -DEBOS_STAMPS ${EBOS_STAMPS_MORE--DEBOS_STAMPS_EPI}"
/opt/rocm/bin/hipcc $FLAGS -mllvm -sink-insts-to-avoid-spills=1 -x hip -c event_based_bos_amd/csrc/cmax_resident_45x80.hip -o /tmp/ebos_stamps/cmax_resident_45x80.o &
/opt/rocm/bin/hipcc $FLAGS -x hip -c event_based_bos_amd/csrc/iwe_tiled_45x80x32.hip -o /tmp/ebos_stamps/iwe_tiled_45x80x32.o &
wait
OBJS=$(ls event_based_bos_amd/lib/obj/*.o | grep -v "cmax_resident_45x80\.o" | grep -v "iwe_tiled_45x80x32\.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o event_based_bos_amd/lib/libebos_stamps.so /tmp/ebos_stamps/cmax_resident_45x80.o /tmp/ebos_stamps/iwe_tiled_45x80x32.o $OBJS
ls -la event_based_bos_amd/lib/libebos_stamps.so
